cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
timeout 300 ./scratch/bench_conv_rows 2>&1 | tee $O/bench_conv_rows.log
timeout 300 ./scratch/bench_conv_rows_st 2>&1 | tee $O/bench_conv_rows_stamps.log
