cd $GRAFT_REPO_ROOT/scratch
for r in 1 2 3; do for b in bench_res_prev bench_res_dg; do for a in "256 16 0" "256 16 1" "64 16 0"; do echo "== $b $a"; timeout 120 ./$b $a 2>&1 | grep -v "^$" | head -2 | cut -c1-200; done; done; done 2>&1 | tee ../gpurun_out/r06_dg.txt
for a in "256 16 0" "64 16 0"; do echo "== bench_res_dg_st $a"; timeout 120 ./bench_res_dg_st $a 2>&1 | grep -v "^$" | sed -n 3,11p; done 2>&1 | tee -a ../gpurun_out/r06_dg.txt
