cd $GRAFT_REPO_ROOT/scratch
for a in "256 16 0 1" "256 16 1 1" "64 16 0 1"; do echo "== bench_fused_ns $a"; timeout 120 ./bench_fused_ns $a 2>&1 | head -2; echo "== bench_fused_st $a"; timeout 120 ./bench_fused_st $a 2>&1 | head -24; done 2>&1 | tee ../gpurun_out/r06_fused_st.txt
