cd $GRAFT_REPO_ROOT/scratch
for r in 1 2 3; do for b in bench_res_prev5 bench_res_t7; do for a in "256 16 0" "256 16 1" "256 32 0"; do echo "== $b $a"; timeout 120 ./$b $a 2>&1 | grep -v "^$" | head -2 | cut -c1-200; done; done; done 2>&1 | tee ../gpurun_out/r06_res_t7.txt
