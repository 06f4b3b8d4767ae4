cd $GRAFT_REPO_ROOT/scratch
for r in 1 2 3; do for b in bench_res_dg bench_res_lb2; do for a in "64 16 0" "64 32 0"; do echo "== $b $a"; timeout 120 ./$b $a 2>&1 | grep -v "^$" | head -2 | cut -c1-200; done; done; done 2>&1 | tee ../gpurun_out/r06_res_lb2.txt
