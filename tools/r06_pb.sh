cd $GRAFT_REPO_ROOT/scratch
for r in 1 2 3; do for b in bench_fused_prev4 bench_fused_i32; do for a in "256 16 0 1" "64 16 0 1" "256 16 1 1"; do echo "== $b $a"; timeout 120 ./$b $a 2>&1 | grep -v "^$" | head -2 | cut -c1-200; done; done; done 2>&1 | tee ../gpurun_out/r06_fused_i32.txt
