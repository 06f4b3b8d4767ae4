cd $GRAFT_REPO_ROOT/scratch
for r in 1 2 3; do for b in bench_fused_prev bench_fused_o3 bench_fused_o4; do for a in "64 16 0 1" "64 16 0 0" "64 32 0 1"; do echo "== $b $a"; timeout 120 ./$b $a 2>&1 | grep -v "^$" | head -2 | cut -c1-200; done; done; done 2>&1 | tee ../gpurun_out/r06_fused_occ4.txt
for a in "64 16 0 1"; do echo "== bench_fused_o4_st $a"; timeout 120 ./bench_fused_o4_st $a 2>&1 | grep -v "^$" | sed -n 3,14p; done 2>&1 | tee -a ../gpurun_out/r06_fused_occ4.txt
