# round 6: "memory-free" knock-out of the resident attention backward (every workgroup on one of 64 windows: loads / stores hit L2)
cd $GRAFT_REPO_ROOT/scratch
for hot in 0 64; do for c in 64 256; do echo "== M2T_WIN_HOT=$hot C=$c"; timeout 100 ./bench_res_hot${hot}_ns $c 16 | head -1; timeout 100 ./bench_res_hot$hot $c 16 | sed -n 3,11p; done; done 2>&1 | tee ../gpurun_out/r06_res_hot.txt
