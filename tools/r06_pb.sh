cd $GRAFT_REPO_ROOT/scratch
for b in bench_res_base bench_res_i32b; do for a in "256 16 1"; do echo "== $b $a"; timeout 120 ./$b $a 2>&1 | grep -v "^$" | head -3 | cut -c1-170; done; done 2>&1 | tee ../gpurun_out/r06_det.txt
