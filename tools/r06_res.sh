# round 6: resident attention backward with operand preloading per phase (scratch/bench_res.hip variants); hashes must agree with p0
cd $GRAFT_REPO_ROOT/scratch
for v in 0 15 1 2 4 8; do echo "== C=64 preload mask $v"; timeout 120 ./bench_res_c64_p$v 64 16 2>&1 | grep -v "^wave 3\|min .* max\|^$" | head -14; done 2>&1 | tee ../gpurun_out/r06_res_c64.txt
for v in 0; do echo "== C=256 base"; timeout 120 ./bench_res_c64_p0 256 16 2>&1 | head -12; done 2>&1 | tee ../gpurun_out/r06_res_c256.txt
for v in 2 4 6; do [ -x ./bench_res_c256_p$v ] && { echo "== C=256 preload mask $v"; timeout 120 ./bench_res_c256_p$v 256 16 2>&1 | head -12; }; done 2>&1 | tee -a ../gpurun_out/r06_res_c256.txt
