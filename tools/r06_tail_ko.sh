cd $GRAFT_REPO_ROOT/scratch
for ko in 1 2 4 8 16 32 3 127; do echo "== T32_KO=$ko"; timeout 120 ./bench_tail_ko$ko 2>&1 | grep -A1 "B=16 LR 128x128 (L1" | grep "kernel"; done 2>&1 | tee ../gpurun_out/r06_tail32_ko.txt
