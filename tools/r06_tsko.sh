# round 6: knock-outs of the forward row-streaming tail (scratch/bench_tail.hip -DTS_KO=mask: 1 GELU -> bias add, 2 no barrier, 4 no consumer)
cd $GRAFT_REPO_ROOT/scratch
for ko in 0 1 2 4 7; do echo "== TS_KO=$ko"; timeout 200 ./bench_tail_tsko$ko 2>&1 | grep "B=16 LR 128x128" -A40 | grep "tail_fwd_stream, 4 workgroups\|tail_fwd_stream seg rows   0"; done 2>&1 | tee ../gpurun_out/r06_tail_fwd_ko.txt
