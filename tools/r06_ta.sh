# round 6: the cost of one vector-memory load instruction per CU (scratch/bench_ta.hip)
cd $GRAFT_REPO_ROOT/scratch && timeout 200 ./bench_ta 2>&1 | tee ../gpurun_out/r06_ta.txt
