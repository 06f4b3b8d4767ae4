cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in st0 st4; do
  echo "== $v"; timeout 300 ./scratch/bench_tail_$v | tail -4
done 2>&1 | tee gpurun_out/r04_tail_st.txt
