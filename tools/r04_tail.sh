# round 4: stand-alone tail kernels, committed build (base) against the work-in-progress build (new); hashes must agree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in base new; do
  echo "== $v"; timeout 300 ./scratch/bench_tail_$v $1
done 2>&1 | tee gpurun_out/r04_tail.txt
