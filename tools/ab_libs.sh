# usage: ab_libs.sh "<bench args>" [rounds]   -- alternates scratch/libA.so and scratch/libB.so on the same box
cd $GRAFT_REPO_ROOT
ARGS="$1"; N=${2:-3}
cp m2trans_amd/libm2t.so /tmp/lib_keep.so
for r in $(seq 1 $N); do
  for v in A B; do
    cp scratch/lib$v.so m2trans_amd/libm2t.so
    python bench.py --no-cpu-baseline --no-kernel-events --no-also $ARGS 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  done
done
cp /tmp/lib_keep.so m2trans_amd/libm2t.so
