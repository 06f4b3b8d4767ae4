cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $@ 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for v in 0 96 128 192 256 384 512; do run --option wgrad_big_tiles=$v; done
done
for rep in 1 2; do
for v in 0 128 256 512; do run --config 3 --option wgrad_big_tiles=$v; done
done
