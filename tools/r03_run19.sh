cd $GRAFT_REPO_ROOT
ab() { python bench.py --no-cpu-baseline --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}; print('$1', d['value'], d['ms_per_step'], r.get('category'), r.get('avg_launch_us'), r.get('launches'), r.get('frac'))"; }
for i in 1 2 3; do
ab events_1_in_3 ""
ab events_every "--kernel-event-stride 1"
ab no_events "--no-kernel-events"
done
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1500
