# round 5, first GPU call: whole -m gpu suite on the new build, the default bench line (with its `also` list), same-box A/B of
# the round-4 library (scratch/libA.so) against the new one (scratch/libB.so) on configs 4 / 1 / 3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 2>&1 | tail -25 > $O/pytest.log
tail -4 $O/pytest.log
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
cut -c1-400 $O/bench_default.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05a/bench_default.json").read().strip().splitlines()[-1])
for a in d.get("also", []):
    print({k: a.get(k) for k in ("workload", "value", "ms_per_step", "error", "skipped")}, (a.get("dominant_kernel") or {}).get("category"), (a.get("dominant_kernel") or {}).get("frac"))
print("others:", [(o["category"], o["est_ms_per_step"]) for o in (d["roofline"].get("others") or [])])
PY
( echo "A = round-4 library, B = round-5 build; config 4"; bash tools/ab_libs.sh "--config 4 --steps 20" 3
  echo "config 1"; bash tools/ab_libs.sh "--config 1 --steps 30" 3
  echo "config 3"; bash tools/ab_libs.sh "--config 3 --steps 20" 2 ) > $O/ab_libs.txt 2>&1
cat $O/ab_libs.txt
