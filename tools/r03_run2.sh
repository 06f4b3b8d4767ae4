# round 3, call 2: after retiring the variants: GPU suite, bf16 probe v2, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout 900 python tools/bf16_quality.py --json $O/bf16_quality.json > $O/bf16_quality.log 2>&1; tail -22 $O/bf16_quality.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
timeout 2400 python -m pytest tests -q -m gpu -x -s -k "not drift_at_a_25dB" 2>&1 | grep -v "^$" | tail -40 | tee $O/pytest.log
