cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
timeout 1200 python tools/bf16_quality.py --json $O/bf16_quality.json > $O/bf16_quality.log 2>&1; tail -22 $O/bf16_quality.log
timeout 2400 python -m pytest tests -q -m gpu -x -s -k "not drift_at_a_25dB" 2>&1 | grep -v "^$" | tail -40 | tee $O/pytest.log
