# round 3, call 1: state of the tree at round start (GPU suite), the bf16 quality probe, a bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
timeout 600 python tools/bf16_quality.py --json $O/bf16_quality.json > $O/bf16_quality.log 2>&1; tail -30 $O/bf16_quality.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest.log
