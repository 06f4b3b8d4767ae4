# round 6: refresh of the build-stamped evidence after a kernel-source edit: the traffic
# counters of configs 1 / 3 / 4 (bench.py reports roofline.traffic only for the stamp of the library it runs), the default line, the suite
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e
mkdir -p $O
timeout 900 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json; tail -2 $O/pmc_traffic.log
timeout 900 python tools/pmc_traffic.py --config 4 > $O/pmc_traffic_config4.log 2>&1; cp profiles/pmc_traffic_config4.json $O/pmc_traffic_config4.json
timeout 900 python tools/pmc_traffic.py --config 3 > $O/pmc_traffic_config3.log 2>&1; cp profiles/pmc_traffic_config3.json $O/pmc_traffic_config3.json
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; grep real $O/bench_default.time; cut -c1-300 $O/bench_default.json; tail -c 700 $O/bench_default.json
timeout 1700 python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -30 > $O/pytest.log
grep -n "passed\|failed" $O/pytest.log
