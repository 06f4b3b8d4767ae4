# final build of round 3: HBM traffic counters (build-stamped), then the four bench lines
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f2; mkdir -p $O
timeout 1200 python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json; tail -18 $O/pmc_traffic.log
for c in 1 2 3 4; do
  extra="--no-cpu-baseline"; [ $c = 1 ] && extra=""
  timeout 600 python bench.py --config $c $extra > $O/bench_c$c.json 2> $O/bench_c$c.err
  cut -c1-260 $O/bench_c$c.json
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --force-comm-path --no-cpu-baseline > $O/bench_c1_rccl_one_rank.json 2> $O/bench_rccl.err; cut -c1-200 $O/bench_c1_rccl_one_rank.json
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
