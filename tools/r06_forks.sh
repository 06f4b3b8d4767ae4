cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "bitwise_reproducible or fork_event or graph_capturable or reference_training" 2>&1 | tail -3 | tee gpurun_out/r06_forks_tests.txt
( echo "A = merge_c256_forks 0, B = default (1)"; bash tools/ab_opts.sh "--option merge_c256_forks=0" "" 4; echo "config 3"; bash tools/ab_opts.sh "--config 3 --option merge_c256_forks=0" "--config 3" 3; echo "config 4"; bash tools/ab_opts.sh "--config 4 --option merge_c256_forks=0" "--config 4" 2 ) 2>&1 | tee gpurun_out/r06_ab_forks.txt
