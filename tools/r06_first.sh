cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "deferred_l1_seed or reference_training_loop or data_parallel_over_two" 2>&1 | tail -15 > gpurun_out/r06_first_new.txt
cat gpurun_out/r06_first_new.txt
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_first_all.txt
cat gpurun_out/r06_first_all.txt
python bench.py > gpurun_out/r06_first_bench.json 2> gpurun_out/r06_first_bench.err
tail -c 2100 gpurun_out/r06_first_bench.json
