cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "x2_x3_row_streaming or tail or l1_seed" 2>&1 | tail -4 | tee gpurun_out/r06_ab4_tests.txt
bash tools/ab_libs.sh "--config 4 --steps 10" 3 2>&1 | tee gpurun_out/r06_ab_config4.txt
bash tools/ab_libs.sh "--steps 20" 2 2>&1 | tee gpurun_out/r06_ab_config1_b.txt
