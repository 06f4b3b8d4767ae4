cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "per_plane_on_the_side or bitwise_reproducible or c16_prep_launch or reference_training or graph_capturable" 2>&1 | tail -6 | tee gpurun_out/r06_norm_tests.txt
bash tools/ab_libs.sh "--steps 20" 3 2>&1 | tee gpurun_out/r06_ab_norm_config1.txt
bash tools/ab_libs.sh "--config 3 --steps 10" 2 2>&1 | tee gpurun_out/r06_ab_norm_config3.txt
bash tools/ab_libs.sh "--config 4 --steps 10" 2 2>&1 | tee gpurun_out/r06_ab_norm_config4.txt
