# round 6: selected GPU tests on the new build, then same-box A/B of scratch/libA.so (base) against scratch/libB.so (new)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py -q -m gpu -x -k "${1:-tail or l1_seed or reference_training or bitwise}" 2>&1 | tail -6 | tee gpurun_out/r06_ab_tests.txt
bash tools/ab_libs.sh "--steps 20" 3 2>&1 | tee gpurun_out/r06_ab_config1.txt
bash tools/ab_libs.sh "--config 3 --steps 10" 2 2>&1 | tee gpurun_out/r06_ab_config3.txt
