cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x --durations=5 2>&1 | tail -12 > $O/pytest.log
tail -4 $O/pytest.log
( echo "A = first round-5 build, B = + head conv / pack / folded add; config 1"; bash tools/ab_libs.sh "--config 1 --steps 30" 3
  echo "config 3"; bash tools/ab_libs.sh "--config 3 --steps 20" 2
  echo "config 4"; bash tools/ab_libs.sh "--config 4 --steps 20" 2 ) 2>&1 | tee $O/ab_libs.txt
bash tools/kstat.sh "--config 1" "head_conv\|pack_kernel\|add_kernel\|instnorm_bwd_apply" 2>&1 | tail -6
