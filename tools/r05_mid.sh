cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x --durations=5 2>&1 | tail -12 > $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python scratch/meas_x23_tail_tol.py 2>&1 | tail -10 | tee $O/x23_tol.txt
