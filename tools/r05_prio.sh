cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q
mkdir -p $O
( echo "config 2: A = main priority 0, B = main priority -1 (side inherits)"; bash tools/ab_opts.sh "--config 2 --main-priority 0" "--config 2 --main-priority -1" 3
  echo "config 1: A = 0, B = -1"; bash tools/ab_opts.sh "--config 1 --main-priority 0" "--config 1 --main-priority -1" 2
  echo "libs: A = first round-5 build, B = now (pack chunk 8192, templated apply); config 1"; bash tools/ab_libs.sh "--config 1 --steps 30" 3
  echo "config 3"; bash tools/ab_libs.sh "--config 3 --steps 20" 2 ) 2>&1 | tee $O/ab.txt
bash tools/kstat.sh "--config 1" "pack_kernel\|add_kernel\|instnorm_bwd_apply" 2>&1 | tail -6
