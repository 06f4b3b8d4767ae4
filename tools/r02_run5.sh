cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
bash tools/ab_opts.sh "--option conv_variant=2" "--option conv_variant=1" 3 2>&1 | tee $O/ab_conv2.log
bash tools/ab_opts.sh "--config 3 --option conv_variant=2" "--config 3 --option conv_variant=1" 2 2>&1 | tee $O/ab_conv2_c3.log
