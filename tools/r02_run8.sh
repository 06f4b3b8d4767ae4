cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/ab_opts.sh "--option last_block_conv_first=1" "--option last_block_conv_first=0" 3
bash tools/ab_opts.sh "--config 3 --option last_block_conv_first=1" "--config 3 --option last_block_conv_first=0" 2
