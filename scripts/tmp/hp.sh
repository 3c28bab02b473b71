cd $GRAFT_REPO_ROOT
timeout 600 python scripts/bench_conv_hs.py f16 --sweep 2>&1 | grep -v Warning
