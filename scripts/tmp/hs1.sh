cd $GRAFT_REPO_ROOT
python scripts/tmp/hs_diag.py 2>&1 | grep -v Warning
