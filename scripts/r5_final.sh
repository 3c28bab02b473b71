R=$GRAFT_REPO_ROOT; cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash scripts/r5_evidence.sh 2>&1 | tail -45
