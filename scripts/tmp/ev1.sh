cd $GRAFT_REPO_ROOT
bash scripts/gpu_evidence_lines.sh 2>&1 | tail -40
