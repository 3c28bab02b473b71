cd $GRAFT_REPO_ROOT
bash scripts/gpu_evidence_profiles.sh 2>&1 | tail -50
