# Round 2 evidence refresh (final code): bench lines + rocprofv3 + PMC
R=$GRAFT_REPO_ROOT
bash scripts/gpu_r2p1.sh 2>&1 | tail -14
bash scripts/gpu_r2p2.sh 2>&1 | tail -3
