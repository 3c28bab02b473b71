# Everything under profiles/r03_* in one gpurun call: bench lines + rocprofv3 + PMC
R=$GRAFT_REPO_ROOT
bash scripts/gpu_evidence_lines.sh 2>&1 | tail -14
bash scripts/gpu_evidence_profiles.sh 2>&1 | tail -3
