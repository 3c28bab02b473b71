cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 1200 python -m pytest tests/test_gpu_ssd.py -m gpu -q -s -k "train_step_matches_oracle" 2>&1 | tail -15 | cut -c1-300
done
