cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q 2>&1 | grep -v "^tensor\|^  \|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
# A/B on one box: side-stream plumbing old (torch contexts) vs new is not switchable; compare against the committed tree's number
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH f32 headline', d['ms_per_step'], d['value'])"
done
git stash -q 2>/dev/null || true
