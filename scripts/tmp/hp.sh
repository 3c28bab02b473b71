cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_hs.py -m gpu -q -x -k "kernels or vs_oracle" 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --workload frcnn_r50_coco --dtype f16 --batch 8 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B8', d['ms_per_step'], d['value'])"
timeout 300 python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B2', d['ms_per_step'], d['value'])"
done
