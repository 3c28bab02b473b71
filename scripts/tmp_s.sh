cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ref_tf_golden.py -m gpu -q -x -k "nms or proposal" 2>&1 | tail -3
python scripts/bench_nms.py 2>&1 | grep -v Warning | tail -2
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ssd.py -m gpu -q -x -k "matches_oracle or prefetch or inference or predict" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32', d['ms_per_step'], d['value'])"
timeout 300 python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f16', d['ms_per_step'], d['value'])"
done
