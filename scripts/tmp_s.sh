cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "sort or proposal or nms" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_ref_tf_golden.py tests/test_gpu_model.py -m gpu -q -x -k "proposal or matches_oracle or prefetch" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32', d['ms_per_step'], d['value'])"
timeout 300 python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f16', d['ms_per_step'], d['value'])"
done
