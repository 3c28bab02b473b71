set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv_fwd_bwd or rcnn_loss or half" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_ref_tf_golden.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -5
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['ms_per_step'], d['value'])"; done
