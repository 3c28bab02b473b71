cd $GRAFT_REPO_ROOT
timeout 600 python scripts/bench_conv_hs.py f16 --only=wgr 2>&1 | grep -v Warning
timeout 1500 python -m pytest tests/test_gpu_hs.py -m gpu -q 2>&1 | grep -v "^tensor\|^  " | tail -8
for a in "" "--batch 8"; do
timeout 300 python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-roofline $a 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH f16 $a', d['ms_per_step'], d['value'], d['config']['final_total_loss'])"
done
