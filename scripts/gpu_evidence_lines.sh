# Round 3 evidence, part 1: bench lines of every workload / dtype (copied into profiles/ afterwards)
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/ev
timeout 600 python bench.py > gpurun_out/ev/r03_bench_line.json 2> gpurun_out/ev/r03_bench_line.err; tail -c 600 gpurun_out/ev/r03_bench_line.json; echo
timeout 200 python bench.py --serial --no-cpu-baseline > gpurun_out/ev/r03_bench_serial_line.json 2>/dev/null
timeout 200 python bench.py --no-lookahead --no-cpu-baseline --no-roofline > gpurun_out/ev/r03_bench_nolookahead_line.json 2>/dev/null
timeout 300 python bench.py --alt --no-cpu-baseline --no-roofline > gpurun_out/ev/r03_bench_with_alt.json 2>/dev/null
for dt in f32 f16 bf16 bf16x3; do
  timeout 200 python bench.py --workload frcnn_r50_coco --dtype $dt --no-cpu-baseline > gpurun_out/ev/r03_bench_frcnn_r50_coco_$dt.json 2>/dev/null
done
# half-storage trunk (round 3): the f16 / bf16 lines above use it; the round-2 path and a larger per-GPU batch beside them
timeout 200 python bench.py --workload frcnn_r50_coco --dtype f16 --fp32-storage --no-cpu-baseline > gpurun_out/ev/r03_bench_frcnn_r50_coco_f16_fp32storage.json 2>/dev/null
for dt in f32 f16 bf16; do
  timeout 200 python bench.py --workload frcnn_r50_coco --dtype $dt --batch 8 --no-cpu-baseline > gpurun_out/ev/r03_bench_frcnn_r50_coco_${dt}_batch8.json 2>/dev/null
done
timeout 300 python scripts/bench_conv_hs.py f16 > gpurun_out/ev/r03_per_layer_conv_hs.log 2>/dev/null; tail -3 gpurun_out/ev/r03_per_layer_conv_hs.log
timeout 300 python scripts/bench_conv_hs.py bf16 >> gpurun_out/ev/r03_per_layer_conv_hs.log 2>/dev/null
timeout 300 python bench.py --workload frcnn_vgg16 --cpu-steps 3 > gpurun_out/ev/r03_bench_frcnn_vgg16_f32.json 2>/dev/null
timeout 300 python bench.py --workload ssd300_b32 --no-cpu-baseline > gpurun_out/ev/r03_bench_ssd300_b32_f32.json 2>/dev/null
timeout 300 python bench.py --workload frcnn_r101 --no-cpu-baseline > gpurun_out/ev/r03_bench_frcnn_r101_f32.json 2>/dev/null
python - <<PY
import glob, json
for f in sorted(glob.glob('gpurun_out/ev/*.json')):
    try:
        d = json.load(open(f))
        r = d.get('roofline') or {}
        a = d.get('alt_arithmetic') or {}
        print('%-48s %8.1f img/s %8.3f ms  %s %s frac %s  cpu %s  alt %s %s' % (f.split('/')[-1], d['value'], d['ms_per_step'], r.get('kernel'), r.get('bound'), round(r['frac'], 3) if r else None, (d.get('cpu_baseline') or {}).get('value'), a.get('dtype'), round(a['value'], 1) if a else None))
    except Exception as e:
        print(f, 'FAILED', e)
PY
