R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_model.py -m gpu -q -x -k "early_range or replayed or unannounced or variable or buckets or resnet101_free or fused_two_stream or optimizer_step or non_default or rccl_single" > $O/t.log 2>&1; echo "rc $?" >> $O/t.log; tail -n 12 $O/t.log
B32="python bench.py --no-cpu-baseline --no-other-configs --no-roofline --phases 20 --steps 40 --warmup 10"
B16="python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-roofline --phases 20 --steps 40 --warmup 10"
for i in 1 2; do
  LUMINOTH_AMD_EARLY_UPDATE=0 $B32 > $O/f32_e0_$i.json 2>/dev/null
  LUMINOTH_AMD_EARLY_UPDATE=1 $B32 > $O/f32_e1_$i.json 2>$O/f32_e1_$i.err
done
LUMINOTH_AMD_EARLY_UPDATE=0 $B16 > $O/f16_e0.json 2>/dev/null
LUMINOTH_AMD_EARLY_UPDATE=1 $B16 > $O/f16_e1.json 2>/dev/null
python - <<'P'
import json,os,glob
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5g'
for f in sorted(glob.glob(O+'/f*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ph=d.get('phases_ms') or {}
        print('%-14s %.3f ms median %.3f  joined %.3f bwd %.3f wgrad_joined %.3f tails %.3f next %.3f'%(os.path.basename(f), d['ms_per_step'], d['ms_per_step_median'], ph.get('joined',0), ph.get('trunk_bwd_data_done',0), ph.get('wgrad_stream_joined',0), ph.get('tails_done',0), ph.get('next_step_start',0)))
    except Exception as e: print(os.path.basename(f),'ERR',e, open(f.replace('.json','.err')).read()[-600:] if os.path.exists(f.replace('.json','.err')) else '')
P
