R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; mkdir -p $O; cd $R
for only in "" heads trunk; do
  echo "== only=$only"; LUMINOTH_AMD_EARLY_UPDATE_ONLY=$only timeout 600 python -m pytest tests/test_gpu_plan.py -m gpu -q -x -k "early_range" 2>&1 | grep -E "passed|failed|assert |Error" | head -5
done
B32="python bench.py --no-cpu-baseline --no-other-configs --no-roofline --phases 20 --steps 40 --warmup 10"
LUMINOTH_AMD_EARLY_UPDATE=0 $B32 > $O/f32_e0.json 2>/dev/null
LUMINOTH_AMD_EARLY_UPDATE=1 $B32 > $O/f32_e1.json 2>/dev/null
python - <<'P'
import json,os,glob
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5h'
for f in sorted(glob.glob(O+'/f*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); ph=d.get('phases_ms') or {}
    print('%-14s %.3f ms median %.3f  joined %.3f bwd %.3f wgrad_joined %.3f tails %.3f next %.3f'%(os.path.basename(f), d['ms_per_step'], d['ms_per_step_median'], ph.get('joined',0), ph.get('trunk_bwd_data_done',0), ph.get('wgrad_stream_joined',0), ph.get('tails_done',0), ph.get('next_step_start',0)))
P
