R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5i; mkdir -p $O; cd $R
B32="python bench.py --no-cpu-baseline --no-other-configs --no-roofline --phases 20 --steps 40 --warmup 10"
B16="python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-roofline --phases 20 --steps 40 --warmup 10"
run16() { name=$1; shift; env "$@" $B16 > $O/f16_$name.json 2>/dev/null; }
run32() { name=$1; shift; env "$@" $B32 > $O/f32_$name.json 2>/dev/null; }
run16 base A=1
run16 cap1 LMH_OPT_HS_SLAB_CAP=1
run16 cap4 LMH_OPT_HS_SLAB_CAP=4
run16 cap0 LMH_OPT_HS_SLAB_CAP=0
run16 et4 LUMINOTH_AMD_EARLY_TAILS=4
run16 et16 LUMINOTH_AMD_EARLY_TAILS=16
run16 base2 A=1
run32 base A=1
run32 et4 LUMINOTH_AMD_EARLY_TAILS=4
run32 et16 LUMINOTH_AMD_EARLY_TAILS=16
python - <<'P'
import json,os,glob
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5i'
for f in sorted(glob.glob(O+'/f*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ph=d.get('phases_ms') or {}
        print('%-14s %.3f ms median %.3f  joined %.3f bwd %.3f wgrad_joined %.3f tails %.3f next %.3f'%(os.path.basename(f), d['ms_per_step'], d['ms_per_step_median'], ph.get('joined',0), ph.get('trunk_bwd_data_done',0), ph.get('wgrad_stream_joined',0), ph.get('tails_done',0), ph.get('next_step_start',0)))
    except Exception as e: print(os.path.basename(f),'ERR',e)
P
