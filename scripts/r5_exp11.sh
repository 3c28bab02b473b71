R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j; mkdir -p $O; cd $R
B16="python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-roofline --phases 20 --steps 40 --warmup 10"
for v in middle side aux middle; do LUMINOTH_AMD_PREFIX_AT=$v $B16 > $O/f16_$v$RANDOM.json 2>$O/err_$v.txt; done
python - <<'P'
import json,os,glob
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5j'
for f in sorted(glob.glob(O+'/f*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ph=d.get('phases_ms') or {}
        print('%-20s %.3f ms median %.3f  fwd %.2f heads %.2f joined %.2f bwd %.2f next %.2f'%(os.path.basename(f), d['ms_per_step'], d['ms_per_step_median'], ph.get('trunk_fwd_done',0), ph.get('rpn_heads_done',0), ph.get('joined',0), ph.get('trunk_bwd_data_done',0), ph.get('next_step_start',0)))
    except Exception as e: print(os.path.basename(f),'ERR',e)
P
tail -3 $O/err_side.txt
