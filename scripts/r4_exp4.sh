R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04e; mkdir -p $O
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline --phases 20"
B5="python bench.py --workload frcnn_r50_coco --dtype f16 --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --phases 20"
cd $R
LUMINOTH_AMD_PREFIX_SPLIT=0 $B > $O/f32_nosplit.json 2>/dev/null
$B > $O/f32_split.json 2>/dev/null
LUMINOTH_AMD_PREFIX_SPLIT=0 $B5 > $O/f16_nosplit.json 2>/dev/null
$B5 > $O/f16_split.json 2>/dev/null
LUMINOTH_AMD_RPN_BWD_SIDE=1 $B5 > $O/f16_split_rpnside.json 2>/dev/null
for f in $O/*.json; do python - $f <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%-28s %.3f ms  median %.3f  min %.3f  %.1f img/s'%(sys.argv[1].split('/')[-1], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'], d['value']))
    if 'phases_ms' in d: print('   ', {k:v for k,v in d['phases_ms'].items() if not k.startswith('host')})
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
