R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04d; mkdir -p $O
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline"
B5="python bench.py --workload frcnn_r50_coco --dtype f16 --steps 60 --warmup 15 --no-cpu-baseline --no-roofline"
cd $R
python scripts/probe_cu_mask.py 2>&1 | grep -v amdgpu.ids > $O/probe_cu_mask.log
$B > $O/free_for_all.json 2>/dev/null
for m in 4:1 3:1 2:1 4:3 8:7; do
  LUMINOTH_AMD_SIDE_CU_MASK=$m $B > $O/mask_${m/:/of}.json 2>$O/mask_${m/:/of}.err
done
LUMINOTH_AMD_SIDE_STREAM=0 $B > $O/serial_wgrad.json 2>$O/serial_wgrad.err
$B > $O/free_for_all_again.json 2>/dev/null
$B5 > $O/f16_free_for_all.json 2>/dev/null
LUMINOTH_AMD_SIDE_CU_MASK=2:1 $B5 > $O/f16_mask_2of1.json 2>/dev/null
LUMINOTH_AMD_SIDE_CU_MASK=4:1 $B5 > $O/f16_mask_4of1.json 2>/dev/null
LUMINOTH_AMD_SIDE_STREAM=0 $B5 > $O/f16_serial_wgrad.json 2>/dev/null
cat $O/probe_cu_mask.log
for f in $O/*.json; do python - $f <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%-28s %.3f ms  median %.3f  min %.3f  %.1f img/s'%(sys.argv[1].split('/')[-1], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'], d['value']))
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
