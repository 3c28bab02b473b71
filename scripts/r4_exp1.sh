R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b; mkdir -p $O
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline"
cd $R
$B > $O/base.json 2>/dev/null
LMH_OPT_ROI_MEAN_CS=4 $B > $O/roi_mean_cs4.json 2>/dev/null
LUMINOTH_AMD_INLINE_LAYERS=3 $B > $O/inline3.json 2>/dev/null
LUMINOTH_AMD_INLINE_LAYERS=2 $B > $O/inline2.json 2>/dev/null
LUMINOTH_AMD_RPN_BWD_SIDE=1 $B --phases 20 > $O/rpn_side.json 2>$O/rpn_side.err
LUMINOTH_AMD_RPN_BWD_SIDE=1 LMH_OPT_ROI_MEAN_CS=4 $B > $O/rpn_side_cs4.json 2>/dev/null
LUMINOTH_AMD_RPN_BWD_SIDE=1 LMH_OPT_ROI_MEAN_CS=4 LUMINOTH_AMD_INLINE_LAYERS=3 $B > $O/rpn_side_cs4_inl3.json 2>/dev/null
$B > $O/base2.json 2>/dev/null
python -m pytest tests/test_gpu_plan.py -x -q -m gpu 2>&1 | tail -n 5 > $O/plan_tests.log
for f in $O/*.json; do python - $f <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%-28s %.3f ms  median %.3f  min %.3f  %.1f img/s'%(sys.argv[1].split('/')[-1], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'], d['value']))
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
