R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "pp_equals or conv_fwd_bwd" > $O/t.log 2>&1; echo "rc $?" >> $O/t.log; tail -n 5 $O/t.log
B32="python bench.py --no-cpu-baseline --no-other-configs --no-roofline --phases 20 --steps 40 --warmup 10"
for i in 1 2; do
  LMH_OPT_CONV_PP=0 $B32 > $O/pp0_$i.json 2>/dev/null
  LMH_OPT_CONV_PP=1 $B32 > $O/pp1_$i.json 2>/dev/null
done
python - <<'P'
import json,os,glob
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5f'
for f in sorted(glob.glob(O+'/pp*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); ph=d.get('phases_ms') or {}
    print('%-12s %.3f ms median %.3f  fwd %.3f heads %.3f joined %.3f bwd %.3f'%(os.path.basename(f), d['ms_per_step'], d['ms_per_step_median'], ph.get('trunk_fwd_done',0), ph.get('rpn_heads_done',0), ph.get('joined',0), ph.get('trunk_bwd_data_done',0)))
P
