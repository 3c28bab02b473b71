# round 5, GPU call 2: tile x stagger sweep of the 1x1 layers; RPN backward behind the join (A/B + parity)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
python scripts/r5_tile_stagger_sweep.py > $O/sweep.log 2>&1
B="python bench.py --no-cpu-baseline --no-other-configs --no-roofline --phases 30 --steps 40 --warmup 10"
for i in 1 2; do
  LUMINOTH_AMD_RPN_BWD_LATE=0 $B > $O/late0_$i.json 2>/dev/null
  LUMINOTH_AMD_RPN_BWD_LATE=1 $B > $O/late1_$i.json 2>/dev/null
done
LUMINOTH_AMD_RPN_BWD_LATE=1 python -m pytest tests/test_gpu_model.py tests/test_gpu_plan.py -m gpu -q -x -k "fused_two_stream or test_train_step_matches_oracle or replayed or next_image" > $O/t_late.log 2>&1; echo "rc $?" >> $O/t_late.log
cat $O/sweep.log; tail -n 4 $O/t_late.log
python - <<'P'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5b'
for f in ('late0_1','late1_1','late0_2','late1_2'):
    try:
        d=json.loads(open('%s/%s.json'%(O,f)).read().strip().splitlines()[-1])
        print(f, '%.3f ms median %.3f'%(d['ms_per_step'], d['ms_per_step_median']), d.get('phases_ms'))
    except Exception as e: print(f,'ERR',e)
P
