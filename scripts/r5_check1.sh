# round 5, first GPU contact: the new host logic (live rocprofv3 leg, ladder, variable shapes, head-layout fixtures)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_plan.py tests/test_gpu_ref_tf_golden.py -m gpu -q -k "variable or module_layout" > $O/t1.log 2>&1; echo "t1 rc $?" >> $O/t1.log
python -m pytest tests/test_gpu_ssd.py tests/test_gpu_model.py -m gpu -q -k "free_running or ladder or fused_two_stream or next_image or two_ranks_on_one" > $O/t2.log 2>&1; echo "t2 rc $?" >> $O/t2.log
( time python bench.py --rocprof-keep $O/rp > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --no-other-configs --no-roofline > $O/bench_q8.json 2>/dev/null
python bench.py --no-cpu-baseline --no-other-configs --no-roofline > $O/bench_q4.json 2>/dev/null
python bench.py --no-cpu-baseline --no-other-configs --no-roofline --phases 30 --steps 40 --warmup 10 > $O/phases.json 2>/dev/null
tail -5 $O/t1.log $O/t2.log; cat $O/bench.time; python - <<'P'
import json,glob,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5a'
for f in ('bench','bench_q8','bench_q4','phases'):
    try:
        d=json.loads(open('%s/%s.json'%(O,f)).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f, '%.3f ms median %.3f'%(d['ms_per_step'], d['ms_per_step_median']), {k:r.get(k) for k in ('frac','frac_rocprofv3','frac_raw_event_interval','frac_event_interval_minus_empty_pair')})
        o=(d.get('other_configs') or {}).get('frcnn_r50_coco_f16')
        if o: print('  f16', o['ms_per_step'], {k:(o.get('roofline') or {}).get(k) for k in ('frac','frac_rocprofv3','frac_raw_event_interval','kernel')})
        if d.get('phases_ms'): print(d['phases_ms'])
    except Exception as e: print(f,'ERR',e)
P
tail -3 $O/bench.err
