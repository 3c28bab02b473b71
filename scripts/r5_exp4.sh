# round 5, GPU call 4: narrow slabs of the fused ROI pooling (LDS 128 KB -> 16 KB per block) alone and inside both steps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "roi" > $O/t.log 2>&1; echo "rc $?" >> $O/t.log; tail -n 5 $O/t.log
python scripts/bench_roi.py > $O/bench_roi.log 2>&1; tail -n 20 $O/bench_roi.log
B32="python bench.py --no-cpu-baseline --no-other-configs --no-roofline --phases 20 --steps 40 --warmup 10"
B16="python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-roofline --phases 20 --steps 40 --warmup 10"
for cs in -1 1 2 4 -1 1; do
  LMH_OPT_ROI_MEAN_CS=$cs $B32 > $O/f32_cs${cs}_$RANDOM.json 2>/dev/null
  LMH_OPT_ROI_MEAN_CS=$cs $B16 > $O/f16_cs${cs}_$RANDOM.json 2>/dev/null
done
python - <<'P'
import json,os,glob
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5d'
for f in sorted(glob.glob(O+'/f*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        ph=d.get('phases_ms') or {}
        print('%-26s %.3f ms median %.3f  heads %.2f prop %.2f rcnn_tgt %.2f rcnn_loss %.2f rcnn_bwd %.2f joined %.2f bwd %.2f'%(os.path.basename(f), d['ms_per_step'], d['ms_per_step_median'], ph.get('rpn_heads_done',0), ph.get('aux:proposals_done',0), ph.get('aux:rcnn_targets_done',0), ph.get('aux:rcnn_loss_done',0), ph.get('aux:rcnn_bwd_done',0), ph.get('joined',0), ph.get('trunk_bwd_data_done',0)))
    except Exception as e: print(os.path.basename(f),'ERR',e)
P
