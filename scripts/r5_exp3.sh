# round 5, GPU call 3: schedule A/Bs (aux prologue, CU-partitioned chain) in fp32 and f16; parity of the aux prologue
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
B32="python bench.py --no-cpu-baseline --no-other-configs --no-roofline --phases 20 --steps 40 --warmup 10"
B16="python bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline --no-roofline --phases 20 --steps 40 --warmup 10"
run() { name=$1; shift; env "$@" $B32 > $O/f32_$name.json 2>$O/f32_$name.err; env "$@" $B16 > $O/f16_$name.json 2>$O/f16_$name.err; }
run base      LUMINOTH_AMD_AUX_PROLOGUE=1
run noauxpro  LUMINOTH_AMD_AUX_PROLOGUE=0
run base2     LUMINOTH_AMD_AUX_PROLOGUE=1
# the chain on 1/8 of the compute units of every XCD (32 CUs), the convolution streams on the other 7/8; and 1/4 : 3/4
run cu8       LUMINOTH_AMD_AUX_CU_MASK=8:0:1 LUMINOTH_AMD_MAIN_CU_MASK=8:1:8 LUMINOTH_AMD_SIDE_CU_MASK=8:1:8
run cu4       LUMINOTH_AMD_AUX_CU_MASK=4:0:1 LUMINOTH_AMD_MAIN_CU_MASK=4:1:4 LUMINOTH_AMD_SIDE_CU_MASK=4:1:4
# only the chain confined (the others may use every CU): does a narrower chain hurt by itself?
run auxonly4  LUMINOTH_AMD_AUX_CU_MASK=4:0:1
run roics4    LMH_OPT_ROI_MEAN_CS=4
run rpnside   LUMINOTH_AMD_RPN_BWD_SIDE=1
python -m pytest tests/test_gpu_plan.py tests/test_gpu_model.py -m gpu -q -x -k "replayed or unannounced or variable or next_image or fused_two_stream or buckets" > $O/t.log 2>&1; echo "rc $?" >> $O/t.log
tail -n 4 $O/t.log
python - <<'P'
import json,os,glob
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5c'
for f in sorted(glob.glob(O+'/f*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        ph=d.get('phases_ms') or {}
        print('%-22s %.3f ms median %.3f  fwd %.2f heads %.2f joined %.2f bwd %.2f next %.2f'%(os.path.basename(f), d['ms_per_step'], d['ms_per_step_median'], ph.get('trunk_fwd_done',0), ph.get('rpn_heads_done',0), ph.get('joined',0), ph.get('trunk_bwd_data_done',0), ph.get('next_step_start',0)))
    except Exception as e: print(os.path.basename(f),'ERR',e, open(f.replace('.json','.err')).read()[-300:])
P
