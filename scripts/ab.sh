#!/bin/bash
# ONE A/B runner for schedule / tuning switches (replaces scripts/r4_exp*.sh, r5_exp*.sh, r5_chk*.sh, r6_ab.sh, r6_base.sh).
# Every variant is a set of environment variables in front of the same bench.py command, run back to back on ONE box
# (`gpurun -- bash scripts/ab.sh ...`: box-to-box differences are 3-5 % of the step, larger than most effects), REPS times
# in alternation (A B A B), with the phase marks of the step printed beside the step time.
#
#   bash scripts/ab.sh [-r REPS] [-a "<bench.py args>"] [-t "<pytest -k expression to run first>"] "ENV=.. ENV=.." "ENV=.." ...
#   e.g.  bash scripts/ab.sh -r 2 -a "--dtype bf16x3" "" "LMH_OPT_X3_PIPE=1" "LUMINOTH_AMD_INLINE_LAYERS=10"
#         bash scripts/ab.sh -a "--workload frcnn_r50_coco --dtype f16" "" "LMH_OPT_HS_SLAB_CAP=4"
# An empty string is the default configuration.  Output: gpurun_out/ab/<n>_<variant>.json + one summary line per run; copy the
# summary into profiles/ when it decides something (profiles/r04_*_ab.log, r05_schedule_ab.md, r06_ab.md are such logs).
#
# Switches the deleted one-off scripts exercised (what each one is: the module that reads it; verdicts: DESIGN.md 4.2 / docs/history.md):
#   LMH_OPT_<NAME>=<int>            any library option of csrc/api.hip (ROI_MEAN_CS, NMS_STAGE_MULT, HEAD_GEMM, CONV_PP, HS_SLAB_CAP,
#                                   X3_PIPE, X3_TILE_SLOTS, X3_BW_SLOTS, BW_SLOTS, WG_SLOTS, ...), forwarded by luminoth_amd/_lib.py
#   LUMINOTH_AMD_INLINE_LAYERS=n    weight gradients of the last n trunk layers on the main stream (layers.py)
#   LUMINOTH_AMD_RPN_BWD_SIDE=0|1|auto, LUMINOTH_AMD_RPN_BWD_LATE=0|1      where the RPN backward runs (fasterrcnn.py)
#   LUMINOTH_AMD_PREFIX_AT=middle|side|aux, LUMINOTH_AMD_PREFIX_SPLIT=0|1  where the next batch's frozen prefix runs
#   LUMINOTH_AMD_RCNN_LOSS_LATE=0|1, LUMINOTH_AMD_WINO_BWD_LATE=0|1|2, LUMINOTH_AMD_EARLY_TAILS=n, LUMINOTH_AMD_EARLY_UPDATE=0|1
#   LUMINOTH_AMD_SIDE_STREAM=0, LUMINOTH_AMD_{SIDE,AUX,MAIN}_CU_MASK=den:lo:hi   stream removal / CU-masked streams
#   LUMINOTH_AMD_PLAN=0             eager launches instead of the recorded launch plan
REPS=1; ARGS=""; TESTS=""
while getopts "r:a:t:" o; do case $o in r) REPS=$OPTARG;; a) ARGS=$OPTARG;; t) TESTS=$OPTARG;; esac; done
shift $((OPTIND - 1))
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out/ab; mkdir -p $O
if [ -n "$TESTS" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$TESTS" 2>&1 | tail -n 3; fi
n=0
for rep in $(seq 1 $REPS); do
  for E in "$@"; do
    n=$((n + 1)); tag=$(echo "${E:-default}" | tr ' /=' '__:')
    env $E python bench.py $ARGS --steps 40 --warmup 10 --phases 20 --no-cpu-baseline --no-other-configs --no-native --no-roofline > $O/${n}_$tag.json 2>/dev/null
    python - "$O/${n}_$tag.json" "${E:-default}" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); ph = d.get('phases_ms') or d.get('phases') or {}
    print('%-44s %.3f ms  median %.3f  min %.3f | fwd %.2f heads %.2f props %.2f joined %.2f bwd %.2f | launches %s' % (
        sys.argv[2][:44], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'], ph.get('trunk_fwd_done', 0),
        ph.get('rpn_heads_done', 0), ph.get('aux:proposals_done', 0), ph.get('joined', 0), ph.get('trunk_bwd_data_done', 0),
        d['config']['launch_plan']['kernel_launches_per_step']))
except Exception as e:
    print(sys.argv[2], 'ERR', e)
P
  done
done
