#!/bin/bash
# PMC counters of ONE layer's forward kernel under the two bf16x3 schedules (scripts/bench_conv_plan.py filter)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_pmc; rm -rf $O; mkdir -p $O
L="${1:-b3 1x1 256->1024}"
cd /tmp && export TMPDIR=/tmp
for P in 0 1; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC"; do
    tag=$(echo $set | cut -c1-12 | tr ' ' '_')
    BENCH_N=4 BENCH_COMPUTE=bf16x3 LMH_OPT_X3_PIPE=$P timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p${P}_$tag -o p -- python $R/scripts/bench_conv_plan.py "$L" > $O/p${P}_$tag.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'r6_pmc')
for P in (0, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, 'p%d_*' % P, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if 'k_x3' not in k:
                continue
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in agg.items():
        print('pipe', P, k)
        for c, v in sorted(cs.items()):
            print('   %-28s %14.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
rm -rf $O/p*_*/
