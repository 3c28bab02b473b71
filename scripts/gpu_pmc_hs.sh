# PMC passes over the half-storage kernels (csrc/conv_hs.h) on representative layers: matrix-pipe busy cycles, LDS bank
# conflicts (the swizzled images are claimed conflict-free), L2 hit rate.  usage: bash scripts/gpu_pmc_hs.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "rpn 3x3|fwd" "rpn 3x3|wgr" "b3 1x1 1024->256|bwd" "b3 1x1 512->1024|wgr" "b1 1x1 64->256|bwd"; do
  L="${spec%%|*}"; OP="${spec##*|}"
  echo "=== $L $OP"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    i=$((i+1))
    rm -rf $R/gpurun_out/pmchs_$i
    timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmchs_$i -o p -- python $R/scripts/bench_conv_hs.py f16 "$L" --only=$OP > $R/gpurun_out/pmchs_$i.log 2>&1
  done
  cd $R; python - <<PY
import csv, collections, os
for i in (1, 2, 3):
    f = 'gpurun_out/pmchs_%d/p_counter_collection.csv' % i
    if not os.path.exists(f): print('set', i, 'missing'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_conv_hs' in r['Kernel_Name'] or 'k_wgrad_hs' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0][-44:], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print('%-46s %-28s n=%d mean=%.4g' % (k[0], k[1], len(v), sum(v[1:]) / max(len(v) - 1, 1)))
PY
  cd /tmp
done
