# usage: bash scripts/gpu_pmc.sh "<layer substr>" <op>   (PMC passes for one conv kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L="$1"; OP=$2
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcx_$i -o p -- python $R/scripts/pmc_conv.py "$L" $OP > $R/gpurun_out/pmcx_$i.log 2>&1
done
cd $R; python - <<PY
import csv, collections
for i in (1, 2, 3):
    if not __import__('os').path.exists('gpurun_out/pmcx_%d/p_counter_collection.csv' % i): print('set', i, 'missing'); continue
    rows = list(csv.DictReader(open('gpurun_out/pmcx_%d/p_counter_collection.csv' % i)))
    agg = collections.defaultdict(list)
    for r in rows:
        if ('k_conv' in r['Kernel_Name'] or 'k_wgrad' in r['Kernel_Name']) and 'gen' not in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print('%-42s %-28s n=%d mean=%.4g' % (k[0], k[1], len(v), sum(v[1:]) / max(len(v) - 1, 1)))
    kt = list(csv.DictReader(open('gpurun_out/pmcx_%d/p_kernel_trace.csv' % i)))
    print('durations us:', [round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 1) for r in kt if ('k_conv' in r['Kernel_Name'] or 'k_wgrad' in r['Kernel_Name']) and 'gen' not in r['Kernel_Name']][:8])
PY
