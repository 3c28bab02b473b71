# MFMA-busy / wait / LDS / L2 counters of the native fp32 kernels on representative layers (rocprofv3 --pmc, separate passes)
for spec in "b3 1x1 256->1024|fwd" "b3 1x1 512->1024|bwd_weight" "b3 1x1 512->1024|bwd_data"; do
  L="${spec%%|*}"; OP="${spec##*|}"
  echo "=== $L $OP"
  bash scripts/gpu_pmc.sh "$L" $OP 2>&1 | grep -E "k_conv|k_wgrad|durations" | grep -v "gen"
done
echo "=== rpn 3x3 fwd, direct kernel (LUMINOTH_AMD_WINOGRAD=0)"
LUMINOTH_AMD_WINOGRAD=0 bash scripts/gpu_pmc.sh "rpn 3x3" fwd 2>&1 | grep -E "k_conv|durations" | grep -v "gen"
