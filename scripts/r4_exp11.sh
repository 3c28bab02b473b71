R=$GRAFT_REPO_ROOT
cd $R
L=luminoth_amd/csrc
cp $L/libluminoth_hip.so $L/libluminoth_hip_new.so
bash scripts/r4_trace.sh r04v_new
cp $L/libluminoth_hip_base.so $L/libluminoth_hip.so
bash scripts/r4_trace.sh r04v_base
cp $L/libluminoth_hip_new.so $L/libluminoth_hip.so
LMH_OPT_ROI_MEAN_CS=4 bash scripts/r4_trace.sh r04v_cs4
for t in new base cs4; do echo "=== $t f32"; cat gpurun_out/r04v_$t/chain32.txt; echo "=== $t f16"; cat gpurun_out/r04v_$t/chain16.txt; done
