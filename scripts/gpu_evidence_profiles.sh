# Round 3 evidence, part 2: rocprofv3 kernel trace + stats of `python bench.py --no-cpu-baseline` (20 production steps,
# then the 1 + 3 serialised roofline steps) and FETCH_SIZE / WRITE_SIZE passes over the single-stream schedule
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03 -o r03 -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_r03_line.json 2> $R/gpurun_out/prof_r03.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_r03_$c -o p -- python $R/bench.py --serial --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_r03_$c.log 2>&1
done
cd $R
python scripts/pmc_traffic.py gpurun_out/pmc_r03_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_r03_WRITE_SIZE/p_counter_collection.csv gpurun_out/pmc_r03_traffic.json | head -14
python scripts/make_profile_summary.py gpurun_out/prof_r03 r03_bench "python bench.py --no-cpu-baseline (timed production steps)" 20 4 | head -30
python scripts/make_profile_summary.py gpurun_out/prof_r03 r03_bench_roofline_steps "python bench.py --no-cpu-baseline (the 3 serialised roofline steps at its end)" 3 0 | head -16
head -c 300 gpurun_out/prof_r03_line.json

# half-storage f16 step (BASELINE configs[4]): kernel trace + HBM traffic passes
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_f16hs -o r03 -- python $R/bench.py --workload frcnn_r50_coco --dtype f16 --no-cpu-baseline > $R/gpurun_out/prof_r03_f16hs_line.json 2> $R/gpurun_out/prof_r03_f16hs.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_r03_f16hs_$c -o p -- python $R/bench.py --workload frcnn_r50_coco --dtype f16 --serial --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_r03_f16hs_$c.log 2>&1
done
cd $R
python scripts/pmc_traffic.py gpurun_out/pmc_r03_f16hs_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_r03_f16hs_WRITE_SIZE/p_counter_collection.csv gpurun_out/pmc_r03_f16hs_traffic.json | head -12
