# Round-end evidence run on the GPU box: tests, smoke, kernel-trace profile, PMC traffic, default bench line.
R=$GRAFT_REPO_ROOT
cd $R
[ -n "$SKIP_TESTS" ] || python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/gpu_prof.sh final > /dev/null 2>&1
bash scripts/gpu_pmc_bench.sh final > gpurun_out/pmc_final.log 2>&1; tail -3 gpurun_out/pmc_final.log
cd $R
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 3000 gpurun_out/bench_final.json
