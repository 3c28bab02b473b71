# Round 2, GPU call X: phase timelines on one box (with / without look-ahead), smoke(), 2-rank control-flow run on one GPU
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for la in "" "--no-lookahead"; do
  timeout 200 python bench.py --no-cpu-baseline --no-roofline --phases 10 $la > "gpurun_out/r2x_bench$la.json" 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r2x_bench$la.json"))
print("'$la':", round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms;", json.dumps(d["phases_ms"]))
PY
done
LUMINOTH_AMD_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -3 | cut -c1-400
