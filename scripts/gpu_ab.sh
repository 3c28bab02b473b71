# A/B/n on one box: bash scripts/gpu_ab.sh "ENV_A=..." "ENV_B=..." ["ENV_C=..."]  (3 alternating runs each)
for i in 1 2 3; do
  for cfg in "$@"; do
    echo -n "[$cfg] "; env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))"
  done
done
