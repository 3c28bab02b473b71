#!/bin/bash
# round 6 baseline on one box: headline f32 + bf16x3 alt, then bf16x3 with its own roofline leg
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --alt > gpurun_out/r6_base_f32_alt.json 2> gpurun_out/r6_base_f32_alt.err
python bench.py --dtype bf16x3 --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --phases 20 > gpurun_out/r6_base_x3.json 2> gpurun_out/r6_base_x3.err
tail -c 600 gpurun_out/r6_base_f32_alt.err gpurun_out/r6_base_x3.err
python - <<'PY'
import json
for f in ('gpurun_out/r6_base_f32_alt.json', 'gpurun_out/r6_base_x3.json'):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['dtype'], round(d['ms_per_step'], 3), round(d['value'], 1), (d.get('alt_arithmetic') or {}).get('ms_per_step'),
              d['roofline']['kernel'], round(d['roofline']['frac'], 3), d['config']['launch_plan'])
        print(d.get('phases_ms'))
    except Exception as e:
        print(f, 'ERR', e)
PY
