"""Compact summary of bench.py JSON lines: python scripts/r6_line.py tag < line.json"""
import json
import sys
tag = sys.argv[1] if len(sys.argv) > 1 else ''
for ln in sys.stdin.read().strip().splitlines():
    try:
        d = json.loads(ln)
    except Exception:
        continue
    r = d.get('roofline') or {}
    print('%-14s %-7s %.3f ms (median %.3f) %.1f img/s  launches %s  %s frac %s' % (
        tag, d.get('dtype'), d['ms_per_step'], d.get('ms_per_step_median', 0), d['value'],
        d['config']['launch_plan']['kernel_launches_per_step'], r.get('kernel'), ('%.3f' % r['frac']) if r.get('frac') else None))
    if d.get('phases_ms'):
        print('   phases', d['phases_ms'])
    o = (d.get('other_configs') or {}).get('frcnn_r50_coco_f16')
    if o:
        ro = o.get('roofline') or {}
        print('   f16 leg %.3f ms  %s frac %s' % (o['ms_per_step'], ro.get('kernel'), ro.get('frac')))
