"""profiles/<round>_bench_summary.md + <round>_bench_kernel_stats.csv from a rocprofv3 --kernel-trace --stats run of
bench.py (scripts/gpu_prof.sh).  usage: python scripts/make_profile_summary.py gpurun_out/prof_<tag> r01 [steps]"""
import csv
import io
import os
import shutil
import subprocess
import sys

src, rnd = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = [f for f in os.listdir(src) if f.endswith('kernel_stats.csv')][0]
trace = [f for f in os.listdir(src) if f.endswith('kernel_trace.csv')][0]
dst_csv = os.path.join(root, 'profiles', '%s_bench_kernel_stats.csv' % rnd)
shutil.copy(os.path.join(src, stats), dst_csv)
rows = list(csv.DictReader(open(dst_csv)))
total = sum(float(r['TotalDurationNs']) for r in rows)
out = io.StringIO()
out.write('# Round %s — rocprofv3 --kernel-trace --stats of `python bench.py --steps 5 --warmup 2 --no-cpu-baseline '
          '--no-roofline`\n\n' % rnd[1:].lstrip('0'))
out.write('MI355X, Faster R-CNN ResNet-50, batch 2 x 1024x1024, fp32; %d train steps in the trace (2 warm-up + 5 timed).\n'
          'Source: `%s` (copied next to this file as `%s`).\n\n' % (steps, os.path.join(src, stats), os.path.basename(dst_csv)))
out.write('Sum of kernel durations per step: %.2f ms over three concurrent HIP streams (wall-clock per step is the '
          '`ms_per_step` of the bench line; durations of kernels that share the GPU with another stream include the '
          'slow-down from sharing).\n\n' % (total / steps / 1e6))
out.write('| kernel | calls/step | us/step | avg us | % |\n|---|---|---|---|---|\n')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:32]:
    name = r['Name'].replace('void ', '').split('(')[0]
    if len(name) > 70:
        name = name[:67] + '...'
    calls = float(r['Calls']) / steps
    t = float(r['TotalDurationNs'])
    out.write('| `%s` | %.1f | %.1f | %.1f | %.2f |\n' % (name, calls, t / steps / 1e3, t / float(r['Calls']) / 1e3,
                                                      100.0 * t / total))
tl = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'timeline.py'), os.path.join(src, trace)],
                    capture_output=True, text=True).stdout
out.write('\n## Last step by HIP queue (scripts/timeline.py)\n\n```\n%s```\n' % tl)
out.write('\nqueue 1 = main stream (forward, RPN loss + data gradients, trunk data gradients, update); the queue with the '
          'proposal / RCNN chain is the auxiliary stream; the queue carrying `k_conv_bwd_weight` is the weight-gradient '
          'side stream.  Gaps listed under the profiler are partly host-side (rocprofv3 slows the launch path: the '
          'un-profiled step is ~0.4 ms shorter).\n')
open(os.path.join(root, 'profiles', '%s_bench_summary.md' % rnd), 'w').write(out.getvalue())
print(out.getvalue()[:1500])
