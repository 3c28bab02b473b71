"""profiles/<tag>_summary.md + <tag>_kernel_stats.csv from a rocprofv3 --kernel-trace --stats run of bench.py.

usage: python scripts/make_profile_summary.py gpurun_out/prof_<x> <tag> "<bench command>" [timed_steps [skip_last]]
(skip_last: ignore that many steps at the END of the trace — `python bench.py` appends 1 + 3 serialised roofline
profiling steps after its timed region; "20 4" summarises the timed production steps, "3 0" the profiling steps.)

Per-step figures are computed from the kernel TRACE, over the timed steps only: a step ends with its optimizer
launch (k_sgd_momentum / k_optimizer), and only the dispatches between the optimizer launches of the last
`timed_steps` steps are counted — model construction, weight upload (the `__amd_rocclr_copyBuffer` storm of round 1's
table) and warm-up steps are excluded, so "calls/step" and "us/step" are what one steady-state step does.
The raw rocprofv3 stats CSV is copied next to the summary unchanged."""
import csv
import io
import os
import shutil
import subprocess
import sys
from collections import defaultdict

src, tag, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
timed = int(sys.argv[4]) if len(sys.argv) > 4 else 5
skip_last = int(sys.argv[5]) if len(sys.argv) > 5 else 0
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = [f for f in os.listdir(src) if f.endswith('kernel_stats.csv')][0]
trace = [f for f in os.listdir(src) if f.endswith('kernel_trace.csv')][0]
dst_csv = os.path.join(root, 'profiles', '%s_kernel_stats.csv' % tag)
shutil.copy(os.path.join(src, stats), dst_csv)


def short(name):
    name = name.replace('void ', '')
    i = name.find('(')
    return name[:i] if i > 0 else name


rows = []
with open(os.path.join(src, trace)) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
rows.sort()
ends = [i for i, r in enumerate(rows) if 'k_sgd_momentum' in r[2] or 'k_optimizer' in r[2]]
if len(ends) < timed + skip_last + 1:
    raise SystemExit('trace holds %d optimizer launches, need %d' % (len(ends), timed + skip_last + 1))
if skip_last:
    ends = ends[:-skip_last]
lo, hi = ends[-timed - 1] + 1, ends[-1] + 1
win = rows[lo:hi]
wall = (win[-1][1] - rows[ends[-timed - 1]][1]) / timed
agg = defaultdict(lambda: [0, 0])
for s, e, n in win:
    agg[n][0] += e - s
    agg[n][1] += 1
total = sum(v[0] for v in agg.values())
out = io.StringIO()
out.write('# %s — rocprofv3 --kernel-trace --stats of `%s`\n\n' % (tag, cmd))
which = 'the last ones' + (' before the final %d' % skip_last if skip_last else '')
out.write('MI355X.  Per-step figures over %d steps of the trace (%s; dispatches between optimizer launches; model '
          'load and warm-up excluded).  Raw rocprofv3 stats of the whole process: `%s`.\n\n'
          % (timed, which, os.path.relpath(dst_csv, root)))
out.write('Step (optimizer launch to optimizer launch, under the profiler): **%.3f ms**, %.0f kernel launches/step, sum '
          'of kernel durations %.2f ms/step (streams overlap; durations of kernels that share the GPU include the '
          'slow-down from sharing).\n\n' % (wall / 1e6, len(win) / float(timed), total / timed / 1e6))
out.write('| kernel | calls/step | us/step | avg us | % |\n|---|---|---|---|---|\n')
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):       # every kernel of the step: the launch inventory
    nm = n if len(n) <= 70 else n[:67] + '...'
    out.write('| `%s` | %.1f | %.1f | %.1f | %.2f |\n' % (nm, c / float(timed), t / timed / 1e3, t / c / 1e3, 100.0 * t / total))
tl = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'timeline.py'), os.path.join(src, trace), '--skip', str(skip_last)],
                    capture_output=True, text=True).stdout
out.write('\n## Last step by HIP queue (scripts/timeline.py)\n\n```\n%s```\n' % tl)
out.write('\nThe queue with the convolution forward kernels is the main stream (forward, RPN loss + data gradients, trunk '
          'data gradients, update); the queue with the proposal / RCNN chain is the auxiliary stream; the queue carrying the '
          'weight-gradient kernels is the side stream.  rocprofv3 slows the launch path: the un-profiled step is shorter '
          '(bench line next to this file).\n')
open(os.path.join(root, 'profiles', '%s_summary.md' % tag), 'w').write(out.getvalue())
print(out.getvalue()[:2500])
