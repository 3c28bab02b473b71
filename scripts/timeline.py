"""Per-queue timeline of the LAST train step in a rocprofv3 kernel trace.

usage: python scripts/timeline.py gpurun_out/prof_<tag>/r01_kernel_trace.csv [--dump]

Finds the last k_sgd_momentum launch (end of a step) and the one before it, then reports per HIP queue: busy
time, span, number of kernels and the largest idle gaps, plus the top kernels by time on every queue.  Used to
see which of the three streams (main / weight-gradient side / proposal+RCNN aux) bounds the step.
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace('void ', '')
    i = name.find('(')
    return (name[:i] if i > 0 else name)[:64]


def main():
    path = sys.argv[1]
    dump = '--dump' in sys.argv
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Queue_Id']), r['Kernel_Name']))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if 'k_sgd_momentum' in r[3]]
    if '--skip' in sys.argv:          # ignore that many steps at the end of the trace
        skip = int(sys.argv[sys.argv.index('--skip') + 1])
        ends = ends[:-skip] if skip else ends
    if len(ends) < 2:
        raise SystemExit('need two optimizer launches in the trace')
    lo, hi = ends[-2] + 1, ends[-1] + 1
    step = rows[lo:hi]
    t0 = rows[ends[-2]][1]
    t1 = step[-1][1]
    print('step: %.3f ms, %d kernels' % ((t1 - t0) / 1e6, len(step)))
    byq = defaultdict(list)
    for r in step:
        byq[r[2]].append(r)
    for q, ks in sorted(byq.items()):
        busy = sum(e - s for s, e, _, _ in ks)
        print('\nqueue %d: %d kernels, busy %.3f ms, span %.3f..%.3f ms' % (
            q, len(ks), busy / 1e6, (ks[0][0] - t0) / 1e6, (ks[-1][1] - t0) / 1e6))
        gaps = []
        for a, b in zip(ks[:-1], ks[1:]):
            gaps.append((b[0] - a[1], (a[1] - t0) / 1e6, short(a[3]), short(b[3])))
        gaps.sort(reverse=True)
        for g, at, ka, kb in gaps[:4]:
            print('  gap %7.1f us at %.3f ms: %s -> %s' % (g / 1e3, at, ka, kb))
        agg = defaultdict(lambda: [0, 0])
        for s, e, _, n in ks:
            agg[short(n)][0] += e - s
            agg[short(n)][1] += 1
        for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:10]:
            print('  %8.1f us  x%-3d %s' % (t / 1e3, c, n))
        if dump:
            for s, e, _, n in ks:
                print('    %9.3f %8.1f %s' % ((s - t0) / 1e6, (e - s) / 1e3, short(n)))


if __name__ == '__main__':
    main()
