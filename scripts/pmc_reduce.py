"""profiles/<round>_<tag>_pmc_traffic.json from three rocprofv3 --pmc passes over `python bench.py --roofline-child ...` (the serialised
roofline steps): per kernel of the LAST `nsteps` steps
  * matrix-pipe busy fraction = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs)   [pass 1]
    (GRBM_GUI_ACTIVE spans the dispatch, not just the kernel's waves: a 10 us kernel shows ~6 us of it around its duration,
    so the fraction UNDER-states short kernels; kernels under counter collection also run 5-25 % longer)
  * HBM bytes per launch = 2 x FETCH_SIZE (gfx950: /opt/skills/guides/MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes [2, 3]
  * `step`: the whole step as one number — sum over every kernel of the last nsteps steps of its HBM bytes, / nsteps
    (`hbm_bytes`), launches per step, and the kernels with the most bytes: what bench.py reports as
    `roofline.step.hbm_bytes_counter` for the same `workload` / `dtype` (round 6)
usage: python scripts/pmc_reduce.py <dir_mfma> <dir_fetch> <dir_write> <out.json> [nsteps [workload dtype]]
Layout: {'workload', 'dtype', 'step': {...}, 'kernels': {name: {...}}} (bench.py: pmc_traffic, pmc_step_traffic)."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace('void ', '')
    i = name.find('(')
    return (name[:i] if i > 0 else name).replace(' ', '')


def load(d, nsteps):
    """-> {kernel: {counter: [values of the dispatches in the last nsteps steps]}}, {kernel: [durations ns]}"""
    cc = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    kt = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    if not cc or not kt:
        return {}, {}
    trace = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Dispatch_Id']), short(r['Kernel_Name']))
                    for r in csv.DictReader(open(kt[0]))))
    ends = [i for i, r in enumerate(trace) if 'k_sgd_momentum' in r[3] or 'k_optimizer' in r[3]]
    keep = set()
    dur = collections.defaultdict(list)
    if len(ends) > nsteps:
        for s, e, did, n in trace[ends[-nsteps - 1] + 1:ends[-1] + 1]:
            keep.add(did)
            dur[n].append(e - s)
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        if int(r['Dispatch_Id']) in keep:
            vals[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return vals, dur


def main():
    d_m, d_f, d_w, out = sys.argv[1:5]
    nsteps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    workload = sys.argv[6] if len(sys.argv) > 6 else None
    dtype = sys.argv[7] if len(sys.argv) > 7 else None
    vm, dur = load(d_m, nsteps)
    vf, _ = load(d_f, nsteps)
    vw, _ = load(d_w, nsteps)
    kernels = {}
    for n in sorted(set(vm) | set(vf) | set(vw)):
        k = {}
        m = vm.get(n, {})
        if m.get('SQ_VALU_MFMA_BUSY_CYCLES') and m.get('GRBM_GUI_ACTIVE'):
            busy = sum(m['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024.0
            act = sum(m['GRBM_GUI_ACTIVE']) / 8.0
            k['mfma_busy_frac'] = busy / act if act else None
            k['gui_active_cycles_per_launch'] = act / len(m['GRBM_GUI_ACTIVE'])
            k['launches'] = len(m['GRBM_GUI_ACTIVE'])
            if dur.get(n):
                k['avg_us_under_counters'] = sum(dur[n]) / len(dur[n]) / 1e3
        f, w = vf.get(n, {}).get('FETCH_SIZE'), vw.get(n, {}).get('WRITE_SIZE')
        if f:
            k['fetch_bytes_per_launch'] = 2.0 * 1024.0 * sum(f) / len(f)
        if w:
            k['write_bytes_per_launch'] = 1024.0 * sum(w) / len(w)
        if f and w:
            k['hbm_bytes_per_launch'] = k['fetch_bytes_per_launch'] + k['write_bytes_per_launch']
            k.setdefault('launches', len(f))
        if k:
            kernels[n] = k
    # the step: bytes of EVERY dispatch of the last nsteps steps (kernels seen by both the FETCH and the WRITE pass)
    step_bytes, step_launches, by_kernel = 0.0, 0, {}
    for n in set(vf) & set(vw):
        f, w = vf[n].get('FETCH_SIZE') or [], vw[n].get('WRITE_SIZE') or []
        b = 2.0 * 1024.0 * sum(f) + 1024.0 * sum(w)
        step_bytes += b
        step_launches += len(f)
        by_kernel[n] = b / nsteps
    step = {'hbm_bytes': step_bytes / nsteps, 'launches': step_launches / float(nsteps), 'steps': nsteps,
            'top_kernels_bytes_per_step': dict(sorted(by_kernel.items(), key=lambda kv: -kv[1])[:12])}
    doc = {'workload': workload, 'dtype': dtype, 'step': step, 'note': 'rocprofv3 --pmc passes over the serialised roofline steps of bench.py (--roofline-child), last %d steps: '
                   'mfma_busy_frac = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs); HBM bytes = '
                   '2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, KiB -> bytes, mean per launch' % nsteps,
           'kernels': kernels}
    json.dump(doc, open(out, 'w'), indent=1, sort_keys=True)
    top = sorted(kernels.items(), key=lambda kv: -(kv[1].get('avg_us_under_counters', 0) * kv[1].get('launches', 0)))[:10]
    for n, k in top:
        print('%-44s x%-4d %7.1f us  mfma busy %s  fetch %s MB  write %s MB' % (
            n[:44], k.get('launches', 0), k.get('avg_us_under_counters', 0),
            '%.2f' % k['mfma_busy_frac'] if k.get('mfma_busy_frac') is not None else '-',
            '%.1f' % (k['fetch_bytes_per_launch'] / 1e6) if 'fetch_bytes_per_launch' in k else '-',
            '%.1f' % (k['write_bytes_per_launch'] / 1e6) if 'write_bytes_per_launch' in k else '-'))
    print('step: %.2f GB of HBM traffic, %.0f launches' % (step['hbm_bytes'] / 1e9, step['launches']))


if __name__ == '__main__':
    main()
