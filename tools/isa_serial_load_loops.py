"""Static ISA check (round 4): loops whose global loads are each awaited inside the loop (load -> s_waitcnt vmcnt(0) ->
use -> branch back): every trip is a full round trip of memory latency.  Found this way: the stem's weight fill (40 trips),
the ROI-pooling slab fill (8), the sort's chunk load (8), k_rpn_target_subsample's label read (48).

    python tools/isa_serial_load_loops.py file.s [...]      (file.s from hipcc -S --cuda-device-only)

Prints kernel, loop label, instructions in the loop body, loads, vmcnt waits."""
import re
import subprocess
import sys


def demangle(name):
    try:
        return subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


def scan(path):
    funcs, name = {}, None
    for ln in open(path):
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            name = m.group(1)
            funcs[name] = []
        elif name and ln.startswith('\t.size'):
            name = None
        elif name:
            funcs[name].append(ln.rstrip('\n'))
    out = []
    for k, ins in funcs.items():
        labels = {}
        for i, l in enumerate(ins):
            m = re.match(r'^(\.LBB\w+):', l)
            if m:
                labels[m.group(1)] = i
        for i, l in enumerate(ins):
            m = re.match(r'^\s+s_cbranch_\w+\s+(\.LBB\w+)', l) or re.match(r'^\s+s_branch\s+(\.LBB\w+)', l)
            if not m or m.group(1) not in labels:
                continue
            start = labels[m.group(1)]
            if start >= i or i - start > 400:
                continue
            body = [x.strip() for x in ins[start:i]]
            if any(re.match(r'^\.LBB\w+:', x) for x in body[1:]) and i - start > 120:
                continue                                    # large multi-block region: not an inner loop
            loads = sum(1 for x in body if x.startswith(('global_load', 'buffer_load', 'flat_load')) and 'lds' not in x)
            waits0 = sum(1 for x in body if 's_waitcnt' in x and 'vmcnt(0)' in x)
            if loads and waits0 and loads <= 2 * waits0:
                out.append((demangle(k).split('(')[0][:70], m.group(1), i - start, loads, waits0))
    return out


if __name__ == '__main__':
    for f in sys.argv[1:]:
        for row in scan(f):
            print('%-22s %-70s %-12s %4d instr %2d loads %2d vmcnt(0)' % ((f.split('/')[-1],) + row))
