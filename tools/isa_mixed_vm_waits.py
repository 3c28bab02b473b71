"""Static ISA check (round 4): places where a kernel waits for `vmcnt(0)` between a store and a later load.

On gfx950 loads and stores count on the same vmcnt and retire out of order with respect to each other, so with both kinds
in flight the compiler can only wait for vmcnt(0): a load issued behind a store whose result is needed behind ANOTHER store
costs a full store + load round trip each time (conv_fast.h / conv_hs.h epilogues before round 4).  Usage:

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only file.hip -o file.s
    python tools/isa_mixed_vm_waits.py file.s [...]

Prints, per kernel, the number of store -> s_waitcnt vmcnt(0) -> load sequences (within 80 instructions of each other)."""
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
            funcs[name].append(ln.strip())
    out = []
    for k, ins in funcs.items():
        hits, last_store, wait = 0, None, None
        for i, l in enumerate(ins):
            if l.startswith(('global_store', 'buffer_store', 'flat_store')):
                last_store, wait = i, None
            elif 's_waitcnt' in l and 'vmcnt(0)' in l and last_store is not None and i - last_store < 80:
                wait = i
            elif l.startswith(('global_load', 'buffer_load', 'flat_load')) and wait is not None and i - wait < 80:
                hits += 1
                last_store = wait = None
        if hits:
            out.append((hits, demangle(k)))
    return out


if __name__ == '__main__':
    for f in sys.argv[1:]:
        for hits, k in sorted(scan(f), reverse=True):
            print('%-28s %3d  %s' % (f.split('/')[-1], hits, k[:120]))
