#!/bin/bash
# Builds libluminoth_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result"
OBJS=""
pids=""
for f in api proposals detect targets roi loss optim elementwise ssd conv; do
  if [ ! -f "$f.o" ] || [ "$f.hip" -nt "$f.o" ] || [ lmh_common.h -nt "$f.o" ] || [ conv_common.h -nt "$f.o" ] || [ conv_generic.h -nt "$f.o" ] || [ ../../include/luminoth_hip.h -nt "$f.o" ]; then
    $HIPCC $FLAGS -c "$f.hip" -o "$f.o" &
    pids="$pids $!"
  fi
  OBJS="$OBJS $f.o"
done
for p in $pids; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -o libluminoth_hip.so
echo "built $(pwd)/libluminoth_hip.so"
