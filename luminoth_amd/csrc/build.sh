#!/bin/bash
# Builds libluminoth_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU) and libluminoth_io.so (host C:
# TFRecord framing + CRC32C for the dataset reader).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
if [ "${LMH_PROBES:-0}" != 0 ]; then PROBES="-DLMH_PROBES"; fi   # timing probes in the convolution kernels (scripts/r5_*sweep*, r5_epilogue_decomp)
FLAGS="--offload-arch=gfx950 $PROBES -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result"
OBJS=""
pids=""
# objects built with other flags (e.g. a probe build) are stale whatever their age
if [ "$(cat .build_flags 2>/dev/null)" != "$FLAGS" ]; then rm -f *.o; echo "$FLAGS" > .build_flags; fi
for f in api plan proposals detect targets roi loss optim elementwise bnorm ssd tail halfstore conv conv_x3; do
  stale=0
  if [ ! -f "$f.o" ] || [ "$f.hip" -nt "$f.o" ]; then stale=1; fi
  for h in *.h ../../include/luminoth_hip.h; do
    # (conv_x3.h holds the kernels of conv_x3.hip alone: no other translation unit includes it)
    if [ "$h" = conv_x3.h ] && [ "$f" != conv_x3 ]; then continue; fi
    if [ "$h" -nt "$f.o" ]; then stale=1; fi
  done
  if [ $stale = 1 ]; then
    $HIPCC $FLAGS -c "$f.hip" -o "$f.o" &
    pids="$pids $!"
    COMPILED="$COMPILED $f"
  else
    REUSED="$REUSED $f"
  fi
  OBJS="$OBJS $f.o"
done
for p in $pids; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -o libluminoth_hip.so
echo "built $(pwd)/libluminoth_hip.so  (compiled:${COMPILED:- none}; reused objects newer than their sources and every header:${REUSED:- none})"
if [ ! -f libluminoth_io.so ] || [ hostio.c -nt libluminoth_io.so ] || [ ../../include/luminoth_io.h -nt libluminoth_io.so ]; then
  ${CC:-gcc} -O2 -std=c11 -fPIC -shared -Wall -o libluminoth_io.so hostio.c
fi
echo "built $(pwd)/libluminoth_io.so"
