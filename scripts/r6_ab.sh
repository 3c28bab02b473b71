#!/bin/bash
# A/B of library options on one box: bash scripts/r6_ab.sh "<bench args>" "ENV1=.. ENV2=.." "ENV.." ...
ARGS="$1"; shift
for E in "$@"; do
  env $E python bench.py $ARGS --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python scripts/r6_line.py "$E"
done
