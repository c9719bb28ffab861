#!/bin/bash
# same-box A/B of two bench configurations: ab_bench.sh "<flags A>" "<flags B>"
for i in 1 2; do
  for v in "$1" "$2"; do
    timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-policy-compare --no-eval --no-semi-ar --no-train --profile-stride 0 $v 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_denoise_step'])" || exit 1
  done
done
