#!/bin/bash
# same-box A/B of two library builds under bench.py: ab_bench_lib.sh libA.so libB.so [bench flags]
A=$1; B=$2; shift 2
for i in 1 2; do
  for v in "$A" "$B"; do
    MDLM_HIP_LIB=$PWD/$v timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-policy-compare --no-eval --no-semi-ar --no-train --profile-stride 0 "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_denoise_step'])" || exit 1
  done
done
