#!/bin/bash
# in-loop launch time of the update kernel (bench.py's profiling batch) for two library builds, same box
for v in "$@"; do
  MDLM_HIP_LIB=$PWD/$v timeout -k 10 300 python -u bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-policy-compare --no-eval --no-semi-ar --no-train --profile-stride 4 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
u=[r for r in d['rooflines'] if r['tag']=='update'][0]
print('$v', d['value'], 'update avg ms', u['avg_launch_ms'], 'rows', u['avg_rows_per_launch'])" || exit 1
done
