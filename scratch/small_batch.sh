#!/bin/bash
# how launch-bound is the loop at small batch?  wall time per denoising step against the sum of the
# kernels' own time (HIP events of the profiling batch), B = 1, 4, 16
for b in 1 4 16; do
  timeout -k 10 200 python -u bench.py --batch $b --steps 1 --warmup 1 --no-cpu-baseline --no-policy-compare --no-eval --no-semi-ar --no-train --profile-stride 4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
T=1000
ks=sum(r['total_ms'] for r in d['rooflines'])*d['profile']['stride']/T
print('B=$b', 'samples/s', d['value'], 'wall ms/step', d['ms_per_denoise_step'], 'sum of kernel time ms/step', round(ks,3), flush=True)" || exit 1
done
