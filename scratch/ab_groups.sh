#!/bin/bash
# same-box A/B of stream-group configurations of the sampling loop (bench.py --set ...)
run() {
  timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-policy-compare --no-eval --no-semi-ar --no-train --profile-stride 0 $1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_denoise_step'], flush=True)" || exit 1
}
for v in "--set sampling.group_pipelines=false" "--set sampling.group_pipelines=true" \
         "--set sampling.stream_groups=3" "--set sampling.stream_groups=4" \
         "--set sampling.group_pipelines=false" "--set sampling.group_pipelines=true"; do
  run "$v" || exit 1
done
