#!/bin/bash
# GPU clock and power while the sampling loop runs (rocm-smi polled beside a short bench run)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/clock.txt; : > $O
timeout -k 10 200 python -u $R/bench.py --steps 1 --warmup 0 --T 1000 --no-cpu-baseline --no-policy-compare --no-eval --profile-stride 0 > $R/gpurun_out/clock_bench.log 2>&1 &
BP=$!
sleep 7
for i in $(seq 1 16); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|fclk|Power \(W\)|Socket" >> $O
  echo "--" >> $O
  sleep 0.5
done
wait $BP
tail -1 $R/gpurun_out/clock_bench.log | cut -c1-200
cat $O | sort | uniq -c | sort -rn | head -20
