#!/bin/bash
# Round-end evidence from ONE box: GPU tests, smoke, default bench, training and
# semi-AR benches, kernel traces of the sampling and training workloads.
# Run through gpurun from the repo root:  bash tools/final_run.sh
# (the kernel trace of the sampling bench runs with sampling.stream_groups=1, like bench.py's own
# profiling batch: with the default two groups kernels of the two streams overlap and a kernel's
# duration in the trace is no longer its own)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gputests.log 2>&1 || { tail -30 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 200 python -u __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python -u bench.py > $O/bench_default.log 2>&1 || { tail -20 $O/bench_default.log; exit 1; }
tail -1 $O/bench_default.log | cut -c1-400
timeout -k 10 300 python -u bench.py --workload train > $O/bench_train.log 2>&1 || { tail -20 $O/bench_train.log; exit 1; }
tail -1 $O/bench_train.log
timeout -k 10 300 python -u bench.py --workload semi_ar --steps 1 --warmup 1 > $O/bench_semi_ar.log 2>&1 || { tail -20 $O/bench_semi_ar.log; exit 1; }
tail -1 $O/bench_semi_ar.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sample -o sample -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-policy-compare --no-eval --no-semi-ar --no-train --profile-stride 0 --set sampling.stream_groups=1 > $O/prof_sample.log 2>&1 || { tail -20 $O/prof_sample.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python3 $R/bench.py --workload train --steps 5 > $O/prof_train.log 2>&1 || { tail -20 $O/prof_train.log; exit 1; }
# keep only the summaries (the raw traces are large)
find $O -name "*kernel_trace.csv" -delete
ls -la $O $O/prof_sample $O/prof_train 2>/dev/null | head -40
