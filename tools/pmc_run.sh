#!/bin/bash
# HBM traffic and SQ counters of the sampling loop, in separate rocprofv3 --pmc
# passes (one counter block per run, MI355X_MICROARCH.md), summarised into
# profiles/.  Run through gpurun from the repo root:  bash tools/pmc_run.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc
rm -rf $O; mkdir -p $O
ARGS="--steps 1 --warmup 0 --T 100 --no-eval --no-cpu-baseline --no-policy-compare --profile-stride 0 --no-semi-ar --no-train --set sampling.stream_groups=1"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/bench.py $ARGS > $O/f.log 2>&1 || { tail -5 $O/f.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/bench.py $ARGS > $O/w.log 2>&1 || { tail -5 $O/w.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $R/bench.py $ARGS > $O/sq.log 2>&1 || { tail -5 $O/sq.log; exit 1; }
# is there a counter that separates Infinity-Cache hits from HBM reads?  TCC_EA0_RDREQ_DRAM counts the
# requests routed to local memory (as opposed to GMI / IO), Infinity Cache included: recorded for the record
timeout -s KILL 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $O/dr -- python3 $R/bench.py $ARGS > $O/dr.log 2>&1 || { tail -5 $O/dr.log; exit 1; }
cd $R
python3 tools/pmc_dram.py $O/dr $O/pmc_dram.json > /dev/null
python3 tools/pmc_traffic.py $O/f $O/w $O/pmc_traffic.json > /dev/null && python3 tools/pmc_sq.py $O/sq $O/pmc_sq.json > /dev/null
# keep only the summaries
rm -rf $O/f $O/w $O/sq $O/dr
cat $O/pmc_traffic.json | head -60
