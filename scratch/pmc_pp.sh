#!/bin/bash
# SQ counters of the two-accumulator-set GEMM against the 256x256 kernel (gate/residual shapes,
# 64 samples): one rocprofv3 --pmc pass over scratch/ab_pp.py, summarised per kernel symbol.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_pp
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MODE=16 timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $R/scratch/ab_pp.py $R/mdlm_amd/libmdlm_hip.so 64 > $O/sq.log 2>&1 || { tail -5 $O/sq.log; exit 1; }
cd $R
python3 - $O/sq $O/pmc_pp.json <<'PY'
import collections, csv, glob, json, os, sys
d, out = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'gemm' not in k:
      continue
    # the two shapes of a symbol differ by their grid's work: key on (symbol, LDS size is equal) -> use duration-free split by Dispatch order parity is fragile; keep one entry per symbol
    name = 'pp' if 'gemm_pp_kernel' in k else 'base256'
    a = acc[name][r['Counter_Name']]
    a[0] += float(r['Counter_Value']); a[1] += 1
res = {}
for name, cs in acc.items():
  m = {k: v[0] / v[1] for k, v in cs.items()}
  clk = m.get('GRBM_GUI_ACTIVE', 0) / 8.0
  e = {'launches': max(v[1] for v in cs.values()), 'per_launch': {k: round(v) for k, v in m.items()}}
  if clk:
    e['mfma_util'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * clk), 4)
    e['wave_cycles_waiting_frac'] = round(m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 4)
    e['wave_cycles_issue_stalled_frac'] = round(m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], 4)
    e['wave_cycles_issuing_frac'] = round(m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES'], 4)
  res[name] = e
res['_note'] = ('rocprofv3 --pmc (one pass, 8 SQ + GRBM counters) over scratch/ab_pp.py at 64 samples: gate/residual GEMMs '
                '(out-projection K=768 and MLP-down K=3072 launches averaged together per kernel symbol); '
                'SQ_WAIT_ANY = wave parked at s_waitcnt / s_barrier, SQ_WAIT_INST_ANY = issue stall, quad-cycles per wave')
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $O/sq
