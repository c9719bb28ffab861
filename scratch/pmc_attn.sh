#!/bin/bash
export MDLM_HIP_LIB=$PWD/$1
O=gpurun_out/pmc_attn_$2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$O -- python3 $GRAFT_REPO_ROOT/scratch/attn_only.py > $GRAFT_REPO_ROOT/$O/log.txt 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/log.txt; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $GRAFT_REPO_ROOT/$O/b -- python3 $GRAFT_REPO_ROOT/scratch/attn_only.py > $GRAFT_REPO_ROOT/$O/log2.txt 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/log2.txt; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('$O/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    if 'attn_fwd' in r['Kernel_Name']:
      a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in sorted(acc.items()): print(f'{k:32s} {v/n:16.0f}  ({n} launches)')
PY
