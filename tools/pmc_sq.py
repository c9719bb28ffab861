"""Summarise a rocprofv3 --pmc pass of SQ / GRBM counters per kernel tag.

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU \\
            SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d out -- python3 bench.py ...
  python3 tools/pmc_sq.py out profiles/pmc_sq.json

Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over
SIMDs; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles per wave;
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  mfma_util = MFMA busy cycles /
(1024 SIMDs x GRBM_GUI_ACTIVE / 8)."""
import collections
import csv
import glob
import json
import os
import sys

from pmc_traffic import TAGS


def main():
  d, out = sys.argv[1:3]
  acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
  for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
      for tag, pat in TAGS.items():
        if pat in r['Kernel_Name']:
          a = acc[tag][r['Counter_Name']]
          a[0] += float(r['Counter_Value']); a[1] += 1
  res = {}
  for tag, cs in acc.items():
    m = {k: v[0] / v[1] for k, v in cs.items()}
    e = {'launches': max(v[1] for v in cs.values()), 'per_launch': {k: round(v) for k, v in m.items()}}
    if 'GRBM_GUI_ACTIVE' in m and m['GRBM_GUI_ACTIVE'] > 0:
      clk = m['GRBM_GUI_ACTIVE'] / 8.0
      if 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        e['mfma_util'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * clk), 4)
      if 'SQ_ACTIVE_INST_VALU' in m:
        # quad-cycles of VALU issue summed over waves / (SIMD-cycles / 4)
        e['valu_issue_util'] = round(m['SQ_ACTIVE_INST_VALU'] * 4.0 / (1024.0 * clk), 4)
    res[tag] = e
  json.dump(res, open(out, 'w'), indent=1)
  print(json.dumps(res))


if __name__ == '__main__':
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  main()
