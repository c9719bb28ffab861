"""Summarise a rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum pass per kernel tag: the share of
the L2's fabric-side read requests that are routed to local memory.  (The Infinity Cache sits on the memory side
of that route, so a share of 1.0 means the counter does NOT separate Infinity-Cache hits from HBM reads.)"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import TAGS


def main():
  d, out = sys.argv[1:3]
  acc = collections.defaultdict(lambda: collections.defaultdict(float))
  for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
      for tag, pat in TAGS.items():
        if pat in r['Kernel_Name']:
          acc[tag][r['Counter_Name']] += float(r['Counter_Value'])
  res = {}
  for tag, cs in acc.items():
    tot, dram = cs.get('TCC_EA0_RDREQ_sum', 0.0), cs.get('TCC_EA0_RDREQ_DRAM_sum', 0.0)
    res[tag] = {'rdreq': round(tot), 'rdreq_dram': round(dram), 'dram_share': round(dram / tot, 4) if tot else None}
  json.dump(res, open(out, 'w'), indent=1)
  print(json.dumps(res))


if __name__ == '__main__':
  main()
