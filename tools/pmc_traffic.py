"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per
launch for the kernels bench.py reports a roofline for.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/f -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/w -- python3 bench.py ...
  python3 tools/pmc_traffic.py out/f out/w profiles/pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on
gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced streaming
reads at 64 bytes, i.e. reports exactly half of the bytes -> doubled here;
WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections
import csv
import glob
import json
import os
import sys

TAGS = {'gate_residual': 'gemm_bf16_kernel<2,', 'vocab': 'gemm_bf16_kernel<0, unsigned short',
        'mlp_up_gelu': 'gemm_bf16_kernel<1,', 'qkv_rotary': 'gemm_bf16_kernel<3,',
        'attention': 'attn_fwd_kernel', 'ddpm_update': 'ddpm_update_fast_kernel',
        'ln_modulate': 'ln_modulate_kernel'}


def collect(d, counter):
  acc = collections.defaultdict(lambda: [0.0, 0])
  for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
      if r['Counter_Name'] != counter:
        continue
      for tag, pat in TAGS.items():
        if pat in r['Kernel_Name']:
          acc[tag][0] += float(r['Counter_Value'])
          acc[tag][1] += 1
  return acc


def main():
  fd, wd, out = sys.argv[1:4]
  fe, wr = collect(fd, 'FETCH_SIZE'), collect(wd, 'WRITE_SIZE')
  res = {}
  for tag in TAGS:
    if fe[tag][1] and wr[tag][1]:
      fetch = 2.0 * 1024.0 * fe[tag][0] / fe[tag][1]   # gfx950: x2
      write = 1024.0 * wr[tag][0] / wr[tag][1]
      res[tag] = {'hbm_bytes_per_launch': round(fetch + write), 'fetch_bytes': round(fetch),
                  'write_bytes': round(write), 'launches_profiled': fe[tag][1],
                  'note': 'FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB -> bytes'}
  json.dump(res, open(out, 'w'), indent=1)
  print(json.dumps(res))


if __name__ == '__main__':
  main()
