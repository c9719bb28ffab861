#!/bin/bash
# regression sweep of the fuzzers and stress runs on the current tree (one GPU box, ~5 min)
set -o pipefail
O=gpurun_out/fuzz; mkdir -p $O
for s in fuzz_race fuzz_gemm fuzz_attn fuzz_semi_ar stress_variants stress_sequence stress_loss stress_train_shapes; do
  echo "== $s"
  timeout -k 10 400 python3 -u scratch/$s.py > $O/$s.log 2>&1 || { echo "FAILED $s"; tail -15 $O/$s.log; exit 1; }
  tail -2 $O/$s.log
done
