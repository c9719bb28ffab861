"""Does the leading dimension of the MLP intermediate (3072 bf16 = 48 lines of 128 B per row) cost the
MLP-down GEMM anything?  fc1 -> fc2 with mh rows 3072 / 3072+64 / 3072+128 elements apart."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdlm_amd import ops
from mdlm_amd._lib import EPI_BIAS_GELU, EPI_GATE_RESIDUAL
torch.manual_seed(0)
D, L = 768, 1024
dev = 'cuda'
for n in (40, 64):
  M = n * L
  x = torch.randn(M, D, device=dev)
  h = torch.randn(M, D, device=dev).bfloat16()
  w1 = (torch.randn(4 * D, D, device=dev) * 0.03).bfloat16(); b1 = torch.zeros(4 * D, device=dev)
  w2 = (torch.randn(D, 4 * D, device=dev) * 0.02).bfloat16(); b2 = torch.zeros(D, device=dev)
  mod = torch.randn(1, 6 * D, device=dev) * 0.1
  for pad in (64, 0, 128, 0, 64, 0):
    buf = torch.empty(M, 4 * D + pad, device=dev, dtype=torch.bfloat16)
    a = buf[:, :4 * D]
    def f1(): ops.gemm_bf16(h, w1, a, EPI_BIAS_GELU, bias=b1)
    def f2(): ops.gemm_bf16(a, w2, x, EPI_GATE_RESIDUAL, bias=b2, gate=mod[:, 2 * D:3 * D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D)
    def tm(fn, iters=20):
      e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
      for _ in range(3): fn()
      e0.record()
      for _ in range(iters): fn()
      e1.record(); torch.cuda.synchronize()
      return e0.elapsed_time(e1) / iters * 1e3
    t1 = sorted(tm(f1) for _ in range(3))[1]; t2 = sorted(tm(f2) for _ in range(3))[1]
    print(f'n={n} pad={pad:3d}: fc1 {t1:6.1f} us  fc2 {t2:6.1f} us', flush=True)
