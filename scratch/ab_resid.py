"""GEMM with gate/residual epilogue (fp32 read-modify-write) vs plain bf16 store,
at loop-typical row counts."""
import sys, torch
sys.path.insert(0, '.')
from mdlm_amd import ops
from mdlm_amd._lib import *
torch.manual_seed(0)
def bench(fn, iters=30):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3
L, D = 1024, 768
for n in (21, 36, 40, 42, 43, 48, 64):
  M = n * L
  h = torch.randn(M, D, device='cuda').bfloat16()
  mh = torch.randn(M, 4 * D, device='cuda').bfloat16()
  x = torch.randn(M, D, device='cuda')
  y = torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
  Wo = (torch.randn(D, D, device='cuda') * 0.02).bfloat16()
  W2 = (torch.randn(D, 4 * D, device='cuda') * 0.02).bfloat16(); b2 = torch.randn(D, device='cuda') * 0.02
  mod = torch.randn(1, 6 * D, device='cuda')
  n1 = torch.ones(D, device='cuda')
  a = bench(lambda: ops.gemm_bf16(h, Wo, x, EPI_GATE_RESIDUAL, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D))
  b = bench(lambda: ops.gemm_bf16(h, Wo, y, EPI_STORE))
  c = bench(lambda: ops.gemm_bf16(mh, W2, x, EPI_GATE_RESIDUAL, bias=b2, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D))
  d = bench(lambda: ops.gemm_bf16(mh, W2, y, EPI_STORE, bias=b2))
  e = bench(lambda: ops.ln_modulate(x, n1, mod[:, 0:D], mod[:, D:2 * D], h, L, 1, 6 * D, None))
  print(f'n={n}: out gate/res {a:6.1f} us  bf16 store {b:6.1f} us | fc2 gate/res {c:6.1f}  bf16 store {d:6.1f} | ln {e:5.1f}', flush=True)
