"""MLP-down as gate/residual GEMM + LN  vs  bf16-store GEMM + residual-LN."""
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
for n in (30, 36, 40, 42, 44, 48, 64):
  M = n * L
  h = torch.randn(M, D, device='cuda').bfloat16(); mh = torch.randn(M, 4 * D, device='cuda').bfloat16()
  x = torch.randn(M, D, device='cuda'); y = torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
  Wo = (torch.randn(D, D, device='cuda') * 0.02).bfloat16()
  W2 = (torch.randn(D, 4 * D, device='cuda') * 0.02).bfloat16(); b2 = torch.randn(D, device='cuda') * 0.02
  mod = torch.randn(1, 6 * D, device='cuda'); n1 = torch.ones(D, device='cuda')
  def old(A, W, b):
    ops.gemm_bf16(A, W, x, EPI_GATE_RESIDUAL, bias=b, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D)
    ops.ln_modulate(x, n1, mod[:, 0:D], mod[:, D:2 * D], h, L, 1, 6 * D, None)
  def new(A, W, b):
    ops.gemm_bf16(A, W, y, EPI_STORE, bias=b)
    ops.residual_ln_modulate(x, y, mod[:, :D], n1, mod[:, 0:D], mod[:, D:2 * D], h, L, 1, 6 * D)
  a = bench(lambda: old(mh, W2, b2)); b = bench(lambda: new(mh, W2, b2))
  c = bench(lambda: old(h, Wo, None)); d = bench(lambda: new(h, Wo, None))
  print(f'n={n}: fc2+LN {a:6.1f} -> {b:6.1f} us   out+LN {c:6.1f} -> {d:6.1f} us', flush=True)
