"""Gate/residual GEMM: residual add in registers (load, fma, store) against fp32 L2 atomics (tile | 32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdlm_amd import ops
from mdlm_amd._lib import EPI_GATE_RESIDUAL
torch.manual_seed(0)
D, L = 768, 1024
dev = 'cuda'
for n in (20, 40, 64):
  M = n * L
  x0 = torch.randn(M, D, device=dev)
  mod = torch.randn(1, 6 * D, device=dev) * 0.3
  for name, K in (('out', D), ('mlp2', 4 * D)):
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(D, K, device=dev) * 0.03).bfloat16(); b = torch.randn(D, device=dev) * 0.02
    outs = []
    for tile in (0, 32):
      x = x0.clone()
      ops.gemm_bf16(a, w, x, EPI_GATE_RESIDUAL, bias=b, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D, tile=tile)
      outs.append(x)
    d = (outs[0] - outs[1]).abs().max().item()
    xs = [x0.clone(), x0.clone()]
    def run(i):
      ops.gemm_bf16(a, w, xs[i], EPI_GATE_RESIDUAL, bias=b, gate=mod[:, :D], residual=xs[i], rows_per_mod=L, n_mod=1, mod_ld=6 * D, tile=(0, 32)[i])
    def tm(i, iters=20):
      e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(iters): run(i)
      e1.record(); torch.cuda.synchronize()
      return e0.elapsed_time(e1) / iters * 1e3
    tm(0, 3); tm(1, 3)
    res = [[], []]
    for rep in range(5):
      for i in (0, 1): res[i].append(tm(i))
    t0, t1 = sorted(res[0])[2], sorted(res[1])[2]
    print(f'n={n} {name:4s}: registers {t0:7.1f} us   atomics {t1:7.1f} us ({t1 / t0:.3f})   max |diff| {d:.3g}', flush=True)
