"""Gate/residual GEMMs: 256x256 vs 128x128 tile configuration per active-sample count."""
import sys, torch
sys.path.insert(0, '.')
from mdlm_amd import ops
from mdlm_amd._lib import *
torch.manual_seed(0)
def bench(fn, iters=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3
L, D = 1024, 768
M = 64 * L
h = torch.randn(M, D, device='cuda').bfloat16()
mh = torch.randn(M, 4 * D, device='cuda').bfloat16()
x = torch.randn(M, D, device='cuda')
Wo = (torch.randn(D, D, device='cuda') * 0.02).bfloat16()
W2 = (torch.randn(D, 4 * D, device='cuda') * 0.02).bfloat16(); b2 = torch.randn(D, device='cuda') * 0.02
W1 = (torch.randn(4 * D, D, device='cuda') * 0.02).bfloat16(); b1 = torch.randn(4 * D, device='cuda') * 0.02
mod = torch.randn(1, 6 * D, device='cuda')
tot = {0: 0.0, 1: 0.0, 'best': 0.0}
for n in range(30, 53):
  live = torch.tensor([n * L], dtype=torch.int32, device='cuda')
  r = {}
  for tile in (0, 1):
    a = bench(lambda: ops.gemm_bf16(h, Wo, x, EPI_GATE_RESIDUAL, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D, m_rows_dev=live, tile=tile))
    c = bench(lambda: ops.gemm_bf16(mh, W2, x, EPI_GATE_RESIDUAL, bias=b2, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D, m_rows_dev=live, tile=tile))
    f = bench(lambda: ops.gemm_bf16(h, W1, mh, EPI_BIAS_GELU, bias=b1, m_rows_dev=live, tile=tile))
    r[tile] = (a, c, f)
  print(f'n={n}: out {r[0][0]:6.1f} / {r[1][0]:6.1f}   fc2 {r[0][1]:6.1f} / {r[1][1]:6.1f}   fc1 {r[0][2]:6.1f} / {r[1][2]:6.1f}  (256 / 128 tile, us)', flush=True)
