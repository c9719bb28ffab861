"""gate/residual GEMM time vs number of active samples, both tile configs."""
import sys, torch
sys.path.insert(0, '.')
from mdlm_amd import ops
from mdlm_amd._lib import *
torch.manual_seed(0)
L, D = 1024, 768
def bench(fn, iters=10):
  for _ in range(2): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3
Mmax = 64 * L
h = torch.randn(Mmax, D, device='cuda').bfloat16(); mh = torch.randn(Mmax, 4 * D, device='cuda').bfloat16()
x = torch.randn(Mmax, D, device='cuda')
Wo = (torch.randn(D, D, device='cuda') * 0.02).bfloat16()
W2 = (torch.randn(D, 4 * D, device='cuda') * 0.02).bfloat16(); b2 = torch.randn(D, device='cuda') * 0.02
mod = torch.randn(1, 6 * D, device='cuda')
nrows = torch.zeros(1, dtype=torch.int32, device='cuda')
print('n  out256 out128 mlp2_256 mlp2_128')
for n in list(range(16, 65, 2)):
  nrows.fill_(n * L)
  r = []
  for mode in (0, 1):
    ops.gemm_set_tile(mode)
    r.append(bench(lambda: ops.gemm_bf16(h, Wo, x, EPI_GATE_RESIDUAL, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D, m_rows_dev=nrows, M=Mmax)))
    r.append(bench(lambda: ops.gemm_bf16(mh, W2, x, EPI_GATE_RESIDUAL, bias=b2, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D, m_rows_dev=nrows, M=Mmax)))
  print(f'{n:2d} {r[0]:7.1f} {r[2]:7.1f} {r[1]:8.1f} {r[3]:8.1f}')
ops.gemm_set_tile(0)
