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
L = 1024; M = 65536; D = 768
h = torch.randn(M, D, device='cuda').bfloat16()
x = torch.randn(M, D, device='cuda'); y = torch.empty_like(x)
Wo = (torch.randn(D, D, device='cuda') * 0.02).bfloat16()
mod = torch.randn(1, 6 * D, device='cuda')
xb = torch.empty(M, D, device='cuda', dtype=torch.bfloat16)
for mode in [int(a) for a in sys.argv[1:]] or [0, 8]:
  ops.gemm_set_tile(mode)
  t_in = bench(lambda: ops.gemm_bf16(h, Wo, x, EPI_GATE_RESIDUAL, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D))
  t_out = bench(lambda: ops.gemm_bf16(h, Wo, y, EPI_GATE_RESIDUAL, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D))
  t_st = bench(lambda: ops.gemm_bf16(h, Wo, y, EPI_STORE))
  t_stb = bench(lambda: ops.gemm_bf16(h, Wo, xb, EPI_STORE))
  print(f'mode {mode}: inplace {t_in:.1f}us  out-of-place {t_out:.1f}us  plain fp32 store {t_st:.1f}us  plain bf16 store {t_stb:.1f}us')
t = bench(lambda: x.add_(1.0)); print(f'torch x+=1: {t:.1f}us  {2*x.numel()*4/t/1e6:.2f} TB/s')
t = bench(lambda: torch.add(x, 1.0, out=y)); print(f'torch y=x+1: {t:.1f}us  {2*x.numel()*4/t/1e6:.2f} TB/s')
t = bench(lambda: y.copy_(x)); print(f'torch copy: {t:.1f}us  {2*x.numel()*4/t/1e6:.2f} TB/s')
t = bench(lambda: y.fill_(1.0)); print(f'torch fill: {t:.1f}us  {x.numel()*4/t/1e6:.2f} TB/s')
