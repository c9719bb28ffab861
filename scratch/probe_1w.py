"""Ablation build: bf16-store GEMMs with the epilogue's global stores dropped (tile bit 2)
or the whole epilogue skipped (bit 1): how much of the kernel is store drain?"""
import os, sys, torch
os.environ['MDLM_HIP_LIB'] = os.path.abspath('scratch/ab/lib_abl.so')
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
for n in (40, 64):
  M = n * L
  h = torch.randn(M, D, device='cuda').bfloat16(); mh = torch.randn(M, 4 * D, device='cuda').bfloat16()
  x = torch.randn(M, D, device='cuda')
  qkv = torch.empty(3 * M * D, device='cuda', dtype=torch.bfloat16)
  Wqkv = (torch.randn(3 * D, D, device='cuda') * 0.04).bfloat16()
  Wo = (torch.randn(D, D, device='cuda') * 0.02).bfloat16()
  W1 = (torch.randn(4 * D, D, device='cuda') * 0.02).bfloat16(); b1 = torch.randn(4 * D, device='cuda') * 0.02
  W2 = (torch.randn(D, 4 * D, device='cuda') * 0.02).bfloat16(); b2 = torch.randn(D, device='cuda') * 0.02
  mod = torch.randn(1, 6 * D, device='cuda'); cos = torch.randn(L, 32, device='cuda'); sin = torch.randn(L, 32, device='cuda')
  for tile, name in ((0, 'full'), (2, 'no epilogue'), (16, '1w/SIMD 256x128'), (18, '1w/SIMD no epi')):
    f = bench(lambda: ops.gemm_bf16(h, W1, mh, EPI_BIAS_GELU, bias=b1, tile=tile))
    q = bench(lambda: ops.gemm_bf16(h, Wqkv, qkv, EPI_QKV_ROTARY, rows_per_mod=L, rot_cos=cos, rot_sin=sin, n_heads=12, ldc=0, tile=tile))
    a = bench(lambda: ops.gemm_bf16(h, Wo, x, EPI_GATE_RESIDUAL, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D, tile=tile))
    c = bench(lambda: ops.gemm_bf16(mh, W2, x, EPI_GATE_RESIDUAL, bias=b2, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D, tile=tile))
    print(f'n={n} {name:16s}: fc1 {f:7.1f}  qkv {q:7.1f}  out {a:7.1f}  fc2 {c:7.1f} us', flush=True)
