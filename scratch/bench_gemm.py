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
  return e0.elapsed_time(e1) / iters
L = 1024
def run(M, mode):
  ops.gemm_set_tile(mode)
  D = 768
  h = torch.randn(M, D, device='cuda').bfloat16()
  mh = torch.randn(M, 4 * D, device='cuda').bfloat16()
  x = torch.randn(M, D, device='cuda')
  Wqkv = (torch.randn(3 * D, D, device='cuda') * 0.04).bfloat16()
  Wo = (torch.randn(D, D, device='cuda') * 0.02).bfloat16()
  W1 = (torch.randn(4 * D, D, device='cuda') * 0.02).bfloat16(); b1 = torch.randn(4 * D, device='cuda') * 0.02
  W2 = (torch.randn(D, 4 * D, device='cuda') * 0.02).bfloat16(); b2 = torch.randn(D, device='cuda') * 0.02
  V = 50258
  Wv = (torch.randn(V, D, device='cuda') * 0.08).bfloat16(); bv = torch.randn(V, device='cuda') * 0.02
  qkv = torch.empty(3 * M * D, device='cuda', dtype=torch.bfloat16)
  logits = ops.alloc_logits(M // L, L, V)
  mod = torch.randn(1, 6 * D, device='cuda')
  cos = torch.randn(L, 32, device='cuda'); sin = torch.randn(L, 32, device='cuda')
  res = {}
  res['qkv'] = (bench(lambda: ops.gemm_bf16(h, Wqkv, qkv, EPI_QKV_ROTARY, rows_per_mod=L, rot_cos=cos, rot_sin=sin, n_heads=12, ldc=0)), 2 * M * 3 * D * D)
  res['out'] = (bench(lambda: ops.gemm_bf16(h, Wo, x, EPI_GATE_RESIDUAL, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D)), 2 * M * D * D)
  res['mlp1'] = (bench(lambda: ops.gemm_bf16(h, W1, mh, EPI_BIAS_GELU, bias=b1)), 2 * M * 4 * D * D)
  res['mlp2'] = (bench(lambda: ops.gemm_bf16(mh, W2, x, EPI_GATE_RESIDUAL, bias=b2, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D)), 2 * M * 4 * D * D)
  res['vocab'] = (bench(lambda: ops.gemm_bf16(h, Wv, logits, EPI_STORE, bias=bv, rows_per_mod=L, ldc=logits.stride(-2)), iters=5), 2 * M * V * D)
  print(f'M={M} mode={mode}: ' + '  '.join(f'{k} {t*1e3:7.1f}us {f/t/1e9:6.0f}TF' for k, (t, f) in res.items()))
import sys as _s
modes = [int(a) for a in _s.argv[1:]] or [0, 16]
def check(M=4096):
  D = 768; V = 50258
  h = torch.randn(M, D, device='cuda').bfloat16(); mh = torch.randn(M, 4 * D, device='cuda').bfloat16()
  x0 = torch.randn(M, D, device='cuda')
  Wqkv = (torch.randn(3 * D, D, device='cuda') * 0.04).bfloat16()
  W2 = (torch.randn(D, 4 * D, device='cuda') * 0.02).bfloat16(); b2 = torch.randn(D, device='cuda') * 0.02
  W1 = (torch.randn(4 * D, D, device='cuda') * 0.02).bfloat16(); b1 = torch.randn(4 * D, device='cuda') * 0.02
  Wv = (torch.randn(V, D, device='cuda') * 0.08).bfloat16(); bv = torch.randn(V, device='cuda') * 0.02
  mod = torch.randn(M // L, 6 * D, device='cuda')
  cos = torch.randn(L, 32, device='cuda'); sin = torch.randn(L, 32, device='cuda')
  outs = {}
  for mode in modes:
    ops.gemm_set_tile(mode)
    qkv = torch.zeros(3 * M * D, device='cuda', dtype=torch.bfloat16)
    ops.gemm_bf16(h, Wqkv, qkv, EPI_QKV_ROTARY, rows_per_mod=L, rot_cos=cos, rot_sin=sin, n_heads=12, ldc=0)
    x = x0.clone()
    ops.gemm_bf16(mh, W2, x, EPI_GATE_RESIDUAL, bias=b2, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=M // L, mod_ld=6 * D)
    m1 = torch.zeros(M, 4 * D, device='cuda', dtype=torch.bfloat16)
    ops.gemm_bf16(h, W1, m1, EPI_BIAS_GELU, bias=b1)
    lg = ops.alloc_logits(M // L, L, V); lg.zero_()
    ops.gemm_bf16(h, Wv, lg, EPI_STORE, bias=bv, rows_per_mod=L, ldc=lg.stride(-2))
    outs[mode] = (qkv, x, m1, lg)
  a, b = outs[modes[0]], outs[modes[-1]]
  print('check maxdiff:', [float((p.float() - q.float()).abs().max()) for p, q in zip(a, b)])
import os
Ms = [int(m) for m in os.environ.get('MS', '65536').split(',')]
check()
for M in Ms:
  for mode in modes:
    run(M, mode)
