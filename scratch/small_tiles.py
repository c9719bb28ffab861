"""Small launches (a few samples): 256x256 against 128x128 tiles for the four GEMMs of a block -- time and
bit-identity (same K order per output element, so the outputs must be equal).
  python3 scratch/small_tiles.py [n_samples ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdlm_amd import ops
from mdlm_amd._lib import EPI_QKV_ROTARY, EPI_GATE_RESIDUAL, EPI_BIAS_GELU
ns = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 6, 8, 12, 16]
torch.manual_seed(0)
L, D, H = 1024, 768, 12
dev = 'cuda'
def tm(fn, iters=30):
  for _ in range(3): fn()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3
inv = 1.0 / (10000 ** (torch.arange(0, 64, 2, device=dev).float() / 64))
ang = torch.arange(L, device=dev).float()[:, None] * inv[None]
cos, sin = ang.cos().bfloat16().float().contiguous(), ang.sin().bfloat16().float().contiguous()
Wq = (torch.randn(3 * D, D, device=dev) * 0.04).bfloat16(); Wo = (torch.randn(D, D, device=dev) * 0.02).bfloat16()
W1 = (torch.randn(4 * D, D, device=dev) * 0.02).bfloat16(); b1 = torch.randn(4 * D, device=dev) * 0.02
W2 = (torch.randn(D, 4 * D, device=dev) * 0.02).bfloat16(); b2 = torch.randn(D, device=dev) * 0.02
mod = torch.randn(1, 6 * D, device=dev)
for n in ns:
  M = n * L
  h = torch.randn(M, D, device=dev).bfloat16(); mh = torch.randn(M, 4 * D, device=dev).bfloat16()
  x = torch.randn(M, D, device=dev)
  def qkv(t, out): ops.gemm_bf16(h, Wq, out, EPI_QKV_ROTARY, rows_per_mod=L, rot_cos=cos, rot_sin=sin, n_heads=H, ldc=0, M=M, tile=t)
  def outp(t, out): ops.gemm_bf16(h, Wo, out, EPI_GATE_RESIDUAL, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D, M=M, tile=t)
  def fc1(t, out): ops.gemm_bf16(h, W1, out, EPI_BIAS_GELU, bias=b1, M=M, tile=t)
  def fc2(t, out): ops.gemm_bf16(mh, W2, out, EPI_GATE_RESIDUAL, bias=b2, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D, M=M, tile=t)
  line = [f'n={n:2d}']
  for name, f, mk in (('qkv', qkv, lambda: torch.zeros(3 * M * D, device=dev, dtype=torch.bfloat16)),
                      ('out', outp, lambda: torch.zeros(M, D, device=dev)),
                      ('fc1', fc1, lambda: torch.zeros(M, 4 * D, device=dev, dtype=torch.bfloat16)),
                      ('fc2', fc2, lambda: torch.zeros(M, D, device=dev))):
    o0, o1 = mk(), mk()
    f(0, o0); f(1, o1); torch.cuda.synchronize()
    same = torch.equal(o0, o1)
    a, b = tm(lambda: f(0, o0)), tm(lambda: f(1, o1))
    line.append(f'{name} {a:6.1f} -> {b:6.1f} us ({b / a:.2f}{"" if same else " DIFFERENT"})')
  print('  '.join(line), flush=True)
