"""Per-kernel time vs number of active samples (rows = n*1024, device row count)."""
import sys, torch
sys.path.insert(0, '.')
from mdlm_amd import ops
from mdlm_amd._lib import *
torch.manual_seed(0)
L, D, H = 1024, 768, 12
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
Wqkv = (torch.randn(3 * D, D, device='cuda') * 0.04).bfloat16()
Wo = (torch.randn(D, D, device='cuda') * 0.02).bfloat16()
W1 = (torch.randn(4 * D, D, device='cuda') * 0.02).bfloat16(); b1 = torch.randn(4 * D, device='cuda') * 0.02
W2 = (torch.randn(D, 4 * D, device='cuda') * 0.02).bfloat16(); b2 = torch.randn(D, device='cuda') * 0.02
qkv = torch.empty(3 * Mmax * D, device='cuda', dtype=torch.bfloat16)
ao = torch.empty(Mmax, D, device='cuda', dtype=torch.bfloat16)
nw = torch.randn(D, device='cuda')
mod = torch.randn(1, 6 * D, device='cuda')
cos = torch.randn(L, 32, device='cuda'); sin = torch.randn(L, 32, device='cuda')
nrows = torch.zeros(1, dtype=torch.int32, device='cuda'); nseq = torch.zeros(1, dtype=torch.int32, device='cuda')
print(' n    qkv    att    out   mlp1   mlp2     ln   block  us/sample')
lo, hi = (int(a) for a in sys.argv[1:3]) if len(sys.argv) > 2 else (28, 54)
for n in range(lo, hi + 1):
  nrows.fill_(n * L); nseq.fill_(n)
  M = Mmax
  t = [
    bench(lambda: ops.gemm_bf16(h, Wqkv, qkv, EPI_QKV_ROTARY, rows_per_mod=L, rot_cos=cos, rot_sin=sin, n_heads=H, m_rows_dev=nrows, ldc=0, M=M)),
    bench(lambda: ops.attn_fwd(qkv[:M * D], qkv[M * D:2 * M * D], qkv[2 * M * D:], ao, 64, H, L, n_seq_dev=nseq)),
    bench(lambda: ops.gemm_bf16(h, Wo, x, EPI_GATE_RESIDUAL, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D, m_rows_dev=nrows, M=M)),
    bench(lambda: ops.gemm_bf16(h, W1, mh, EPI_BIAS_GELU, bias=b1, m_rows_dev=nrows, M=M)),
    bench(lambda: ops.gemm_bf16(mh, W2, x, EPI_GATE_RESIDUAL, bias=b2, gate=mod[:, :D], residual=x, rows_per_mod=L, n_mod=1, mod_ld=6 * D, m_rows_dev=nrows, M=M)),
    bench(lambda: ops.ln_modulate(x, nw, mod[:, 0:D], mod[:, D:2 * D], h, L, 1, 6 * D, nrows)),
  ]
  blk = t[0] + t[1] + t[2] + t[3] + t[4] + 2 * t[5]
  print(f'{n:2d} ' + ' '.join(f'{v:6.1f}' for v in t) + f' {blk:7.1f} {blk / n:6.2f}', flush=True)
