"""Vocabulary GEMM with / without the lse partials in its epilogue."""
import sys, torch
sys.path.insert(0, '.')
from mdlm_amd import ops
from mdlm_amd._lib import *
torch.manual_seed(0)
def bench(fn, iters=10):
  for _ in range(2): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3
D, V = 768, 50258
M = 22016
h = torch.randn(M, D, device='cuda').bfloat16()
Wv = (torch.randn(V, D, device='cuda') * 0.08).bfloat16(); bv = torch.randn(V, device='cuda') * 0.02
logits = torch.empty(M, ops.padded_vocab(V), dtype=torch.bfloat16, device='cuda')
part = torch.empty(ops.workspace_bytes('gemm_lse_partials', M, V) // 4, device='cuda')
lse = torch.empty(M, device='cuda')
a = bench(lambda: ops.gemm_bf16(h, Wv, logits, EPI_STORE, bias=bv, ldc=logits.stride(0)))
b = bench(lambda: ops.gemm_bf16(h, Wv, logits, EPI_STORE, bias=bv, ldc=logits.stride(0), lse_partials=part, lse_skip_col=V - 1))
c = bench(lambda: ops.lse_combine(part, M, V, lse))
print(f'vocab GEMM M={M}: plain {a:.1f} us, with partials {b:.1f} us (+{(b/a-1)*100:.1f} %), combine {c:.1f} us')
