import sys, torch
sys.path.insert(0, '.')
from mdlm_amd import ops
torch.manual_seed(0)
def bench(fn, iters=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters
H, L = 12, 1024
for B in (40, 64):
  q = torch.randn(B, H, L, 64, device='cuda').bfloat16(); k = torch.randn_like(q); vt = torch.randn(B, H, 64, L, device='cuda').bfloat16()
  out = torch.empty(B * L, H * 64, device='cuda', dtype=torch.bfloat16)
  t = bench(lambda: ops.attn_fwd(q, k, vt, out, B, H, L))
  print(f'attn B={B}: {t*1e3:.1f} us  {4*L*L*64*H*B/t/1e9:.0f} TF')
