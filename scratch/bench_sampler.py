"""Roofline fractions of the sampler-side kernels at B=64, L=1024, V=50,258 (bf16
logits): algorithmic bytes / measured time vs 8 TB/s."""
import json, sys, torch
sys.path.insert(0, '.')
from mdlm_amd import ops
torch.manual_seed(0)
B, L, V = 64, 1024, 50258
MASK = V - 1
PEAK = 8000.0  # GB/s
def bench(fn, iters=5):
  for _ in range(2): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e-3
logits = ops.alloc_logits(B, L, V); logits.normal_(0, 2)
out = ops.alloc_logits(B, L, V)
res = {}
for frac in (1.0, 0.5):
  x = torch.randint(0, V - 1, (B, L), device='cuda'); x[torch.rand(B, L, device='cuda') < frac] = MASK
  if frac == 1.0: x[:] = MASK
  R = B * L; Rm = int((x == MASK).sum())
  tag = f'mask{int(frac*100)}'
  t = bench(lambda: ops.subs_logprobs(logits, x, MASK, out=out))
  res[f'subs_logprobs_{tag}'] = (t, Rm * V * 2 + R * V * 2 + 8 * R)
  lse = torch.zeros(R, device='cuda')
  t = bench(lambda: ops.subs_row_lse(logits, x, MASK, out=lse))
  res[f'subs_row_lse_{tag}'] = (t, Rm * V * 2 + 12 * R)
  xo = torch.empty_like(x)
  t = bench(lambda: ops.subs_argmax(logits, x, MASK, out=xo))
  res[f'subs_argmax_{tag}'] = (t, Rm * V * 2 + 16 * R)
  tt = torch.full((B,), 0.5, device='cuda')
  fresh = torch.zeros(B, dtype=torch.int32, device='cuda')
  ch = torch.zeros(B, dtype=torch.int32, device='cuda')
  t = bench(lambda: ops.subs_ddpm_update(logits, lse, x, tt, 1e-3, MASK, seed=1, offset=0, changed=ch, x_next=xo, fresh=fresh, fast_math=True))
  res[f'subs_ddpm_update_fast_{tag}'] = (t, Rm * V * 2 + 16 * R + 4 * B)
  t = bench(lambda: ops.subs_ddpm_update(logits, lse, x, tt, 1e-3, MASK, seed=1, offset=0, changed=ch, x_next=xo, fresh=fresh, fast_math=False), iters=2)
  res[f'subs_ddpm_update_exact_{tag}'] = (t, Rm * V * 2 + 16 * R + 4 * B)
  x0 = torch.randint(0, V - 1, (B, L), device='cuda')
  w = torch.ones(B, device='cuda')
  t = bench(lambda: ops.subs_nll(logits, x, x0, w, MASK))
  res[f'subs_nll_{tag}'] = (t, Rm * V * 2 + 24 * R)
x0 = torch.randint(0, V - 1, (B, L), device='cuda')
mc = torch.rand(B, 1, device='cuda')
t = bench(lambda: ops.q_xt(x0, mc, MASK, seed=1, offset=0), iters=50)
res['q_xt'] = (t, 16 * B * L + 4 * B)
D = 768
xx = torch.randn(B * L, D, device='cuda'); h = torch.empty(B * L, D, device='cuda', dtype=torch.bfloat16)
wln = torch.ones(D, device='cuda'); mod = torch.randn(1, 2 * D, device='cuda')
t = bench(lambda: ops.ln_modulate(xx, wln, mod[:, :D], mod[:, D:], h, L, 1, 2 * D), iters=50)
res['ln_modulate'] = (t, B * L * D * 6)
table = {k: {'us': round(t * 1e6, 1), 'algorithmic_GB': round(b / 1e9, 3), 'GBps': round(b / t / 1e9, 1),
             'frac_of_8TBps': round(b / t / 1e9 / PEAK, 3)} for k, (t, b) in res.items()}
for k, v in table.items(): print(f'{k:34s} {v["us"]:10.1f} us {v["algorithmic_GB"]:8.3f} GB {v["GBps"]:8.1f} GB/s  {v["frac_of_8TBps"]:.3f}')
json.dump(table, open('gpurun_out/r01_sampler_rooflines.json', 'w'), indent=1)
