"""Update kernel: cost of the in-kernel lse recomputation for fresh rows."""
import sys, torch
sys.path.insert(0, '.')
from mdlm_amd import ops
torch.manual_seed(0)
B, L, V = 64, 1024, 50258
MASK = V - 1
buf = torch.empty(B, L, ops.padded_vocab(V), dtype=torch.bfloat16, device='cuda')
logits = buf[:, :, :V]
logits.copy_((torch.randn(B, L, V, device='cuda') * 4).bfloat16())
lse = torch.zeros(B * L, device='cuda')
t = torch.full((B,), 0.5, device='cuda')
xn = torch.empty(B, L, dtype=torch.int64, device='cuda')
ch = torch.zeros(B, dtype=torch.int32, device='cuda')
def bench(fn, iters=10):
  for _ in range(2): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters
for frac in (1.0, 0.55):
  x = torch.randint(0, V - 1, (B, L), device='cuda')
  x[torch.rand(B, L, device='cuda') < frac] = MASK
  for nf in (0, 40, 64):
    fresh = torch.zeros(B, dtype=torch.int32, device='cuda'); fresh[:nf] = 1
    ops.subs_row_lse(logits, x, MASK, out=lse)
    ms = bench(lambda: ops.subs_ddpm_update(logits, lse, x, t, 1e-3, MASK, seed=1, offset=0, changed=ch, x_next=xn, fresh=fresh, fast_math=True))
    print(f'masked {frac:.2f} fresh samples {nf:2d}: {ms:.3f} ms', flush=True)
