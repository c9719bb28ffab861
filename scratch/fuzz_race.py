"""Fast race against the reference race kernel (same Philox stream) over odd vocabulary sizes, row strides, dtypes
and peaked / flat logits."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdlm_amd import ops
torch.manual_seed(0)
bad = 0
for V in (30522, 50257, 50258, 1000, 129, 8191, 65, 4099):
  for dtype in (torch.bfloat16, torch.float32):
    for scale in (0.3, 3.0, 12.0):
      for mask in (V - 1, 103 % V):
        B, L = 4, 64
        dl = ops.alloc_logits(B, L, V, dtype)
        dl.copy_((torch.randn(B, L, V, device='cuda') * scale).to(dtype))
        x = torch.randint(0, V, (B, L), device='cuda'); x[x == mask] = 0
        x[torch.rand(B, L, device='cuda') < 0.7] = mask
        t = (torch.rand(B, device='cuda') * 0.95 + 0.02)
        dt = float(torch.rand(1)) * 0.02 + 1e-3
        lse = ops.subs_row_lse(dl, x, mask)
        seed, off = int(torch.randint(0, 2**31, (1,))), int(torch.randint(0, 2**40, (1,)))
        a = ops.subs_ddpm_update(dl, lse, x, t, dt, mask, seed=seed, offset=off, fast_math=False)
        b = ops.subs_ddpm_update(dl, lse, x, t, dt, mask, seed=seed, offset=off, fast_math=True)
        if not torch.equal(a, b):
          bad += 1
          print('MISMATCH', V, dtype, scale, mask, int((a != b).sum()), flush=True)
print('cases with a mismatch:', bad)
