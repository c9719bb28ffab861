"""How many 64-column groups of a masked row could the update kernel skip outright?  A column can only win the
race against the mask column if p_v * dt / g_min > key_mask = (t - dt) / g_mask, g_min = -ln(1 - 2^-24) ~ 6e-8."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mdlm_amd.config import load_config
cfg = load_config(['model=small', 'loader.eval_batch_size=2', 'sampling.steps=1000'])
model = bench.build_model(cfg, seed=1)
V = model.vocab_size; mask = model.mask_index
T = 1000; dt = (1 - 1e-5) / T
gmin = 5.97e-8
torch.manual_seed(0)
for frac in (1.0, 0.5, 0.1):
  x = torch.randint(0, V - 1, (2, 1024), device='cuda')
  m = torch.rand(2, 1024, device='cuda') < frac
  x[m] = mask
  logits = model.backbone(x, torch.zeros(2, device='cuda')).float()
  logits[..., mask] = -1e6
  lp = torch.log_softmax(logits, -1)[m]                      # masked rows
  pad = (64 - V % 64) % 64
  gm = torch.nn.functional.pad(lp, (0, pad), value=-1e9).view(lp.shape[0], -1, 64).max(-1).values   # log p group max
  t = frac   # masked fraction ~ t
  g_mask = torch.distributions.Exponential(1.0).sample((lp.shape[0], 1)).cuda()
  key_mask = (t - dt) / g_mask
  thr = torch.log(key_mask * gmin / dt)                      # log p threshold
  skip = (gm < thr).float().mean().item()
  rows_all = (gm < thr).all(-1).float().mean().item()
  print(f'masked fraction {frac}: row max p {lp.max(-1).values.exp().mean().item():.2e}, logit std {logits[m][:, :-1].std().item():.2f}; '
        f'groups skippable {100 * skip:.1f} %, whole rows {100 * rows_all:.1f} %', flush=True)
