"""stream_groups=2 against 1: same tokens, same NFE (DiT-small, L=1024, V=50,258)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mdlm_amd.config import load_config
B, T = 16, 120
outs = {}
for name, ov in (('groups=1', []), ('groups=2', ['sampling.stream_groups=2']), ('groups=4', ['sampling.stream_groups=4'])):
  cfg = load_config(['model=small', f'loader.eval_batch_size={B}', f'sampling.steps={T}', *ov])
  model = bench.build_model(cfg, seed=1)
  model.seed_rng(123)
  for rep in range(2):
    model.seed_rng(123)
    t0 = time.perf_counter()
    x = model.restore_model_and_sample(num_steps=T)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
  outs[name] = (x.cpu(), int(model.last_nfe.item()))
  print(f'{name}: {dt:6.2f} s, NFE/sample {outs[name][1] / B:.1f}, identical to groups=1: '
        f'{torch.equal(outs["groups=1"][0], outs[name][0])}, nfe equal {outs["groups=1"][1] == outs[name][1]}', flush=True)
  del model
  torch.cuda.empty_cache()
