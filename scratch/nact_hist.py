import sys, torch, collections
sys.path.insert(0, '.')
from mdlm_amd.config import load_config
from bench import build_model
cfg = load_config(['model=small', 'loader.eval_batch_size=64', 'sampling.steps=1000', 'sampling.fast_math=true'])
model = build_model(cfg, seed=1)
model.seed_rng(1)
hist = collections.Counter()
orig = model._refresh_cache
def wrapped(st, x, sigma=None, want_lse=False):
  orig(st, x, sigma, want_lse)
  hist[int(st['n_active'].item())] += 1
model._refresh_cache = wrapped
model.restore_model_and_sample(num_steps=1000)
print(sorted(hist.items()))
