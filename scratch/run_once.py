# one timed ddpm_cache pass (B=64, T=1000) with whatever MDLM_HIP_LIB points to
import sys, time, torch
sys.path.insert(0, '.')
from mdlm_amd.config import load_config
from bench import build_model
cfg = load_config(['model=small', 'loader.eval_batch_size=64', 'sampling.steps=1000', 'sampling.fast_math=true'])
model = build_model(cfg, seed=1)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for rep in range(reps):
  model.seed_rng(1)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  out = model.restore_model_and_sample(num_steps=1000)
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  print(f'{64/dt:.4f} samples/s  nfe {int(model.last_nfe.item())/64:.1f}  checksum {int(out.sum().item())}', flush=True)
