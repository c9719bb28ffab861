"""Full-size trajectories (DiT-small, L=1024, V=50,258): token ids of the default path vs
(a) the reference race kernel (fast_math=false: must be identical),
(b) lse from the update kernel instead of the GEMM epilogue (identical up to exact mask-key ties),
(c) the reference's all-or-nothing cache policy (identical)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from mdlm_amd.config import load_config
B, T = 8, 200
outs = {}
for name, ov in (('default', []), ('race=reference', ['sampling.fast_math=false']),
                 ('lse in update kernel', ['sampling.lse_from_gemm=false']),
                 ('cache_policy=batch', ['sampling.cache_policy=batch'])):
  cfg = load_config(['model=small', f'loader.eval_batch_size={B}', f'sampling.steps={T}', *ov])
  model = bench.build_model(cfg, seed=1)
  model.seed_rng(123)
  t0 = time.perf_counter()
  x = model.restore_model_and_sample(num_steps=T)
  torch.cuda.synchronize()
  outs[name] = x.cpu()
  print(f'{name:24s}: {time.perf_counter() - t0:6.2f} s, NFE/sample {int(model.last_nfe.item()) / B:.1f}, '
        f'identical to default: {torch.equal(outs["default"], outs[name])}, '
        f'differing tokens {(outs["default"] != outs[name]).sum().item()}', flush=True)
  del model
  torch.cuda.empty_cache()
