"""Host cost of one denoising step: the Python loop with the C-ABI calls replaced by a counter
(so the GPU queue never fills), stream_groups = 1 / 2 (B = 64, L = 1024)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mdlm_amd import ops
from mdlm_amd.config import load_config
B, T = 64, 200
for g in (1, 2):
  cfg = load_config(['model=small', f'loader.eval_batch_size={B}', f'sampling.steps={T}', f'sampling.stream_groups={g}'])
  model = bench.build_model(cfg, seed=1)
  model.seed_rng(123)
  model.restore_model_and_sample(num_steps=8); torch.cuda.synchronize()
  real = ops.call
  n = [0]
  def fake(name, *a):
    n[0] += 1
    return 0
  ops.call = fake
  t0 = time.perf_counter()
  model.restore_model_and_sample(num_steps=T)
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  ops.call = real
  print(f'groups={g}: {n[0] / (T + 1):.0f} C-ABI calls per step, host {1e3 * (t1 - t0) / (T + 1):.2f} ms/step without the calls themselves', flush=True)
  # the calls themselves: an argument-checked no-op launch (n_seq = 0 returns before the launch)
  q = torch.empty(16, device='cuda')
  t0 = time.perf_counter()
  for _ in range(20000):
    real('mdlm_attn_fwd', q.data_ptr(), q.data_ptr(), q.data_ptr(), q.data_ptr(), 0, 12, 1024, 0, 0)
  print(f'  ctypes round trip {1e6 * (time.perf_counter() - t0) / 20000:.2f} us per call', flush=True)
  del model; torch.cuda.empty_cache()
