"""NELBO eval only (B = 64, L = 1024), for a kernel trace: python3 scratch/eval_only.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mdlm_amd.config import load_config
B, L = 64, 1024
cfg = load_config(['model=small', f'loader.eval_batch_size={B}'])
model = bench.build_model(cfg, seed=1)
V = model.vocab_size
g = torch.Generator(device='cuda').manual_seed(1)
ids = torch.randint(0, V - 1, (B, L), generator=g, device='cuda'); ids[:, 0] = 50256; ids[:, -1] = 50256
batch = {'input_ids': ids, 'attention_mask': torch.ones(B, L, device='cuda')}
model.on_validation_epoch_start()
model.validation_step(batch, 0); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(8): model.validation_step(batch, i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'{8 * B * L / dt / 1e6:.3f} M tokens/s, {dt / 8 * 1e3:.2f} ms per batch', flush=True)
