"""update kernel alone on near-uniform and peaked logits (bench.update_kernel_block)"""
import json, sys, torch
sys.path.insert(0, '.')
import bench
from mdlm_amd.config import load_config
cfg = load_config(['model=small', 'loader.eval_batch_size=64', 'sampling.steps=1000'])
model = bench.build_model(cfg, seed=1)
print(json.dumps(bench.update_kernel_block(model, 64, cfg.model.length, model.vocab_size), indent=1))
