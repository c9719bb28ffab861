"""Launcher with the reference's command-line shape (main.py:186-201):

  python main.py mode=sample_eval sampling.predictor=ddpm_cache sampling.steps=1000 \
      loader.eval_batch_size=64 model.length=1024 [eval.checkpoint_path=...] \
      [sampling.semi_ar=True sampling.stride_length=512 sampling.num_strides=2]
  python main.py mode=ppl_eval loader.eval_batch_size=16 eval.num_batches=8

Multi-GPU: `python -m torch.distributed.run --nproc-per-node N main.py mode=...`
(one process per GPU; samples / eval sequences are sharded by rank, the only
collective is the fp64 NELBO pair all-reduce).

  python main.py mode=train data=lm1b model.length=128 loader.global_batch_size=512 \
      trainer.max_steps=100        (scripts/train_lm1b_d3pm.sh recipe with MDLM defaults)

`mode=train` runs mdlm_amd.train.Trainer: hand-written backward
kernels, flat AdamW/EMA buffers and the DDP gradient all-reduce on RCCL, on
synthetic LM1B/OWT-shaped batches (the datasets need the network).
"""
import json
import os
import sys
import time

import torch

from mdlm_amd import dist as mdist
from mdlm_amd.config import load_config
from mdlm_amd.tokenizer import get_tokenizer


def _load_backbone_weights(model, path):
  """Reads a reference Lightning `.ckpt` (state_dict['backbone.*'], EMA shadow
  params under ckpt['ema'], diffusion.py:168-181) or a plain backbone
  state-dict file, with a loader that executes nothing from the file."""
  if path.endswith('.safetensors'):
    from safetensors.torch import load_file
    sd = load_file(path)
    ckpt = {'state_dict': sd}
  else:
    ckpt = torch.load(path, map_location='cpu', weights_only=True)
  sd = ckpt.get('state_dict', ckpt)
  bb = {k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')} or sd
  model.backbone.load_state_dict(bb)
  model.on_load_checkpoint(ckpt)


def _synthetic_weights(model, seed):
  """Seeded N(0, s) weights (the reference zero-initialises adaLN and the
  output layer, which would give all-zero logits)."""
  g = torch.Generator(device='cuda').manual_seed(int(seed))
  with torch.no_grad():
    for name, p in model.backbone.named_parameters():
      if name.endswith('norm1.weight') or name.endswith('norm2.weight') or name.endswith('norm_final.weight'):
        p.copy_(1 + 0.02 * torch.randn(p.shape, generator=g, device='cuda'))
      else:
        s = 0.15 if 'vocab_embed' in name else 0.08 if 'output_layer.linear.weight' in name else 0.04
        p.copy_(s * torch.randn(p.shape, generator=g, device='cuda'))
  model.backbone.refresh()
  model.ema = None


def _load_from_checkpoint(config, tokenizer):  # main.py:25-33
  from mdlm_amd.diffusion import Diffusion
  model = Diffusion(config, tokenizer=tokenizer).cuda()
  if config.eval.checkpoint_path:
    _load_backbone_weights(model, config.eval.checkpoint_path)
  else:
    _synthetic_weights(model, config.seed)
  if config.eval.disable_ema:
    model.ema = None
  return model


def generate_samples(config, tokenizer):  # main.py:86-117
  model = _load_from_checkpoint(config, tokenizer)
  text_samples = None
  t0 = time.perf_counter()
  n = 0
  for _ in range(config.sampling.num_sample_batches):
    if config.sampling.semi_ar:
      _, intermediate_samples, _ = model.restore_model_and_semi_ar_sample(
        stride_length=config.sampling.stride_length,
        num_strides=config.sampling.num_strides, dt=1 / config.sampling.steps)
      text_samples = intermediate_samples[-1]
    else:
      samples = model.restore_model_and_sample(num_steps=config.sampling.steps)
      text_samples = model.tokenizer.batch_decode(samples.cpu().numpy())
    n += config.loader.eval_batch_size
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  if mdist.rank() == 0:
    print('Text samples:', [s[:120] for s in text_samples[:config.sampling.num_sample_log]])
    print(json.dumps({'mode': 'sample_eval', 'samples_per_rank': n, 'world_size': mdist.world_size(),
                      'samples_per_sec': n * mdist.world_size() / dt}))
  return text_samples


def _ppl_eval(config, tokenizer):  # main.py:119-145 + diffusion.py:399-413
  model = _load_from_checkpoint(config, tokenizer)
  B, L = config.loader.eval_batch_size, config.model.length
  world, rank = mdist.world_size(), mdist.rank()
  n_seq = config.eval.num_batches * B * world
  # synthetic OWT-shaped validation set: ids uniform in [0, 50257) with the
  # BOS/EOS wrap of dataloader.py:286-296, attention_mask of ones
  g = torch.Generator().manual_seed(int(config.seed))
  data = torch.randint(0, tokenizer.vocab_size, (n_seq, L), generator=g)
  data[:, 0] = tokenizer.bos_token_id
  data[:, -1] = tokenizer.eos_token_id
  mine = mdist.shard_indices(n_seq, rank, world)
  model.on_validation_epoch_start()
  torch.manual_seed(int(config.seed) + rank)
  t0 = time.perf_counter()
  for i in range(0, len(mine), B):
    ids = data[mine[i:i + B]].cuda()
    if ids.shape[0] != B:
      break
    batch = {'input_ids': ids, 'attention_mask': torch.ones(B, L, device='cuda')}
    model.validation_step(batch, i // B)
  res = model.on_validation_epoch_end()  # all-reduces the fp64 pair
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  if rank == 0:
    res.update(mode='ppl_eval', tokens_per_sec=n_seq * L / dt, world_size=world)
    print(json.dumps(res))
  return res


def _train(config, tokenizer):  # main.py:148-183 + diffusion.py:390-397, 254-259
  from mdlm_amd.diffusion import Diffusion
  from mdlm_amd.train import Trainer
  world, rank = mdist.world_size(), mdist.rank()
  model = Diffusion(config, tokenizer=tokenizer).cuda()
  if config.eval.checkpoint_path:
    _load_backbone_weights(model, config.eval.checkpoint_path)
  else:
    _synthetic_weights(model, config.seed)   # same seed on every rank (DDP replicas)
  model.seed_rng(mdist.rank_seed(config.seed, rank))
  trainer = Trainer(model, max_norm=config.trainer.gradient_clip_val,
                    num_warmup_steps=config.lr_scheduler.num_warmup_steps)
  # main.py:156-162: resume from checkpointing.resume_ckpt_path when it exists
  ck = config.checkpointing
  last = os.path.join(ck.save_dir, 'checkpoints', 'last.ckpt') if ck.save_dir else None
  resume = ck.resume_ckpt_path or last
  first_step = 0
  if ck.resume_from_ckpt and resume and os.path.exists(resume):
    first_step = int(trainer.load_checkpoint(resume))
    if rank == 0:
      print(json.dumps({'mode': 'train', 'resumed_from': resume, 'step': first_step}), flush=True)
  B, L = config.loader.batch_size, config.model.length
  accum = int(config.trainer.accumulate_grad_batches)
  g = torch.Generator(device='cuda').manual_seed(int(config.seed) + 1000 * rank)
  torch.manual_seed(int(config.seed) + rank)   # _sample_t draws from the default generator
  t0, tok = time.perf_counter(), 0
  for step in range(first_step, int(config.trainer.max_steps)):
    for micro in range(accum):
      ids = torch.randint(0, tokenizer.vocab_size, (B, L), generator=g, device='cuda')
      ids[ids == model.mask_index] = 0
      ids[:, 0] = tokenizer.bos_token_id
      ids[:, -1] = tokenizer.eos_token_id
      batch = {'input_ids': ids, 'attention_mask': torch.ones(B, L, device='cuda')}
      loss = trainer.training_step(batch, accumulate=micro > 0, step=micro == accum - 1)
      tok += B * L
    if rank == 0 and (step + 1) % int(config.trainer.log_every_n_steps) == 0:
      torch.cuda.synchronize()
      dt = time.perf_counter() - t0
      print(json.dumps({'mode': 'train', 'step': step + 1, 'trainer/loss': float(loss),
                        'lr': trainer.last_lr, 'tokens_per_sec': tok * world / dt,
                        'world_size': world}), flush=True)
    if rank == 0 and last and (step + 1) % int(ck.every_n_train_steps) == 0:
      trainer.save_checkpoint(last)
  if rank == 0 and last:
    trainer.save_checkpoint(last)
  return trainer


def main(argv=None):
  argv = sys.argv[1:] if argv is None else argv
  rank, local_rank, world = mdist.init()
  config = load_config(argv, world_size=world)
  if not torch.cuda.is_available():
    raise SystemExit('mdlm_amd runs on an MI355X (ROCm) device; there is no CPU fallback')
  torch.cuda.set_device(mdist.local_device(local_rank))
  torch.manual_seed(config.seed)
  tokenizer = get_tokenizer(config)
  if config.mode == 'sample_eval':
    generate_samples(config, tokenizer)
  elif config.mode == 'ppl_eval':
    _ppl_eval(config, tokenizer)
  elif config.mode == 'train':
    _train(config, tokenizer)
  else:
    raise SystemExit(f'mode={config.mode}: expected sample_eval, ppl_eval or train')
  mdist.barrier()


if __name__ == '__main__':
  main()
