"""Configuration tree with the reference's key names (configs/config.yaml:1-101
and configs/model/small.yaml), restated as Python defaults, plus Hydra-style
`key=value` / `+key=value` overrides.  Hydra/OmegaConf are not required."""
import copy

import yaml


class Cfg(dict):
  """dict with attribute access (OmegaConf-like reads)."""

  def __getattr__(self, k):
    try:
      return self[k]
    except KeyError as e:
      raise AttributeError(k) from e

  def __setattr__(self, k, v):
    self[k] = v

  def get(self, k, default=None):
    return self[k] if k in self else default


def _wrap(x):
  if isinstance(x, dict):
    return Cfg({k: _wrap(v) for k, v in x.items()})
  return x


MODELS = {
  'small': dict(name='small', type='ddit', hidden_size=768, cond_dim=128,
                length=1024, n_blocks=12, n_heads=12, scale_by_sigma=True,
                dropout=0.1, tie_word_embeddings=False),
  'medium': dict(name='medium', type='ddit', hidden_size=1024, cond_dim=128,
                 length=1024, n_blocks=24, n_heads=16, scale_by_sigma=True,
                 dropout=0.1, tie_word_embeddings=False),
  # head_dim is 64 in every kernel-supported model
  'tiny': dict(name='tiny', type='ddit', hidden_size=128, cond_dim=64,
               length=128, n_blocks=2, n_heads=2, scale_by_sigma=True,
               dropout=0.1, tie_word_embeddings=False),
}

# configs/data/{openwebtext-split,lm1b}.yaml: the keys the hot path reads
DATA = {
  'openwebtext': dict(train='openwebtext', valid='openwebtext',
                      tokenizer_name_or_path='gpt2', synthetic=True, vocab_size=50257),
  'openwebtext-split': dict(train='openwebtext-train', valid='openwebtext-valid',
                            tokenizer_name_or_path='gpt2', synthetic=True, vocab_size=50257),
  # bert-base-uncased has its own [MASK] (id 103) -> V stays 30,522 (diffusion.py:85-90)
  'lm1b': dict(train='lm1b', valid='lm1b', tokenizer_name_or_path='bert-base-uncased',
               synthetic=True, vocab_size=30522),
}

DEFAULTS = dict(
  mode='sample_eval', diffusion='absorbing_state', backbone='dit',
  parameterization='subs', time_conditioning=False, T=0, subs_masking=False,
  seed=1,
  loader=dict(global_batch_size=512, eval_global_batch_size=512,
              batch_size=None, eval_batch_size=None, num_workers=0,
              pin_memory=True),
  sampling=dict(predictor='ddpm_cache', steps=128, noise_removal=True,
                num_sample_batches=2, num_sample_log=2, semi_ar=False,
                stride_length=1, num_strides=1,
                # mdlm_amd extensions.  Result-identical knobs (same token ids on
                # the same random stream, checked bit for bit by the GPU tests):
                # fused, cache_policy, masked_rows_only, lse_from_gemm (row lse out of
                # the vocabulary GEMM's epilogue: same value to ~1e-6, same tokens up
                # to exact mask-key ties), and fast_math -- true
                # ranks the race by approximate keys and re-evaluates near-ties
                # in the defining arithmetic, false evaluates every element in it
                # (include/mdlm_hip.h).  lazy_unmask is NOT result-identical
                # (same law, different map from random numbers): opt-in.
                # stream_groups: the batch is cut into that many sample ranges whose
                # backbone forwards run on separate HIP streams (result-identical).
                fused=True, cache_policy='sample', fast_math=True,
                masked_rows_only=True, lse_from_gemm=True, lazy_unmask=False,
                stream_groups=2,
                p_x0_dtype='bf16'),
  training=dict(ema=0.9999, antithetic_sampling=True,
                importance_sampling=False, sampling_eps=1e-3,
                change_of_variables=False),
  eval=dict(checkpoint_path='', disable_ema=False,
            compute_generative_perplexity=False, perplexity_batch_size=8,
            compute_perplexity_on_sanity=False,
            gen_ppl_eval_model_name_or_path='gpt2-large',
            generate_samples=True, num_batches=4),
  optim=dict(weight_decay=0, lr=3e-4, beta1=0.9, beta2=0.999, eps=1e-8),
  trainer=dict(accelerator='cuda', num_nodes=1, devices=1, precision='bf16',
               accumulate_grad_batches=None, gradient_clip_val=1.0, max_steps=1_000_000,
               log_every_n_steps=10),
  lr_scheduler=dict(name='constant_warmup', num_warmup_steps=2500),
  # configs/config.yaml:96-101 + callbacks/checkpoint_every_n_steps.yaml
  # (save_dir '' = no checkpoints; last.ckpt is rewritten every n steps and at the end)
  checkpointing=dict(save_dir='', resume_from_ckpt=True, resume_ckpt_path=None,
                     every_n_train_steps=500),
  noise=dict(type='loglinear', sigma_min=1e-4, sigma_max=20),
  data=dict(train='openwebtext', valid='openwebtext',
            tokenizer_name_or_path='gpt2', synthetic=True, vocab_size=50257),
  model=MODELS['small'],
)


def _div_up(x, y):
  return (x + y - 1) // y


def _set(cfg, dotted, value, create):
  keys = dotted.split('.')
  d = cfg
  for k in keys[:-1]:
    if k not in d:
      if not create:
        raise KeyError(f'unknown config key {dotted!r} (use +{dotted}=...)')
      d[k] = Cfg()
    d = d[k]
  if keys[-1] not in d and not create:
    raise KeyError(f'unknown config key {dotted!r} (use +{dotted}=...)')
  d[keys[-1]] = value


def load_config(overrides=(), world_size=1):
  """Compose defaults + overrides (`a.b=c`, `+a.b=c`, `model=small`)."""
  cfg = _wrap(copy.deepcopy(DEFAULTS))
  for ov in overrides:
    if '=' not in ov:
      raise ValueError(f'override {ov!r} is not key=value')
    key, val = ov.split('=', 1)
    create = key.startswith('+')
    key = key.lstrip('+')
    if key == 'model' and val in MODELS:
      cfg['model'] = _wrap(copy.deepcopy(MODELS[val]))
      continue
    if key == 'data' and val in DATA:
      cfg['data'] = _wrap(copy.deepcopy(DATA[val]))
      continue
    _set(cfg, key, yaml.safe_load(val), create)
  cfg.trainer.devices = cfg.trainer.devices or world_size
  n = max(1, int(cfg.trainer.devices) * int(cfg.trainer.num_nodes))
  # configs/config.yaml:24-25: per-device batch = div_up(global, devices*nodes)
  if cfg.loader.batch_size is None:
    cfg.loader.batch_size = _div_up(cfg.loader.global_batch_size, n)
  if cfg.loader.eval_batch_size is None:
    cfg.loader.eval_batch_size = _div_up(cfg.loader.eval_global_batch_size, n)
  # configs/config.yaml:68: accumulate = div_up(global, devices * batch * nodes)
  if cfg.trainer.accumulate_grad_batches is None:
    cfg.trainer.accumulate_grad_batches = _div_up(cfg.loader.global_batch_size,
                                                  n * cfg.loader.batch_size)
  return cfg
