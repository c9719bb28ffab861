"""CPU oracle for the MDLM denoise-and-sample hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-torch fp32 CPU restatement of
the reference algorithm (kuleshov-group/mdlm).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it; the product package `mdlm_amd` never does (it fails loudly when the HIP
library is missing).

Parity pin: the reference ships no tests and no golden vectors,
so this restatement is pinned against outputs of the reference itself, produced
in the build container by `oracle/gen_golden.py` (imports /root/reference with
stub modules for the absent third-party wheels) and committed under
`tests/golden/`.  `tests/test_oracle_golden.py` checks every function here
against those vectors.  Third-party arithmetic that is absent from the
reference tree (flash-attn 2.5.6 attention + rotary, requirements.yaml:36) is
restated from its published contract (SDPA softmax(QK^T/sqrt(d))V, NeoX-style
non-interleaved rotary) -- that boundary is "parity unpinned" by any reference
test.

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

NEG_INFINITY = -1000000.0  # diffusion.py:150


# --------------------------------------------------------------------------
# noise schedule
# --------------------------------------------------------------------------
def loglinear_total_noise(t, eps=1e-3):
  """noise_schedule.py:143-144  sigma(t) = -log1p(-(1-eps) t)."""
  return -torch.log1p(-(1 - eps) * t)


def loglinear_rate_noise(t, eps=1e-3):
  """noise_schedule.py:140-141  sigma'(t) = (1-eps)/(1-(1-eps) t)."""
  return (1 - eps) / (1 - (1 - eps) * t)


def loglinear_noise(t, eps=1e-3):
  """noise_schedule.py:41-43 (Noise.forward) -> (total, rate)."""
  return loglinear_total_noise(t, eps), loglinear_rate_noise(t, eps)


# --------------------------------------------------------------------------
# sampler pieces
# --------------------------------------------------------------------------
def sample_categorical(categorical_probs, u):
  """diffusion.py:24-28 with the uniform draw `u` supplied by the caller.

  gumbel_norm = 1e-10 - log(u + 1e-10); argmax(p / gumbel_norm).
  """
  gumbel_norm = 1e-10 - (u + 1e-10).log()
  return (categorical_probs / gumbel_norm).argmax(dim=-1)


def subs_parameterization(logits, xt, mask_index, neg_infinity=NEG_INFINITY):
  """diffusion.py:261-277.  Mutates `logits[:, :, mask]` like the reference."""
  logits[:, :, mask_index] += neg_infinity
  logits = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
  unmasked_indices = (xt != mask_index)
  logits[unmasked_indices] = neg_infinity
  logits[unmasked_indices, xt[unmasked_indices]] = 0
  return logits


def process_sigma(sigma, time_conditioning=False):
  """diffusion.py:301-310."""
  if sigma.ndim > 1:
    sigma = sigma.squeeze(-1)
  if not time_conditioning:
    sigma = torch.zeros_like(sigma)
  assert sigma.ndim == 1, sigma.shape
  return sigma


def q_xt(x, move_chance, u, mask_index):
  """diffusion.py:575-586 with the uniform draw supplied."""
  move_indices = u < move_chance
  return torch.where(move_indices, mask_index, x)


def sample_t(n, u, sampling_eps=1e-3, antithetic=True):
  """diffusion.py:800-808 (importance_sampling=False) with `u = rand(n)`."""
  _eps_t = u
  if antithetic:
    offset = torch.arange(n) / n
    _eps_t = (_eps_t / n + offset) % 1
  return (1 - sampling_eps) * _eps_t + sampling_eps


def ddpm_caching_update(forward_fn, x, t, dt, u, mask_index, p_x0=None):
  """diffusion.py:592-610.  `forward_fn(x, sigma)` returns SUBS log-probs."""
  sigma_t, _ = loglinear_noise(t)
  if t.ndim > 1:
    t = t.squeeze(-1)
  assert t.ndim == 1
  move_chance_t = t[:, None, None]
  move_chance_s = (t - dt)[:, None, None]
  if p_x0 is None:
    p_x0 = forward_fn(x, sigma_t).exp()
  q_xs = p_x0 * (move_chance_t - move_chance_s)
  q_xs[:, :, mask_index] = move_chance_s[:, :, 0]
  _x = sample_categorical(q_xs, u)
  copy_flag = (x != mask_index).to(x.dtype)
  return p_x0, copy_flag * x + (1 - copy_flag) * _x


# --------------------------------------------------------------------------
# DiT backbone (models/dit.py), fp32, SDPA attention
# --------------------------------------------------------------------------
def timestep_embedding(t, dim, max_period=10000):
  """dit.py:162-184."""
  half = dim // 2
  freqs = torch.exp(
    -math.log(max_period)
    * torch.arange(start=0, end=half, dtype=torch.float32) / half)
  args = t[:, None].float() * freqs[None]
  return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def rotary_cos_sin(seq_len, head_dim, base=10000):
  """dit.py:81-104 -> cos, sin of shape (L, head_dim/2) (dit.py:112-114)."""
  inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
  t = torch.arange(seq_len).float()
  freqs = torch.einsum('i,j->ij', t, inv_freq)
  return freqs.cos(), freqs.sin()


def apply_rotary_qk(qkv, cos, sin):
  """flash_attn.layers.rotary.apply_rotary_emb_qkv_ contract (dit.py:112-115):
  non-interleaved (NeoX) rotation of q and k over the full head dim; v kept.
  qkv (B, L, 3, H, d); cos/sin (L, d/2)."""
  d2 = cos.shape[-1]
  out = qkv.clone()
  c = cos[None, :, None, :]
  s = sin[None, :, None, :]
  for i in (0, 1):
    x1 = qkv[:, :, i, :, :d2]
    x2 = qkv[:, :, i, :, d2:2 * d2]
    out[:, :, i, :, :d2] = x1 * c - x2 * s
    out[:, :, i, :, d2:2 * d2] = x1 * s + x2 * c
  return out


def layer_norm_w(x, weight):
  """dit.py:126-134: F.layer_norm (no affine, eps 1e-5) in fp32 times weight."""
  return F.layer_norm(x.float(), [x.shape[-1]]) * weight[None, None, :]


def modulate(x, shift, scale):
  """dit.py:46-49 / 74-78 (shift, scale already (B,1,D))."""
  return x * (1 + scale) + shift


def dit_block(p, pre, x, cos, sin, c, n_heads):
  """dit.py:244-288 (eval mode: dropout is the identity)."""
  B, L, D = x.shape
  mod = F.linear(c, p[pre + 'adaLN_modulation.weight'],
                 p[pre + 'adaLN_modulation.bias'])[:, None]
  (shift_msa, scale_msa, gate_msa,
   shift_mlp, scale_mlp, gate_mlp) = mod.chunk(6, dim=2)
  x_skip = x
  h = modulate(layer_norm_w(x, p[pre + 'norm1.weight']), shift_msa, scale_msa)
  qkv = F.linear(h, p[pre + 'attn_qkv.weight'])
  qkv = qkv.view(B, L, 3, n_heads, D // n_heads)
  qkv = apply_rotary_qk(qkv, cos, sin)
  q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))  # (B,H,L,d)
  a = F.scaled_dot_product_attention(q, k, v)  # dit.py:272-273 (non-causal)
  a = a.transpose(1, 2).reshape(B, L, D)
  x = x_skip + gate_msa * F.linear(a, p[pre + 'attn_out.weight'])
  h = modulate(layer_norm_w(x, p[pre + 'norm2.weight']), shift_mlp, scale_mlp)
  h = F.linear(h, p[pre + 'mlp.0.weight'], p[pre + 'mlp.0.bias'])
  h = F.gelu(h, approximate='tanh')
  h = F.linear(h, p[pre + 'mlp.2.weight'], p[pre + 'mlp.2.bias'])
  return x + gate_mlp * h


def dit_forward(p, indices, sigma, n_heads, n_blocks):
  """dit.py:359-370 with a flat state-dict `p` (reference key names)."""
  x = p['vocab_embed.embedding'][indices]
  cond_dim = p['sigma_map.mlp.0.weight'].shape[0]
  freq = p['sigma_map.mlp.0.weight'].shape[1]
  t_emb = timestep_embedding(sigma, freq)
  t_emb = F.linear(t_emb, p['sigma_map.mlp.0.weight'],
                   p['sigma_map.mlp.0.bias'])
  t_emb = F.linear(F.silu(t_emb), p['sigma_map.mlp.2.weight'],
                   p['sigma_map.mlp.2.bias'])
  c = F.silu(t_emb)
  assert c.shape[-1] == cond_dim
  D = x.shape[-1]
  cos, sin = rotary_cos_sin(x.shape[1], D // n_heads)
  for i in range(n_blocks):
    x = dit_block(p, f'blocks.{i}.', x, cos, sin, c, n_heads)
  mod = F.linear(c, p['output_layer.adaLN_modulation.weight'],
                 p['output_layer.adaLN_modulation.bias'])[:, None]
  shift, scale = mod.chunk(2, dim=2)
  x = modulate(layer_norm_w(x, p['output_layer.norm_final.weight']),
               shift, scale)
  return F.linear(x, p['output_layer.linear.weight'],
                  p['output_layer.linear.bias'])


def synthetic_dit_params(hidden, n_heads, n_blocks, cond_dim, vocab, seed=1,
                         std=0.02, freq=256):
  """Seeded synthetic weights with the reference state-dict layout (SURVEY
  §8(a)): N(0,std) for every matrix incl. the reference's zero-initialised
  adaLN / output layers, LN weights 1 + N(0,std), biases N(0,std)."""
  g = torch.Generator().manual_seed(seed)

  def rn(*shape, s=std):
    return torch.randn(*shape, generator=g) * s
  p = {}
  p['vocab_embed.embedding'] = rn(vocab, hidden, s=1.0 / math.sqrt(hidden) * 4)
  p['sigma_map.mlp.0.weight'] = rn(cond_dim, freq, s=0.05)
  p['sigma_map.mlp.0.bias'] = rn(cond_dim)
  p['sigma_map.mlp.2.weight'] = rn(cond_dim, cond_dim, s=0.05)
  p['sigma_map.mlp.2.bias'] = rn(cond_dim)
  for i in range(n_blocks):
    pre = f'blocks.{i}.'
    p[pre + 'norm1.weight'] = 1 + rn(hidden)
    p[pre + 'attn_qkv.weight'] = rn(3 * hidden, hidden, s=0.04)
    p[pre + 'attn_out.weight'] = rn(hidden, hidden)
    p[pre + 'norm2.weight'] = 1 + rn(hidden)
    p[pre + 'mlp.0.weight'] = rn(4 * hidden, hidden)
    p[pre + 'mlp.0.bias'] = rn(4 * hidden)
    p[pre + 'mlp.2.weight'] = rn(hidden, 4 * hidden)
    p[pre + 'mlp.2.bias'] = rn(hidden)
    p[pre + 'adaLN_modulation.weight'] = rn(6 * hidden, cond_dim, s=0.05)
    p[pre + 'adaLN_modulation.bias'] = rn(6 * hidden, s=0.1)
  p['output_layer.norm_final.weight'] = 1 + rn(hidden)
  p['output_layer.linear.weight'] = rn(vocab, hidden, s=0.08)
  p['output_layer.linear.bias'] = rn(vocab)
  p['output_layer.adaLN_modulation.weight'] = rn(2 * hidden, cond_dim, s=0.05)
  p['output_layer.adaLN_modulation.bias'] = rn(2 * hidden, s=0.1)
  return p


# --------------------------------------------------------------------------
# Diffusion-level restatement
# --------------------------------------------------------------------------
class OracleDiffusion:
  """CPU restatement of the hot-path methods of `diffusion.Diffusion`
  (diffusion.py:68-1024) over a flat DiT state dict.  RNG draws come from
  `rand_fn(shape)`, which defaults to `torch.rand` on the global CPU generator
  (what `torch.rand_like` / `torch.rand` consume in the reference)."""

  def __init__(self, params, n_heads, n_blocks, vocab_size, mask_index,
               length, time_conditioning=False, sampling_eps=1e-3,
               antithetic_sampling=True, rand_fn=None):
    self.p = params
    self.n_heads, self.n_blocks = n_heads, n_blocks
    self.vocab_size, self.mask_index = vocab_size, mask_index
    self.length = length
    self.time_conditioning = time_conditioning
    self.sampling_eps = sampling_eps
    self.antithetic_sampling = antithetic_sampling
    self.neg_infinity = NEG_INFINITY
    self.rand = rand_fn or (lambda *shape: torch.rand(*shape))
    self.nfe = 0

  def backbone(self, x, sigma):
    self.nfe += 1
    return dit_forward(self.p, x, sigma, self.n_heads, self.n_blocks)

  def forward(self, x, sigma):
    """diffusion.py:312-327 (subs)."""
    sigma = process_sigma(sigma, self.time_conditioning)
    logits = self.backbone(x, sigma)
    return subs_parameterization(logits, x, self.mask_index)

  def _ddpm_caching_update(self, x, t, dt, p_x0=None):
    """diffusion.py:592-610; the draw has the shape of q_xs (B,L,V)."""
    sigma_t, _ = loglinear_noise(t)
    tt = t.squeeze(-1) if t.ndim > 1 else t
    mt = tt[:, None, None]
    ms = (tt - dt)[:, None, None]
    if p_x0 is None:
      p_x0 = self.forward(x, sigma_t).exp()
    q_xs = p_x0 * (mt - ms)
    q_xs[:, :, self.mask_index] = ms[:, :, 0]
    u = self.rand(*q_xs.shape)
    _x = sample_categorical(q_xs, u)
    copy_flag = (x != self.mask_index).to(x.dtype)
    return p_x0, copy_flag * x + (1 - copy_flag) * _x

  def _sample(self, batch_size, num_steps, eps=1e-5, noise_removal=True,
              trace=None):
    """diffusion.py:657-698 (sampler == 'ddpm_cache')."""
    x = self.mask_index * torch.ones(batch_size, self.length,
                                     dtype=torch.int64)
    timesteps = torch.linspace(1, eps, num_steps + 1)
    dt = (1 - eps) / num_steps
    p_x0_cache = None
    for i in range(num_steps):
      t = timesteps[i] * torch.ones(x.shape[0], 1)
      p_x0_cache, x_next = self._ddpm_caching_update(x, t, dt, p_x0=p_x0_cache)
      if (not torch.allclose(x_next, x)) or self.time_conditioning:
        p_x0_cache = None
      x = x_next
      if trace is not None:
        trace.append(x.clone())
    if noise_removal:
      t = timesteps[-1] * torch.ones(x.shape[0], 1)
      x = self.forward(x, loglinear_total_noise(t)).argmax(dim=-1)
    return x

  def sample_subs_guidance(self, n_samples, stride_length, num_strides,
                           dt=0.001):
    """diffusion.py:957-987 (token part; decode/EOS bookkeeping is host
    string work outside the numeric path)."""
    ones = torch.ones(n_samples, dtype=torch.float32)
    num_steps = int(1 / dt)
    sampling_steps = 0
    intermediate_tokens = []
    target = None
    for _ in range(num_strides + 1):
      p_x0_cache = None
      x = self.mask_index * torch.ones(n_samples, self.length,
                                       dtype=torch.int64)
      if target is not None:
        x[:, :-stride_length] = target
      for i in range(num_steps + 1):
        p_x0_cache, x_next = self._ddpm_caching_update(
          x=x, t=(1 - i * dt) * ones, dt=dt, p_x0=p_x0_cache)
        if (not torch.allclose(x_next, x)) or self.time_conditioning:
          p_x0_cache = None
          sampling_steps += 1
        x = x_next
      x = self.forward(x, 0 * ones).argmax(dim=-1)
      intermediate_tokens.append(x[:, :stride_length].clone())
      target = x[:, stride_length:]
    intermediate_tokens.append(target.clone())
    return sampling_steps, torch.cat(intermediate_tokens, dim=1)

  def _forward_pass_diffusion(self, x0, t=None, u_xt=None):
    """diffusion.py:847-894, subs / T=0 / no change_of_variables branch."""
    if t is None:
      t = sample_t(x0.shape[0], self.rand(x0.shape[0]), self.sampling_eps,
                   self.antithetic_sampling)
    sigma, dsigma = loglinear_noise(t)
    move_chance = 1 - torch.exp(-sigma[:, None])
    if u_xt is None:
      u_xt = self.rand(*x0.shape)
    xt = q_xt(x0, move_chance, u_xt, self.mask_index)
    model_output = self.forward(xt, sigma[:, None])
    log_p_theta = torch.gather(model_output, -1, x0[:, :, None]).squeeze(-1)
    return -log_p_theta * (dsigma / torch.expm1(sigma))[:, None], xt

  def _loss(self, x0, attention_mask, t=None, u_xt=None):
    """diffusion.py:896-916 -> (token_nll, nlls, token_mask)."""
    loss, _ = self._forward_pass_diffusion(x0, t, u_xt)
    nlls = loss * attention_mask
    count = attention_mask.sum()
    return nlls.sum() / count, nlls, attention_mask


def nelbo_from_logits(logits, xt, x0, t, attention_mask, mask_index):
  """diffusion.py:884-894 + 908-912 given backbone logits (fp32)."""
  sigma, dsigma = loglinear_noise(t)
  lp = subs_parameterization(logits.clone(), xt, mask_index)
  log_p_theta = torch.gather(lp, -1, x0[:, :, None]).squeeze(-1)
  loss = -log_p_theta * (dsigma / torch.expm1(sigma))[:, None]
  nlls = loss * attention_mask
  return nlls, nlls.sum() / attention_mask.sum()


def perplexity(nll_sum, weight_sum):
  """diffusion.py:58-65 (fp64 MeanMetric): exp(sum(nll*w)/sum(w))."""
  return math.exp(float(nll_sum) / float(weight_sum))


def tiny_config():
  return SimpleNamespace(hidden=64, n_heads=4, n_blocks=2, cond_dim=16,
                         vocab=64, length=16)
