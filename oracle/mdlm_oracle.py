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


def ddpm_update(forward_fn, x, t, dt, u, mask_index):
  """diffusion.py:612-637 (sampler == 'ddpm': no cache, sigma-conditioned)."""
  sigma_t = loglinear_total_noise(t)
  sigma_s = loglinear_total_noise(t - dt)
  if sigma_t.ndim > 1:
    sigma_t = sigma_t.squeeze(-1)
  if sigma_s.ndim > 1:
    sigma_s = sigma_s.squeeze(-1)
  move_chance_t = (1 - torch.exp(-sigma_t))[:, None, None]
  move_chance_s = (1 - torch.exp(-sigma_s))[:, None, None]
  q_xs = forward_fn(x, sigma_t).exp() * (move_chance_t - move_chance_s)
  q_xs[:, :, mask_index] = move_chance_s[:, :, 0]
  _x = sample_categorical(q_xs, u)
  copy_flag = (x != mask_index).to(x.dtype)
  return copy_flag * x + (1 - copy_flag) * _x


def get_score(forward_fn, x, sigma, mask_index, neg_infinity=NEG_INFINITY):
  """diffusion.py:721-764 (subs branch): ratio p_t(y)/p_t(x) for every y."""
  model_output = forward_fn(x, sigma)
  log_k = -torch.log(torch.expm1(sigma)).squeeze(-1)
  assert log_k.ndim == 1
  masked_score = model_output + log_k[:, None, None]
  masked_score[:, :, mask_index] = 0
  unmasked_score = neg_infinity * torch.ones_like(model_output)
  unmasked_score = torch.scatter(unmasked_score, -1, x[..., None],
                                 torch.zeros_like(unmasked_score[..., :1]))
  unmasked_score[:, :, mask_index] = -(log_k[:, None] * torch.ones_like(x))
  masked = (x == mask_index).to(model_output.dtype)[:, :, None]
  return (masked_score * masked + unmasked_score * (1 - masked)).exp()


def staggered_score(score, dsigma, mask_index):
  """diffusion.py:766-771."""
  score = score.clone()
  extra_const = (1 - dsigma.exp()) * score.sum(dim=-1)
  score *= dsigma.exp()[:, None]
  score[..., mask_index] += extra_const
  return score


def transp_transition(i, sigma, mask_index, vocab_size):
  """diffusion.py:791-798 (sigma broadcast against i[..., None], utils 18-21)."""
  sigma = sigma.view(*sigma.shape, *((1,) * (i[..., None].dim() - sigma.dim())))
  edge = torch.exp(-sigma) * F.one_hot(i, num_classes=vocab_size)
  edge += torch.where(i == mask_index, 1 - torch.exp(-sigma).squeeze(-1),
                      0)[..., None]
  return edge


def analytic_update(forward_fn, x, t, step_size, u, mask_index, vocab_size):
  """diffusion.py:773-780 (sampler == 'analytic')."""
  curr_sigma = loglinear_total_noise(t)
  next_sigma = loglinear_total_noise(t - step_size)
  dsigma = curr_sigma - next_sigma
  score = get_score(forward_fn, x, curr_sigma, mask_index)
  stag_score = staggered_score(score, dsigma, mask_index)
  probs = stag_score * transp_transition(x, dsigma, mask_index, vocab_size)
  return sample_categorical(probs, u)


def denoiser_update(forward_fn, x, t, u, mask_index, vocab_size):
  """diffusion.py:782-789 (noise removal of the analytic sampler)."""
  sigma = loglinear_total_noise(t)
  score = get_score(forward_fn, x, sigma, mask_index)
  stag_score = staggered_score(score, sigma, mask_index)
  probs = stag_score * transp_transition(x, sigma, mask_index, vocab_size)
  probs[..., mask_index] = 0
  return sample_categorical(probs, u)


def d3pm_loss(model_output, xt, x0, t, T, mask_index):
  """diffusion.py:329-358 (discrete-time NELBO term, T > 0)."""
  dt = 1 / T
  t = t[:, None].clamp(0., 1. - 1e-4)
  alpha_t = 1 - t + torch.zeros_like(xt)
  alpha_s = 1 - (t - dt) + torch.zeros_like(xt)
  log_x_theta_at_x0 = torch.gather(model_output, -1, x0[:, :, None]).squeeze(-1)
  x_theta_at_m = model_output[:, :, mask_index].exp()
  term_1_coef = dt / t
  term_1_log_nr = torch.log(alpha_t * x_theta_at_m / t + 1)
  term_1_log_dr = log_x_theta_at_x0
  term_2_coef = 1 - dt / t
  term_2_log_nr = term_1_log_nr
  term_2_log_dr = torch.log(alpha_s * x_theta_at_m / (t - dt) + 1)
  L_vb_masked = (term_1_coef * (term_1_log_nr - term_1_log_dr)
                 + term_2_coef * (term_2_log_nr - term_2_log_dr))
  return T * (L_vb_masked * (xt == mask_index))


def discretize_t(t, T):
  """diffusion.py:849-853: t -> {1/T, ..., 1} grid."""
  t = (t * T).to(torch.int)
  t = t / T
  t += (1 / T)
  return t


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


def _dropout(y, keep, prob):
  """F.dropout(y, p=prob, training=True) (dit.py:27-29) with the keep mask
  supplied by the caller; keep=None is eval mode (identity)."""
  if keep is None:
    return y
  return y * keep.to(y.dtype) / (1.0 - prob)


def dit_block(p, pre, x, cos, sin, c, n_heads, keep=None, prob=0.1):
  """dit.py:244-288.  `keep` = None (eval: dropout is the identity) or a pair
  of (B,L,D) keep masks for the two bias_dropout_add_scale calls (training,
  dit.py:20-34, 277-287)."""
  k1, k2 = keep if keep is not None else (None, None)
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
  x = x_skip + gate_msa * _dropout(F.linear(a, p[pre + 'attn_out.weight']), k1, prob)
  h = modulate(layer_norm_w(x, p[pre + 'norm2.weight']), shift_mlp, scale_mlp)
  h = F.linear(h, p[pre + 'mlp.0.weight'], p[pre + 'mlp.0.bias'])
  h = F.gelu(h, approximate='tanh')
  h = F.linear(h, p[pre + 'mlp.2.weight'], p[pre + 'mlp.2.bias'])
  return x + gate_mlp * _dropout(h, k2, prob)


def dit_forward(p, indices, sigma, n_heads, n_blocks, keep=None, prob=0.1):
  """dit.py:359-370 with a flat state-dict `p` (reference key names).  `keep`:
  None (eval) or a list of per-block keep-mask pairs (training)."""
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
    x = dit_block(p, f'blocks.{i}.', x, cos, sin, c, n_heads,
                  None if keep is None else keep[i], prob)
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
               antithetic_sampling=True, rand_fn=None, sampler='ddpm_cache',
               T=0):
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
    self.sampler = sampler
    self.T = T

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
    """diffusion.py:657-698 (samplers 'ddpm_cache', 'ddpm', 'analytic')."""
    x = self.mask_index * torch.ones(batch_size, self.length,
                                     dtype=torch.int64)
    timesteps = torch.linspace(1, eps, num_steps + 1)
    dt = (1 - eps) / num_steps
    p_x0_cache = None
    shape_v = (batch_size, self.length, self.vocab_size)
    for i in range(num_steps):
      t = timesteps[i] * torch.ones(x.shape[0], 1)
      if self.sampler == 'ddpm':
        x = ddpm_update(self.forward, x, t, dt, self.rand(*shape_v),
                        self.mask_index)
      elif self.sampler == 'ddpm_cache':
        p_x0_cache, x_next = self._ddpm_caching_update(x, t, dt, p_x0=p_x0_cache)
        if (not torch.allclose(x_next, x)) or self.time_conditioning:
          p_x0_cache = None
        x = x_next
      else:
        x = analytic_update(self.forward, x, t, dt, self.rand(*shape_v),
                            self.mask_index, self.vocab_size)
      if trace is not None:
        trace.append(x.clone())
    if noise_removal:
      t = timesteps[-1] * torch.ones(x.shape[0], 1)
      if self.sampler == 'analytic':
        x = denoiser_update(self.forward, x, t, self.rand(*shape_v),
                            self.mask_index, self.vocab_size)
      else:
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
    """diffusion.py:847-894, subs / no change_of_variables; T = 0 (continuous
    time) or T > 0 (d3pm-style discrete-time term, reconstruction loss 0)."""
    if t is None:
      t = sample_t(x0.shape[0], self.rand(x0.shape[0]), self.sampling_eps,
                   self.antithetic_sampling)
    if self.T > 0:
      t = discretize_t(t, self.T)
    sigma, dsigma = loglinear_noise(t)
    move_chance = 1 - torch.exp(-sigma[:, None])
    if u_xt is None:
      u_xt = self.rand(*x0.shape)
    xt = q_xt(x0, move_chance, u_xt, self.mask_index)
    model_output = self.forward(xt, sigma[:, None])
    if self.T > 0:
      return d3pm_loss(model_output, xt, x0, t, self.T, self.mask_index), xt
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


# --------------------------------------------------------------------------
# Training step: loss + autograd gradients, global-norm clip,
# AdamW, EMA, constant-with-warmup learning rate
# --------------------------------------------------------------------------
def training_loss(params, x0, attention_mask, t, u_xt, n_heads, n_blocks,
                  mask_index, keep=None, prob=0.1):
  """diffusion.py:390-397 -> _compute_loss (360-384) -> _loss (896-916) with
  the backbone in train mode.  Returns (token_nll, nlls)."""
  sigma, dsigma = loglinear_noise(t)
  move_chance = 1 - torch.exp(-sigma[:, None])
  xt = q_xt(x0, move_chance, u_xt, mask_index)
  logits = dit_forward(params, xt, torch.zeros_like(sigma), n_heads, n_blocks,
                       keep, prob)
  lp = subs_parameterization(logits, xt, mask_index)
  log_p_theta = torch.gather(lp, -1, x0[:, :, None]).squeeze(-1)
  loss = -log_p_theta * (dsigma / torch.expm1(sigma))[:, None]
  nlls = loss * attention_mask
  return nlls.sum() / attention_mask.sum(), nlls


def loss_and_grads(params, *args, **kwargs):
  """Gradients of `training_loss` w.r.t. every entry of the flat state dict
  (torch autograd on the fp32 CPU restatement)."""
  leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
  with torch.enable_grad():
    loss, nlls = training_loss(leaves, *args, **kwargs)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
  return loss.detach(), nlls.detach(), {
    k: (g if g is not None else torch.zeros_like(params[k]))
    for k, g in zip(leaves, grads)}


def clip_coef(grads, max_norm=1.0):
  """Lightning `gradient_clip_val: 1.0` (configs/config.yaml:69) ->
  torch.nn.utils.clip_grad_norm_: total 2-norm over all gradients,
  coef = min(1, max_norm / (total + 1e-6)).  Returns (total_norm, coef)."""
  total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
  return total, torch.clamp(max_norm / (total + 1e-6), max=1.0)


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8,
               weight_decay=0.0):
  """torch.optim.AdamW as configured at diffusion.py:449-459 (single-tensor
  path of torch/optim/adamw.py): decoupled decay, bias-corrected moments.
  `step` is 1-based.  Updates p, m, v in place."""
  p.mul_(1 - lr * weight_decay)
  m.lerp_(g, 1 - beta1)
  v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
  bc1 = 1 - beta1 ** step
  bc2 = 1 - beta2 ** step
  denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
  p.addcdiv_(m, denom, value=-(lr / bc1))


def ema_decay(decay, num_updates):
  """models/ema.py:40-43 (`num_updates` already incremented)."""
  return min(decay, (1 + num_updates) / (10 + num_updates))


def ema_update(shadow, p, decay, num_updates):
  """models/ema.py:29-49: s -= (1 - decay_t) * (s - p)."""
  shadow.sub_((1.0 - ema_decay(decay, num_updates)) * (shadow - p))


def constant_warmup_lr(base_lr, step, num_warmup_steps=2500):
  """configs/lr_scheduler/constant_warmup.yaml ->
  transformers.get_constant_schedule_with_warmup: the lr used by optimizer step
  number `step` (0-based) is base_lr * min(1, step / max(1, warmup))."""
  if step < num_warmup_steps:
    return base_lr * float(step) / float(max(1.0, num_warmup_steps))
  return base_lr


def training_step(state, x0, attention_mask, t, u_xt, n_heads, n_blocks,
                  mask_index, lr, keep=None, prob=0.1, max_norm=1.0,
                  ema=0.9999, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
  """One optimizer step on `state` = dict(params, m, v, shadow, step):
  loss.backward -> clip -> AdamW (diffusion.py:449-459) -> EMA update
  (diffusion.py:254-259).  Returns (loss, total_norm, grads)."""
  loss, _, grads = loss_and_grads(state['params'], x0, attention_mask, t, u_xt,
                                  n_heads, n_blocks, mask_index, keep, prob)
  total, coef = clip_coef(grads, max_norm)
  state['step'] += 1
  for k, p in state['params'].items():
    adamw_step(p, grads[k] * coef, state['m'][k], state['v'][k], state['step'],
               lr, betas[0], betas[1], eps, weight_decay)
    ema_update(state['shadow'][k], p, ema, state['step'])
  return loss, total, grads


def new_train_state(params):
  return dict(params={k: v.clone() for k, v in params.items()},
              m={k: torch.zeros_like(v) for k, v in params.items()},
              v={k: torch.zeros_like(v) for k, v in params.items()},
              shadow={k: v.clone() for k, v in params.items()}, step=0)


def tiny_config():
  return SimpleNamespace(hidden=64, n_heads=4, n_blocks=2, cond_dim=16,
                         vocab=64, length=16)
