"""Generate golden vectors by running the REFERENCE itself (container only).

TEST INFRASTRUCTURE.  Imports /root/reference/{diffusion,models/dit,
noise_schedule}.py with in-memory stub modules for the third-party wheels that
are absent here (hydra, lightning, torchmetrics, omegaconf, timm, flash_attn,
mamba_ssm, causal_conv1d) and calls the hot-path methods unbound on a
SimpleNamespace `self`.  flash-attn's two entry points
are replaced by stubs written here from their API contract: SDPA for
`flash_attn_varlen_qkvpacked_func`, NeoX (non-interleaved) rotary for
`apply_rotary_emb_qkv_`.

Outputs only *data* (inputs + expected outputs) to tests/golden/*.npz; no
reference source text is copied.  /root/reference does not exist on the GPU
box, so nothing at test time imports this file's products other than the npz.

Run:  python oracle/gen_golden.py
"""
import hashlib
import importlib.machinery
import os
import sys
import types

sys.dont_write_bytecode = True
REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import transformers  # noqa: F401  (real; must precede the stubs)
import datasets  # noqa: F401


def _stub(name, **attrs):
  m = types.ModuleType(name)
  m.__spec__ = importlib.machinery.ModuleSpec(name, None)
  m.__path__ = []
  for k, v in attrs.items():
    setattr(m, k, v)
  sys.modules[name] = m
  return m


def install_stubs():
  class LightningModule(nn.Module):
    def save_hyperparameters(self, *a, **k):
      pass

  class MeanMetric:
    pass

  class _OmegaConf:
    @staticmethod
    def create(x):
      return x

    @staticmethod
    def register_new_resolver(*a, **k):
      pass

  def apply_rotary_emb_qkv_(qkv, cos, sin):
    # contract: in-place NeoX rotary on q,k of (B,L,3,H,d); cos/sin (L,d/2)
    d2 = cos.shape[-1]
    c = cos[None, :, None, :]
    s = sin[None, :, None, :]
    for i in (0, 1):
      x1 = qkv[:, :, i, :, :d2].clone()
      x2 = qkv[:, :, i, :, d2:2 * d2].clone()
      qkv[:, :, i, :, :d2] = x1 * c - x2 * s
      qkv[:, :, i, :, d2:2 * d2] = x1 * s + x2 * c
    return qkv

  def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, p,
                                       causal=False):
    # contract: qkv (B*L,3,H,d), equal lengths -> (B*L,H,d); softmax(QK^T/sqrt d)V
    T, _, H, d = qkv.shape
    B = T // max_seqlen
    q, k, v = (qkv[:, i].view(B, max_seqlen, H, d).transpose(1, 2)
               for i in range(3))
    o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
    return o.transpose(1, 2).reshape(T, H, d)

  _stub('hydra'); _stub('hydra.utils', instantiate=lambda *a, **k: None)
  sys.modules['hydra'].utils = sys.modules['hydra.utils']
  L = _stub('lightning', LightningModule=LightningModule,
            seed_everything=lambda s: torch.manual_seed(s))
  _stub('lightning.pytorch')
  _stub('lightning.pytorch.utilities', rank_zero_only=lambda f: f)
  L.pytorch = sys.modules['lightning.pytorch']
  L.pytorch.utilities = sys.modules['lightning.pytorch.utilities']
  tm = _stub('torchmetrics', MetricCollection=object)
  _stub('torchmetrics.aggregation', MeanMetric=MeanMetric)
  tm.aggregation = sys.modules['torchmetrics.aggregation']
  _stub('omegaconf', OmegaConf=_OmegaConf, DictConfig=dict)
  _stub('timm'); _stub('timm.scheduler', CosineLRScheduler=type('CosineLRScheduler', (), {}))
  fa = _stub('flash_attn')
  _stub('flash_attn.layers')
  _stub('flash_attn.layers.rotary', apply_rotary_emb_qkv_=apply_rotary_emb_qkv_)
  _stub('flash_attn.flash_attn_interface',
        flash_attn_varlen_qkvpacked_func=flash_attn_varlen_qkvpacked_func)
  fa.layers = sys.modules['flash_attn.layers']
  fa.layers.rotary = sys.modules['flash_attn.layers.rotary']
  fa.flash_attn_interface = sys.modules['flash_attn.flash_attn_interface']
  _stub('causal_conv1d', causal_conv1d_fn=None, causal_conv1d_update=None)
  _stub('mamba_ssm')
  _stub('mamba_ssm.ops')
  _stub('mamba_ssm.ops.selective_scan_interface', mamba_inner_fn=None,
        selective_scan_fn=None)
  _stub('mamba_ssm.ops.triton')
  _stub('mamba_ssm.ops.triton.layernorm', RMSNorm=None, layer_norm_fn=None,
        rms_norm_fn=None)
  _stub('mamba_ssm.ops.triton.selective_state_update',
        selective_state_update=None)
  _stub('wandb')


def sha(t):
  return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def make_self(diffusion, noise_schedule, backbone, mask_index, vocab_size,
              length, eval_batch_size, time_conditioning=False):
  ns = types.SimpleNamespace()
  ns.mask_index = mask_index
  ns.vocab_size = vocab_size
  ns.neg_infinity = -1000000.0
  ns.noise = noise_schedule.LogLinearNoise()
  ns.parameterization = 'subs'
  ns.time_conditioning = time_conditioning
  ns.sampler = 'ddpm_cache'
  ns.T = 0
  ns.change_of_variables = False
  ns.importance_sampling = False
  ns.antithetic_sampling = True
  ns.sampling_eps = 1e-3
  ns.device = torch.device('cpu')
  ns.dtype = torch.float32
  ns.backbone = backbone
  ns.config = types.SimpleNamespace(
    noise=types.SimpleNamespace(type='loglinear'),
    loader=types.SimpleNamespace(eval_batch_size=eval_batch_size),
    model=types.SimpleNamespace(length=length),
    sampling=types.SimpleNamespace(steps=16, noise_removal=True))
  D = diffusion.Diffusion
  for name in ('_process_sigma', '_subs_parameterization', 'forward', 'q_xt',
               '_sample_prior', '_ddpm_caching_update', '_sample_t',
               '_maybe_sub_sample', '_forward_pass_diffusion', '_loss',
               '_ddpm_update', 'get_score', '_staggered_score',
               '_transp_transition', '_analytic_update', '_denoiser_update',
               '_d3pm_loss'):
    setattr(ns, name, types.MethodType(getattr(D, name), ns))
  return ns


def main():
  sys.path.insert(0, REF)
  install_stubs()
  import diffusion
  import models.dit as ref_dit
  import noise_schedule
  sys.path.insert(0, HERE)
  import mdlm_oracle as orc

  # utils.print_nans is called in _forward_pass_diffusion (diffusion.py:868)
  os.makedirs(OUT, exist_ok=True)
  torch.set_grad_enabled(False)
  D = diffusion.Diffusion

  # ---------------- G5: LogLinearNoise ----------------
  noise = noise_schedule.LogLinearNoise()
  t = torch.cat([torch.linspace(0, 1, 12), torch.tensor([1e-3, 1e-5, 0.37, 0.999])])
  sig, dsig = noise(t)
  np.savez(os.path.join(OUT, 'g5_noise.npz'), t=t.numpy(), sigma=sig.numpy(),
           dsigma=dsig.numpy(), sigma_max=noise.sigma_max.numpy(),
           sigma_min=np.float32(noise.sigma_min))

  # ---------------- G1: SUBS ----------------
  g1 = {}
  for tag, (B, L, V, mask) in {'small': (2, 8, 64, 63),
                               'midmask': (2, 8, 64, 10)}.items():
    torch.manual_seed(11)
    logits = torch.randn(B, L, V) * 4
    xt = torch.randint(0, V, (B, L))
    xt[torch.rand(B, L) < 0.5] = mask
    xt[0, 0] = mask
    xt[0, 1] = (mask + 1) % V
    ns = types.SimpleNamespace(mask_index=mask, neg_infinity=-1000000.0)
    out = D._subs_parameterization(ns, logits.clone(), xt)
    g1[f'{tag}_logits'] = logits.numpy()
    g1[f'{tag}_xt'] = xt.numpy()
    g1[f'{tag}_out'] = out.numpy()
    g1[f'{tag}_mask'] = np.int64(mask)
    lb = logits.to(torch.bfloat16)
    outb = D._subs_parameterization(ns, lb.clone(), xt)
    g1[f'{tag}_out_bf16'] = outb.float().numpy()
  # full vocab, inputs regenerated from the seed on both sides
  V, mask = 50258, 50257
  torch.manual_seed(12)
  logits = torch.randn(1, 4, V) * 3
  xt = torch.tensor([[mask, 17, mask, 50256]])
  ns = types.SimpleNamespace(mask_index=mask, neg_infinity=-1000000.0)
  out = D._subs_parameterization(ns, logits.clone(), xt)
  probe = torch.tensor([0, 1, 17, 4097, 25000, 50255, 50256, 50257])
  g1['full_seed'] = np.int64(12)
  g1['full_input_sha'] = np.array(sha(logits))
  g1['full_xt'] = xt.numpy()
  g1['full_probe_cols'] = probe.numpy()
  g1['full_probe_vals'] = out[0][:, probe].numpy()
  g1['full_lse'] = torch.logsumexp(
    torch.cat([logits[0, :, :mask], logits[0, :, mask:] - 1e6], -1), -1).numpy()
  g1['full_out_sha'] = np.array(sha(out))
  g1['full_rowsum_exp'] = out[0].exp().sum(-1).numpy()
  np.savez(os.path.join(OUT, 'g1_subs.npz'), **g1)

  # ---------------- G2: sample_categorical / ddpm_cache update ------------
  g2 = {}
  V, mask = 64, 63
  B, L = 3, 8
  torch.manual_seed(21)
  p_x0 = torch.softmax(torch.randn(B, L, V) * 2, -1)
  p_x0[:, :, mask] = 0
  cases = {
    'allmask_t1': (torch.full((B, L), mask), torch.ones(B, 1), 1e-3 * (1 - 1e-5)),
    'partial': (None, torch.full((B, 1), 0.4321), (1 - 1e-5) / 1000),
    'tdt': (None, torch.full((B, 1), (1 - 1e-5) / 128 + 1e-5), (1 - 1e-5) / 128),
    't0_semiar': (None, torch.zeros(B), 0.001),
    't1d': (None, torch.tensor([0.9, 0.5, 0.1]), 0.001),
    'bigdt': (torch.full((B, L), mask), torch.full((B, 1), 0.5), 0.25),
  }
  ns = types.SimpleNamespace(
    mask_index=mask, noise=noise_schedule.LogLinearNoise(),
    config=types.SimpleNamespace(noise=types.SimpleNamespace(type='loglinear')))
  for tag, (x, t, dt) in cases.items():
    if x is None:
      x = torch.randint(0, V - 1, (B, L))
      x[torch.rand(B, L) < 0.6] = mask
    # u: redraw what rand_like(q_xs) will consume
    st = torch.get_rng_state()
    u = torch.rand(B, L, V)
    torch.set_rng_state(st)
    p_out, x_next = D._ddpm_caching_update(ns, x, t, dt, p_x0=p_x0.clone())
    g2[f'{tag}_x'] = x.numpy(); g2[f'{tag}_t'] = t.numpy()
    g2[f'{tag}_dt'] = np.float64(dt); g2[f'{tag}_u'] = u.numpy()
    g2[f'{tag}_x_next'] = x_next.numpy()
  g2['p_x0'] = p_x0.numpy(); g2['mask'] = np.int64(mask)
  # _sample_categorical alone
  torch.manual_seed(22)
  probs = torch.softmax(torch.randn(4, 5, 33), -1)
  st = torch.get_rng_state(); u = torch.rand(4, 5, 33); torch.set_rng_state(st)
  g2['cat_probs'] = probs.numpy(); g2['cat_u'] = u.numpy()
  g2['cat_out'] = diffusion._sample_categorical(probs).numpy()
  # full-V rows regenerated from seeds
  V, mask = 50258, 50257
  torch.manual_seed(23)
  logits = torch.randn(2, 3, V) * 3
  logits[:, :, mask] = -1e6
  p_full = torch.softmax(logits, -1)
  x = torch.tensor([[mask, 5, mask], [mask, mask, 50256]])
  t = torch.tensor([[0.75], [0.02]])
  dt = (1 - 1e-5) / 1000
  st = torch.get_rng_state(); u = torch.rand(2, 3, V); torch.set_rng_state(st)
  ns.mask_index = mask
  _, x_next = D._ddpm_caching_update(ns, x, t, dt, p_x0=p_full.clone())
  g2['full_seed'] = np.int64(23); g2['full_x'] = x.numpy()
  g2['full_t'] = t.numpy(); g2['full_dt'] = np.float64(dt)
  g2['full_x_next'] = x_next.numpy()
  g2['full_p_sha'] = np.array(sha(p_full)); g2['full_u_sha'] = np.array(sha(u))
  np.savez(os.path.join(OUT, 'g2_update.npz'), **g2)

  # ---------------- G3: q_xt ----------------
  torch.manual_seed(31)
  mask = 50257
  x0 = torch.randint(0, 50257, (4, 64))
  mc = torch.tensor([[0.0], [0.3], [0.9], [1.0]])
  st = torch.get_rng_state(); u = torch.rand(4, 64); torch.set_rng_state(st)
  ns = types.SimpleNamespace(mask_index=mask)
  xt = D.q_xt(ns, x0, mc)
  np.savez(os.path.join(OUT, 'g3_qxt.npz'), x0=x0.numpy(), move_chance=mc.numpy(),
           u=u.numpy(), xt=xt.numpy(), mask=np.int64(mask))

  # ---------------- G6: DiT tiny + sub-ops ----------------
  tc = orc.tiny_config()
  V = tc.vocab
  mask = V - 1
  cfg = types.SimpleNamespace(model=types.SimpleNamespace(
    hidden_size=tc.hidden, cond_dim=tc.cond_dim, n_heads=tc.n_heads,
    n_blocks=tc.n_blocks, dropout=0.1, scale_by_sigma=True))
  ref = ref_dit.DIT(cfg, vocab_size=V).eval()
  params = orc.synthetic_dit_params(tc.hidden, tc.n_heads, tc.n_blocks,
                                    tc.cond_dim, V, seed=1)
  sd = ref.state_dict()
  assert set(params) | {'rotary_emb.inv_freq'} == set(sd), (
    set(sd) ^ set(params))
  load = dict(params); load['rotary_emb.inv_freq'] = sd['rotary_emb.inv_freq']
  ref.load_state_dict(load)
  torch.manual_seed(61)
  idx = torch.randint(0, V, (2, tc.length))
  g6 = {'idx': idx.numpy()}
  g6['logits_sigma0'] = ref(idx, torch.zeros(2)).numpy()
  sg = torch.tensor([0.3, 2.0])
  g6['sigma'] = sg.numpy()
  g6['logits_sigma'] = ref(idx, sg).numpy()
  g6['temb_sigma'] = F.silu(ref.sigma_map(sg)).numpy()
  g6['temb_sigma0'] = F.silu(ref.sigma_map(torch.zeros(2))).numpy()
  # sub-ops
  x = torch.randn(2, tc.length, tc.hidden)
  g6['ln_in'] = x.numpy()
  g6['ln_out'] = ref.blocks[0].norm1(x).numpy()
  sh, sc = torch.randn(2, 1, tc.hidden), torch.randn(2, 1, tc.hidden)
  g6['mod_shift'] = sh.numpy(); g6['mod_scale'] = sc.numpy()
  g6['mod_out'] = ref_dit.modulate_fused(ref.blocks[0].norm1(x), sh, sc).numpy()
  cos, sin = ref.rotary_emb(x)
  qkv = torch.randn(2, tc.length, 3, tc.n_heads, tc.hidden // tc.n_heads)
  g6['rot_in'] = qkv.numpy()
  g6['rot_out'] = ref_dit.apply_rotary_pos_emb(qkv.clone(), cos, sin).numpy()
  g6['rot_cos'] = cos[0, :, 0, 0, :cos.shape[-1] // 2].numpy()
  g6['rot_sin'] = sin[0, :, 0, 0, :sin.shape[-1] // 2].numpy()
  c = F.silu(ref.sigma_map(torch.zeros(2)))
  g6['block0_out'] = ref.blocks[0](x, (cos, sin), c, seqlens=None).numpy()
  g6['final_out'] = ref.output_layer(x, c).numpy()
  gate = torch.randn(2, 1, tc.hidden)
  g6['bdas_gate'] = gate.numpy()
  g6['bdas_out'] = ref_dit.bias_dropout_add_scale_fused_inference(
    x, None, gate, x * 0.5, 0.1).numpy()
  np.savez(os.path.join(OUT, 'g6_dit_tiny.npz'), **g6)

  # one DiT-small-shaped block (D=768,H=12,L=128): checksum-sized golden
  cfg2 = types.SimpleNamespace(model=types.SimpleNamespace(
    hidden_size=768, cond_dim=128, n_heads=12, n_blocks=1, dropout=0.1,
    scale_by_sigma=True))
  V2 = 512
  ref2 = ref_dit.DIT(cfg2, vocab_size=V2).eval()
  p2 = orc.synthetic_dit_params(768, 12, 1, 128, V2, seed=2)
  l2 = dict(p2); l2['rotary_emb.inv_freq'] = ref2.state_dict()['rotary_emb.inv_freq']
  ref2.load_state_dict(l2)
  torch.manual_seed(62)
  idx2 = torch.randint(0, V2, (2, 128))
  out2 = ref2(idx2, torch.zeros(2))
  np.savez(os.path.join(OUT, 'g6_dit_small_block.npz'), idx=idx2.numpy(),
           logits_probe=out2[:, ::16, ::32].numpy(),
           logits_mean=out2.mean().numpy(), logits_std=out2.std().numpy())

  # ---------------- G4: NELBO / _loss on the tiny model ----------------
  ns = make_self(diffusion, noise_schedule, ref, mask, V, tc.length, 2)
  diffusion.utils = types.SimpleNamespace(print_nans=lambda *a, **k: None)
  torch.manual_seed(41)
  x0 = torch.randint(0, V - 1, (4, tc.length))
  attn = torch.ones(4, tc.length)
  attn[3, -3:] = 0
  st = torch.get_rng_state()
  u_t = torch.rand(4); u_xt = torch.rand(4, tc.length)
  torch.set_rng_state(st)
  loss = D._loss(ns, x0, attn)
  t_used = orc.sample_t(4, u_t)
  np.savez(os.path.join(OUT, 'g4_nelbo.npz'), x0=x0.numpy(), attn=attn.numpy(),
           u_t=u_t.numpy(), u_xt=u_xt.numpy(), t=t_used.numpy(),
           nlls=loss.nlls.numpy(), token_nll=loss.loss.numpy())
  # _sample_t alone
  torch.manual_seed(42)
  st = torch.get_rng_state(); u8 = torch.rand(8); torch.set_rng_state(st)
  t8 = D._sample_t(types.SimpleNamespace(
    antithetic_sampling=True, sampling_eps=1e-3, importance_sampling=False),
    8, 'cpu')
  np.savez(os.path.join(OUT, 'g4_sample_t.npz'), u=u8.numpy(), t=t8.numpy())

  # ---------------- G7: _sample trajectory (tiny, T=16, B=2) -------------
  trace = []
  orig = D._ddpm_caching_update

  def traced(self, x, t, dt, p_x0=None):
    hit = p_x0 is not None
    p, xn = orig(self, x, t, dt, p_x0=p_x0)
    trace.append((hit, xn.clone()))
    return p, xn
  ns = make_self(diffusion, noise_schedule, ref, mask, V, tc.length, 2)
  ns._ddpm_caching_update = types.MethodType(traced, ns)
  torch.manual_seed(71)
  final = D._sample(ns, num_steps=16)
  np.savez(os.path.join(OUT, 'g7_sample_traj.npz'), seed=np.int64(71),
           steps=np.int64(16),
           cache_hit=np.array([h for h, _ in trace]),
           xs=np.stack([x.numpy() for _, x in trace]),
           final=final.numpy())

  # ---------------- G8: semi-AR (tiny, L=16, stride 8, 2 strides) --------
  ns = make_self(diffusion, noise_schedule, ref, mask, V, tc.length, 2)
  ns.tokenizer = types.SimpleNamespace(
    eos_token_id=V - 2, batch_decode=lambda a: [str(r.tolist()) for r in a])
  torch.manual_seed(81)
  steps, texts, seqlens = D.sample_subs_guidance(
    ns, n_samples=2, stride_length=8, num_strides=2, dt=1 / 8)
  toks = np.array([eval(s) for s in texts[-1]])
  np.savez(os.path.join(OUT, 'g8_semi_ar.npz'), seed=np.int64(81),
           sampling_steps=np.int64(steps), tokens=toks,
           seq_lens=np.asarray(seqlens), eos=np.int64(V - 2))
  # ---------------- G9: 'ddpm' and 'analytic' samplers (tiny model) -------
  g9 = {}
  Bq, Lq = 2, tc.length
  for tc_flag in (False, True):
    tag0 = 'tc' if tc_flag else 'notc'
    ns = make_self(diffusion, noise_schedule, ref, mask, V, Lq, Bq,
                   time_conditioning=tc_flag)
    torch.manual_seed(90)
    x = torch.randint(0, V - 1, (Bq, Lq))
    x[torch.rand(Bq, Lq) < 0.7] = mask
    t = torch.tensor([[0.6], [0.25]])
    dt = (1 - 1e-5) / 16
    g9[f'{tag0}_x'] = x.numpy(); g9[f'{tag0}_t'] = t.numpy()
    g9[f'{tag0}_dt'] = np.float64(dt)
    for name, fn in (('ddpm', lambda: D._ddpm_update(ns, x, t, dt)),
                     ('analytic', lambda: D._analytic_update(ns, x, t, dt)),
                     ('denoiser', lambda: D._denoiser_update(ns, x, t))):
      st = torch.get_rng_state(); u = torch.rand(Bq, Lq, V); torch.set_rng_state(st)
      g9[f'{tag0}_{name}_u'] = u.numpy()
      g9[f'{tag0}_{name}_out'] = fn().numpy()
    for sampler, seed in (('ddpm', 91), ('analytic', 92)):
      ns.sampler = sampler
      torch.manual_seed(seed)
      g9[f'{tag0}_{sampler}_sample_seed'] = np.int64(seed)
      g9[f'{tag0}_{sampler}_sample'] = D._sample(ns, num_steps=16).numpy()
  np.savez(os.path.join(OUT, 'g9_other_samplers.npz'), **g9)

  # ---------------- G10: discrete-time (T > 0) SUBS loss ------------------
  ns = make_self(diffusion, noise_schedule, ref, mask, V, tc.length, 2)
  ns.T = 8
  torch.manual_seed(101)
  x0 = torch.randint(0, V - 1, (8, tc.length))
  attn = torch.ones(8, tc.length)
  st = torch.get_rng_state()
  u_t = torch.rand(8); u_xt = torch.rand(8, tc.length)
  torch.set_rng_state(st)
  loss = D._loss(ns, x0, attn)
  np.savez(os.path.join(OUT, 'g10_d3pm_loss.npz'), x0=x0.numpy(), attn=attn.numpy(),
           u_t=u_t.numpy(), u_xt=u_xt.numpy(), T=np.int64(8),
           nlls=loss.nlls.numpy(), token_nll=loss.loss.numpy())
  # ---------------- G11: training steps (tiny model, dropout 0) -----------
  # training_step (diffusion.py:390-397) -> _loss in train mode, backward,
  # clip_grad_norm_ 1.0 (configs/config.yaml:69), AdamW (449-459), the
  # constant-with-warmup schedule stepped per optimizer step, and the
  # reference's own EMA class (models/ema.py) -- three optimizer steps.
  import models.ema as ref_ema
  import transformers
  cfg0 = types.SimpleNamespace(model=types.SimpleNamespace(
    hidden_size=tc.hidden, cond_dim=tc.cond_dim, n_heads=tc.n_heads,
    n_blocks=tc.n_blocks, dropout=0.0, scale_by_sigma=True))
  ref_t = ref_dit.DIT(cfg0, vocab_size=V)
  ref_t.load_state_dict(load)
  ref_t.train()
  ns = make_self(diffusion, noise_schedule, ref_t, mask, V, tc.length, 4)
  names = [k for k, _ in ref_t.named_parameters()]
  plist = [q for _, q in ref_t.named_parameters()]
  opt = torch.optim.AdamW(plist, lr=3e-3, betas=(0.9, 0.999), eps=1e-8,
                          weight_decay=0)
  sched = transformers.get_constant_schedule_with_warmup(opt, num_warmup_steps=2)
  ema = ref_ema.ExponentialMovingAverage(plist, decay=0.9999)
  keep_keys = ['vocab_embed.embedding', 'sigma_map.mlp.0.weight',
               'blocks.0.norm1.weight', 'blocks.0.attn_qkv.weight',
               'blocks.0.adaLN_modulation.weight', 'blocks.1.attn_out.weight',
               'blocks.1.mlp.0.bias', 'blocks.1.mlp.2.weight',
               'blocks.1.adaLN_modulation.bias',
               'output_layer.norm_final.weight', 'output_layer.linear.weight',
               'output_layer.linear.bias', 'output_layer.adaLN_modulation.weight']
  g11 = {'names': np.array(names), 'keep_keys': np.array(keep_keys)}
  torch.manual_seed(111)
  with torch.enable_grad():
    for it in range(3):
      x0 = torch.randint(0, V - 1, (4, tc.length))
      attn = torch.ones(4, tc.length)
      attn[1, -2:] = 0
      st = torch.get_rng_state()
      u_t = torch.rand(4); u_xt = torch.rand(4, tc.length)
      torch.set_rng_state(st)
      opt.zero_grad()
      loss = D._loss(ns, x0, attn).loss
      loss.backward()
      gn = torch.nn.utils.clip_grad_norm_(plist, 1.0)
      g11[f'lr{it}'] = np.float64(opt.param_groups[0]['lr'])
      g11[f'x0_{it}'] = x0.numpy(); g11[f'attn{it}'] = attn.numpy()
      g11[f'u_t{it}'] = u_t.numpy(); g11[f'u_xt{it}'] = u_xt.numpy()
      g11[f'loss{it}'] = loss.detach().numpy()
      g11[f'grad_norm{it}'] = gn.detach().numpy()
      # gradients after clipping (what AdamW consumes)
      g11[f'gnorms{it}'] = np.array([q.grad.norm().item() for q in plist])
      if it == 0:
        for k in keep_keys:
          g11['grad0/' + k] = dict(zip(names, plist))[k].grad.detach().numpy()
      opt.step()
      sched.step()
      ema.update(plist)
      g11[f'pnorms{it}'] = np.array([q.detach().double().norm().item() for q in plist])
      g11[f'enorms{it}'] = np.array([q.double().norm().item() for q in ema.shadow_params])
  for k in keep_keys:
    i = names.index(k)
    g11['param3/' + k] = plist[i].detach().numpy()
    g11['ema3/' + k] = ema.shadow_params[i].numpy()
  np.savez(os.path.join(OUT, 'g11_train_steps.npz'), **g11)
  torch.set_grad_enabled(False)
  tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
  print('golden written to', OUT, 'bytes', tot)


if __name__ == '__main__':
  main()
