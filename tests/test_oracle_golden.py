"""Pin the CPU oracle (oracle/mdlm_oracle.py) against golden vectors produced
by the reference itself (oracle/gen_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import torch

import mdlm_oracle as orc
from conftest import GOLDEN


def _g(name):
  return np.load(os.path.join(GOLDEN, name))


def _t(a):
  return torch.from_numpy(np.asarray(a))


def _sha(t):
  return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def test_g5_loglinear_noise():
  g = _g('g5_noise.npz')
  t = _t(g['t'])
  s, ds = orc.loglinear_noise(t)
  assert torch.equal(s, _t(g['sigma']))
  assert torch.equal(ds, _t(g['dsigma']))
  assert torch.equal(orc.loglinear_total_noise(torch.tensor(1.0)),
                     _t(g['sigma_max']))


def test_g1_subs_small_exact():
  g = _g('g1_subs.npz')
  for tag in ('small', 'midmask'):
    out = orc.subs_parameterization(_t(g[f'{tag}_logits']).clone(),
                                    _t(g[f'{tag}_xt']), int(g[f'{tag}_mask']))
    assert torch.equal(out, _t(g[f'{tag}_out'])), tag
    outb = orc.subs_parameterization(
      _t(g[f'{tag}_logits']).to(torch.bfloat16), _t(g[f'{tag}_xt']),
      int(g[f'{tag}_mask']))
    assert torch.equal(outb.float(), _t(g[f'{tag}_out_bf16'])), tag


def test_g1_subs_full_vocab():
  g = _g('g1_subs.npz')
  V, mask = 50258, 50257
  torch.manual_seed(int(g['full_seed']))
  logits = torch.randn(1, 4, V) * 3
  assert _sha(logits) == str(g['full_input_sha'])
  out = orc.subs_parameterization(logits.clone(), _t(g['full_xt']), mask)
  assert _sha(out) == str(g['full_out_sha'])
  probe = _t(g['full_probe_cols'])
  assert torch.equal(out[0][:, probe], _t(g['full_probe_vals']))


def test_g2_update_cases():
  g = _g('g2_update.npz')
  mask = int(g['mask'])
  p = _t(g['p_x0'])
  for tag in ('allmask_t1', 'partial', 'tdt', 't0_semiar', 't1d', 'bigdt'):
    x, t, dt = _t(g[f'{tag}_x']), _t(g[f'{tag}_t']), float(g[f'{tag}_dt'])
    _, xn = orc.ddpm_caching_update(None, x, t, dt, _t(g[f'{tag}_u']), mask,
                                    p_x0=p.clone())
    assert torch.equal(xn, _t(g[f'{tag}_x_next'])), tag
    # carry-over: unmasked tokens never change
    keep = x != mask
    assert torch.equal(xn[keep], x[keep])
  out = orc.sample_categorical(_t(g['cat_probs']), _t(g['cat_u']))
  assert torch.equal(out, _t(g['cat_out']))


def test_g2_update_full_vocab():
  g = _g('g2_update.npz')
  V, mask = 50258, 50257
  torch.manual_seed(int(g['full_seed']))
  logits = torch.randn(2, 3, V) * 3
  logits[:, :, mask] = -1e6
  p = torch.softmax(logits, -1)
  u = torch.rand(2, 3, V)
  assert _sha(p) == str(g['full_p_sha']) and _sha(u) == str(g['full_u_sha'])
  _, xn = orc.ddpm_caching_update(None, _t(g['full_x']), _t(g['full_t']),
                                  float(g['full_dt']), u, mask, p_x0=p)
  assert torch.equal(xn, _t(g['full_x_next']))


def test_g3_q_xt():
  g = _g('g3_qxt.npz')
  xt = orc.q_xt(_t(g['x0']), _t(g['move_chance']), _t(g['u']), int(g['mask']))
  assert torch.equal(xt, _t(g['xt']))


def _tiny():
  tc = orc.tiny_config()
  p = orc.synthetic_dit_params(tc.hidden, tc.n_heads, tc.n_blocks, tc.cond_dim,
                               tc.vocab, seed=1)
  return tc, p


def test_g6_dit_tiny():
  g = _g('g6_dit_tiny.npz')
  tc, p = _tiny()
  idx = _t(g['idx'])
  out0 = orc.dit_forward(p, idx, torch.zeros(2), tc.n_heads, tc.n_blocks)
  torch.testing.assert_close(out0, _t(g['logits_sigma0']), rtol=1e-5, atol=2e-6)
  out1 = orc.dit_forward(p, idx, _t(g['sigma']), tc.n_heads, tc.n_blocks)
  torch.testing.assert_close(out1, _t(g['logits_sigma']), rtol=1e-5, atol=2e-6)
  # sub-ops
  x = _t(g['ln_in'])
  ln = orc.layer_norm_w(x, p['blocks.0.norm1.weight'])
  torch.testing.assert_close(ln, _t(g['ln_out']), rtol=1e-6, atol=1e-6)
  mod = orc.modulate(ln, _t(g['mod_shift']), _t(g['mod_scale']))
  torch.testing.assert_close(mod, _t(g['mod_out']), rtol=1e-6, atol=1e-6)
  cos, sin = orc.rotary_cos_sin(tc.length, tc.hidden // tc.n_heads)
  torch.testing.assert_close(cos, _t(g['rot_cos']))
  torch.testing.assert_close(sin, _t(g['rot_sin']))
  rot = orc.apply_rotary_qk(_t(g['rot_in']), cos, sin)
  torch.testing.assert_close(rot, _t(g['rot_out']), rtol=1e-6, atol=1e-6)
  # conditioning vector at sigma = 0 is sincos(0) = [1]*128 ++ [0]*128 -> MLP
  import torch.nn.functional as F
  te = orc.timestep_embedding(torch.zeros(2), 256)
  assert torch.equal(te[:, :128], torch.ones(2, 128))
  assert torch.equal(te[:, 128:], torch.zeros(2, 128))
  h = F.linear(te, p['sigma_map.mlp.0.weight'], p['sigma_map.mlp.0.bias'])
  c = F.silu(F.linear(F.silu(h), p['sigma_map.mlp.2.weight'],
                      p['sigma_map.mlp.2.bias']))
  torch.testing.assert_close(c, _t(g['temb_sigma0']), rtol=1e-6, atol=1e-6)
  blk = orc.dit_block(p, 'blocks.0.', x, cos, sin, c, tc.n_heads)
  torch.testing.assert_close(blk, _t(g['block0_out']), rtol=1e-5, atol=2e-6)


def test_g6_dit_small_block_shape():
  g = _g('g6_dit_small_block.npz')
  p = orc.synthetic_dit_params(768, 12, 1, 128, 512, seed=2)
  out = orc.dit_forward(p, _t(g['idx']), torch.zeros(2), 12, 1)
  torch.testing.assert_close(out[:, ::16, ::32], _t(g['logits_probe']),
                             rtol=1e-4, atol=1e-5)


def test_g4_nelbo_and_sample_t():
  g = _g('g4_sample_t.npz')
  t = orc.sample_t(8, _t(g['u']))
  assert torch.equal(t, _t(g['t']))
  g = _g('g4_nelbo.npz')
  tc, p = _tiny()
  od = orc.OracleDiffusion(p, tc.n_heads, tc.n_blocks, tc.vocab, tc.vocab - 1,
                           tc.length)
  t = orc.sample_t(4, _t(g['u_t']))
  assert torch.equal(t, _t(g['t']))
  token_nll, nlls, _ = od._loss(_t(g['x0']), _t(g['attn']), t=t,
                                u_xt=_t(g['u_xt']))
  torch.testing.assert_close(nlls, _t(g['nlls']), rtol=1e-5, atol=1e-5)
  torch.testing.assert_close(token_nll, _t(g['token_nll']), rtol=1e-5, atol=1e-6)


def test_g7_sample_trajectory():
  g = _g('g7_sample_traj.npz')
  tc, p = _tiny()
  od = orc.OracleDiffusion(p, tc.n_heads, tc.n_blocks, tc.vocab, tc.vocab - 1,
                           tc.length)
  torch.manual_seed(int(g['seed']))
  trace = []
  final = od._sample(2, int(g['steps']), trace=trace)
  assert torch.equal(torch.stack(trace), _t(g['xs']))
  assert torch.equal(final, _t(g['final']))
  # NFE = cache misses + noise-removal call
  assert od.nfe == int((~g['cache_hit']).sum()) + 1


def test_g8_semi_ar():
  g = _g('g8_semi_ar.npz')
  tc, p = _tiny()
  od = orc.OracleDiffusion(p, tc.n_heads, tc.n_blocks, tc.vocab, tc.vocab - 1,
                           tc.length)
  torch.manual_seed(int(g['seed']))
  steps, toks = od.sample_subs_guidance(2, 8, 2, dt=1 / 8)
  assert steps == int(g['sampling_steps'])
  assert torch.equal(toks, _t(g['tokens']))


def test_g9_other_samplers():
  """'ddpm' / 'analytic' samplers (diffusion.py:612-637, 721-798)."""
  g = _g('g9_other_samplers.npz')
  tc, p = _tiny()
  V, mask = tc.vocab, tc.vocab - 1
  for tag0, tcond in (('notc', False), ('tc', True)):
    od = orc.OracleDiffusion(p, tc.n_heads, tc.n_blocks, V, mask, tc.length,
                             time_conditioning=tcond)
    x, t, dt = _t(g[f'{tag0}_x']), _t(g[f'{tag0}_t']), float(g[f'{tag0}_dt'])
    out = orc.ddpm_update(od.forward, x, t, dt, _t(g[f'{tag0}_ddpm_u']), mask)
    assert torch.equal(out, _t(g[f'{tag0}_ddpm_out'])), tag0
    out = orc.analytic_update(od.forward, x, t, dt, _t(g[f'{tag0}_analytic_u']),
                              mask, V)
    assert torch.equal(out, _t(g[f'{tag0}_analytic_out'])), tag0
    out = orc.denoiser_update(od.forward, x, t, _t(g[f'{tag0}_denoiser_u']),
                              mask, V)
    assert torch.equal(out, _t(g[f'{tag0}_denoiser_out'])), tag0
    for sampler in ('ddpm', 'analytic'):
      od = orc.OracleDiffusion(p, tc.n_heads, tc.n_blocks, V, mask, tc.length,
                               time_conditioning=tcond, sampler=sampler)
      torch.manual_seed(int(g[f'{tag0}_{sampler}_sample_seed']))
      final = od._sample(2, 16)
      assert torch.equal(final, _t(g[f'{tag0}_{sampler}_sample'])), (tag0, sampler)


def test_g10_d3pm_loss():
  """Discrete-time (T > 0) SUBS loss (diffusion.py:329-358, 849-853, 874-881)."""
  g = _g('g10_d3pm_loss.npz')
  tc, p = _tiny()
  od = orc.OracleDiffusion(p, tc.n_heads, tc.n_blocks, tc.vocab, tc.vocab - 1,
                           tc.length, T=int(g['T']))
  t = orc.sample_t(8, _t(g['u_t']))
  token_nll, nlls, _ = od._loss(_t(g['x0']), _t(g['attn']), t=t,
                                u_xt=_t(g['u_xt']))
  torch.testing.assert_close(nlls, _t(g['nlls']), rtol=1e-5, atol=1e-5,
                             equal_nan=True)
  torch.testing.assert_close(token_nll, _t(g['token_nll']), rtol=1e-5,
                             atol=1e-6, equal_nan=True)


def test_g11_training_steps():
  """Three optimizer steps of the reference (training_step -> backward ->
  clip_grad_norm_ 1.0 -> AdamW -> warmup schedule -> models/ema.py) against the
  oracle's restatement."""
  g = _g('g11_train_steps.npz')
  tc, p = _tiny()
  names = [str(n) for n in g['names']]
  assert set(names) == set(p)
  keep = [str(k) for k in g['keep_keys']]
  st = orc.new_train_state(p)
  for it in range(3):
    lr = orc.constant_warmup_lr(3e-3, it, num_warmup_steps=2)
    assert abs(lr - float(g[f'lr{it}'])) < 1e-12
    t = orc.sample_t(4, _t(g[f'u_t{it}']))
    loss, total, grads = orc.training_step(
      st, _t(g[f'x0_{it}']), _t(g[f'attn{it}']), t, _t(g[f'u_xt{it}']),
      tc.n_heads, tc.n_blocks, tc.vocab - 1, lr=lr)
    torch.testing.assert_close(loss, _t(g[f'loss{it}']), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(total, _t(g[f'grad_norm{it}']), rtol=1e-4, atol=1e-6)
    _, coef = orc.clip_coef(grads)
    gn = torch.tensor([float((grads[n] * coef).norm()) for n in names])
    torch.testing.assert_close(gn, _t(g[f'gnorms{it}']).float(), rtol=2e-4, atol=1e-7)
    if it == 0:
      for k in keep:
        torch.testing.assert_close(grads[k] * coef, _t(g['grad0/' + k]),
                                   rtol=1e-4, atol=1e-7)
    pn = torch.tensor([float(st['params'][n].double().norm()) for n in names])
    en = torch.tensor([float(st['shadow'][n].double().norm()) for n in names])
    torch.testing.assert_close(pn, _t(g[f'pnorms{it}']).float(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(en, _t(g[f'enorms{it}']).float(), rtol=1e-5, atol=1e-7)
  for k in keep:
    torch.testing.assert_close(st['params'][k], _t(g['param3/' + k]), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(st['shadow'][k], _t(g['ema3/' + k]), rtol=1e-5, atol=1e-6)
