"""GPU parity tests for the sampler-side kernels (a), (a'), (b), (c1), (c2).

Every test calls the HIP kernels through the C-ABI (mdlm_amd.ops -> ctypes)
and checks against the CPU oracle (oracle/mdlm_oracle.py) or the committed
golden fixtures.  Tolerances (BASELINE.json north_star): token ids and
mask/carry-over decisions bit-exact; log-probs 1e-5 (fp32) / 2e-2 (bf16).
"""
import os

import numpy as np
import pytest
import torch

import mdlm_oracle as orc
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

V_OWT, MASK_OWT = 50258, 50257


def _g(name):
  return np.load(os.path.join(GOLDEN, name))


def _t(a):
  return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope='module')
def ops():
  assert torch.cuda.is_available(), 'GPU tests need a MI355X'
  from mdlm_amd import ops as _ops
  return _ops


def _close_bf16(a, b):
  """|a-b| <= 2e-2 * max(1, |b|) (north_star bf16 tolerance)."""
  a, b = a.double(), b.double()
  return bool(((a - b).abs() <= 2e-2 * b.abs().clamp(min=1.0)).all())


# ------------------------------------------------------------------ (a) SUBS
def test_subs_logprobs_golden_fp32(ops):
  g = _g('g1_subs.npz')
  for tag in ('small', 'midmask'):
    logits = _t(g[f'{tag}_logits']).cuda()
    xt = _t(g[f'{tag}_xt']).cuda()
    out = ops.subs_logprobs(logits, xt, int(g[f'{tag}_mask']))
    ref = _t(g[f'{tag}_out'])
    assert (out.cpu() - ref).abs().max().item() <= 1e-5, tag
    # bf16 in / bf16 out
    outb = ops.subs_logprobs(logits.bfloat16(), xt, int(g[f'{tag}_mask']))
    assert outb.dtype == torch.bfloat16
    ref32 = orc.subs_parameterization(
      logits.bfloat16().float().cpu(), xt.cpu(), int(g[f'{tag}_mask']))
    assert _close_bf16(outb.float().cpu(), ref32), tag
    # vs the reference's own bf16 arithmetic: it rounds lse to bf16 (1 ulp =
    # 0.0625 for |lse| in [8,16)) before subtracting; the kernel keeps lse in
    # fp32, so allow that one ulp on top.
    d = (outb.float().cpu() - _t(g[f'{tag}_out_bf16'])).abs()
    assert (d <= 0.0625 + 2e-2 * _t(g[f'{tag}_out_bf16']).abs()).all(), tag


@pytest.mark.parametrize('aligned', [True, False])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_subs_logprobs_full_vocab(ops, aligned, dtype):
  torch.manual_seed(5)
  B, L = 2, 6
  logits = (torch.randn(B, L, V_OWT) * 3).to(dtype)
  xt = torch.full((B, L), MASK_OWT)
  xt[0, 1] = 17; xt[1, 5] = 50256; xt[1, 0] = 0
  ref = orc.subs_parameterization(logits.float().clone(), xt, MASK_OWT)
  if aligned:
    dl = ops.alloc_logits(B, L, V_OWT, dtype)
    dl.copy_(logits)
  else:
    dl = logits.cuda()
  out = ops.subs_logprobs(dl, xt.cuda(), MASK_OWT)
  if dtype == torch.float32:
    assert (out.cpu() - ref).abs().max().item() <= 1e-5
  else:
    assert _close_bf16(out.float().cpu(), ref)
  # size-independent property: masked rows are normalised distributions
  masked = (xt == MASK_OWT)
  s = out.float().exp().sum(-1).cpu()
  tol = 1e-4 if dtype == torch.float32 else 2e-2
  assert (s[masked] - 1).abs().max().item() < tol
  # unmasked rows: -1e6 everywhere, 0 at xt
  o = out.float().cpu()
  for b, l in [(0, 1), (1, 5), (1, 0)]:
    row = o[b, l]
    assert row[xt[b, l]] == 0
    assert (row.sum() - row[xt[b, l]]).item() < -1e9


def test_subs_row_lse_and_argmax(ops):
  torch.manual_seed(6)
  B, L = 2, 5
  logits = torch.randn(B, L, V_OWT) * 3
  logits[0, 0, MASK_OWT] = 50.0  # mask column must never win
  xt = torch.full((B, L), MASK_OWT)
  xt[1, 2] = 123
  ref = orc.subs_parameterization(logits.clone(), xt, MASK_OWT)
  sh = logits.clone(); sh[:, :, MASK_OWT] += orc.NEG_INFINITY
  lse_ref = torch.logsumexp(sh, -1)
  for dt in (torch.float32, torch.bfloat16):
    dl = logits.to(dt).cuda()
    lse = ops.subs_row_lse(dl, xt.cuda(), MASK_OWT).cpu().view(B, L)
    m = xt == MASK_OWT
    tol = 1e-5 if dt == torch.float32 else 2e-2
    assert (lse[m] - lse_ref[m]).abs().max().item() <= tol
    assert (lse[~m] == 0).all()
    am = ops.subs_argmax(dl, xt.cuda(), MASK_OWT).cpu()
    ref_am = orc.subs_parameterization(logits.to(dt).float(), xt, MASK_OWT).argmax(-1)
    assert torch.equal(am, ref_am)
  assert torch.equal(ops.subs_argmax(logits.cuda(), xt.cuda(), MASK_OWT).cpu(),
                     ref.argmax(-1))


def test_subs_argmax_ties_first_index(ops):
  V, mask = 1000, 999
  logits = torch.zeros(3, V)
  logits[1, 500] = 1.0; logits[1, 700] = 1.0
  logits[2, 998] = 2.0
  xt = torch.full((1, 3), mask)
  am = ops.subs_argmax(logits.view(1, 3, V).cuda(), xt.cuda(), mask).cpu()
  assert am.tolist() == [[0, 500, 998]]


# ------------------------------------------------- (b) ddpm_cache update
def test_update_golden_supplied_u(ops):
  g = _g('g2_update.npz')
  mask = int(g['mask'])
  p = _t(g['p_x0']).cuda()
  for tag in ('allmask_t1', 'partial', 'tdt', 't0_semiar', 't1d', 'bigdt'):
    x = _t(g[f'{tag}_x']).cuda()
    t = _t(g[f'{tag}_t']).cuda()
    changed = torch.full((x.shape[0],), -7, dtype=torch.int32, device='cuda')
    xn = ops.ddpm_cache_update(p, x, t, float(g[f'{tag}_dt']), mask,
                               u=_t(g[f'{tag}_u']).cuda(), changed=changed)
    ref = _t(g[f'{tag}_x_next'])
    assert torch.equal(xn.cpu(), ref), tag
    ch_ref = (ref != x.cpu()).any(-1).int()
    assert torch.equal(changed.cpu(), ch_ref), tag


def test_update_golden_full_vocab(ops):
  g = _g('g2_update.npz')
  torch.manual_seed(int(g['full_seed']))
  logits = torch.randn(2, 3, V_OWT) * 3
  logits[:, :, MASK_OWT] = -1e6
  p = torch.softmax(logits, -1)
  u = torch.rand(2, 3, V_OWT)
  xn = ops.ddpm_cache_update(p.cuda(), _t(g['full_x']).cuda(),
                             _t(g['full_t']).cuda(), float(g['full_dt']),
                             MASK_OWT, u=u.cuda())
  assert torch.equal(xn.cpu(), _t(g['full_x_next']))


def _near_tie_only(q_over_g, got, ref, rel=1e-5):
  """Mismatches are tolerated only where the two candidates are within `rel`."""
  bad = (got != ref)
  if not bad.any():
    return True
  a = q_over_g[bad].gather(-1, got[bad][:, None]).squeeze(-1)
  b = q_over_g[bad].gather(-1, ref[bad][:, None]).squeeze(-1)
  return bool(((a - b).abs() <= rel * b.abs()).all())


@pytest.mark.parametrize('aligned', [True, False])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_update_philox_matches_oracle(ops, aligned, dtype):
  """In-kernel Philox path == oracle fed the same uniforms (bit-exact)."""
  torch.manual_seed(7)
  B, L = 3, 4
  logits = torch.randn(B, L, V_OWT) * 3
  logits[:, :, MASK_OWT] = -1e6
  p = torch.softmax(logits, -1).to(dtype)
  x = torch.full((B, L), MASK_OWT)
  x[0, 0] = 11; x[2, 3] = 50256
  t = torch.tensor([0.9, 0.3, 0.004])
  dt = (1 - 1e-5) / 1000
  seed, off = 1234, 77
  u = ops.philox_uniform(B * L, V_OWT, seed, off)
  assert 0 <= u.min().item() and u.max().item() < 1
  assert abs(u.mean().item() - 0.5) < 2e-3
  _, ref = orc.ddpm_caching_update(None, x, t, dt, u.cpu().view(B, L, V_OWT),
                                   MASK_OWT, p_x0=p.float().clone())
  if aligned:
    dp = ops.alloc_logits(B, L, V_OWT, dtype); dp.copy_(p)
  else:
    dp = p.cuda()
  changed = torch.zeros(B, dtype=torch.int32, device='cuda')
  xn = ops.ddpm_cache_update(dp, x.cuda(), t.cuda(), dt, MASK_OWT, seed=seed,
                             offset=off, changed=changed)
  assert torch.equal(xn.cpu(), ref)
  assert torch.equal(changed.cpu(), (ref != x).any(-1).int())
  # supplied-u path gives the same answer
  xn2 = ops.ddpm_cache_update(dp, x.cuda(), t.cuda(), dt, MASK_OWT, u=u)
  assert torch.equal(xn2.cpu(), ref)
  # fast-math path: identical except at near-ties
  xf = ops.ddpm_cache_update(dp, x.cuda(), t.cuda(), dt, MASK_OWT, seed=seed,
                             offset=off, fast_math=True)
  tt = t[:, None, None]
  q = p.float() * (tt - (tt - dt)); q[:, :, MASK_OWT] = (tt - dt)[:, :, 0]
  r = q / (1e-10 - (u.cpu().view(B, L, V_OWT) + 1e-10).log())
  assert _near_tie_only(r, xf.cpu(), ref)


def test_update_unmasks_at_expected_rate(ops):
  """Statistical check of the in-kernel RNG path: P(unmask) = dt/t."""
  B, L, V, mask = 8, 512, 4096, 4095
  torch.manual_seed(8)
  logits = torch.randn(B, L, V); logits[:, :, mask] = -1e6
  p = torch.softmax(logits, -1).cuda()
  x = torch.full((B, L), mask).cuda()
  t = torch.full((B,), 0.5).cuda()
  xn = ops.ddpm_cache_update(p, x, t, 0.1, mask, seed=3, offset=0)
  frac = (xn != mask).float().mean().item()
  assert abs(frac - 0.2) < 0.03, frac
  xn2 = ops.ddpm_cache_update(p, x, t, 0.1, mask, seed=3, offset=0)
  assert torch.equal(xn, xn2)  # deterministic in (seed, offset)
  xn3 = ops.ddpm_cache_update(p, x, t, 0.1, mask, seed=3, offset=10**6)
  assert not torch.equal(xn, xn3)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_update_from_logits(ops, dtype):
  """(a)+(b) fused: logits -> x_next equals oracle SUBS -> exp -> update."""
  torch.manual_seed(9)
  B, L = 2, 6
  logits = (torch.randn(B, L, V_OWT) * 3).to(dtype)
  x = torch.full((B, L), MASK_OWT)
  x[0, 2] = 42; x[1, 0] = 7
  t = torch.tensor([[0.8], [0.05]])
  dt = (1 - 1e-5) / 1000
  u = torch.rand(B, L, V_OWT)
  logp = orc.subs_parameterization(logits.float().clone(), x, MASK_OWT)
  _, ref = orc.ddpm_caching_update(None, x, t, dt, u, MASK_OWT, p_x0=logp.exp())
  dl = ops.alloc_logits(B, L, V_OWT, dtype); dl.copy_(logits)
  lse = ops.subs_row_lse(dl, x.cuda(), MASK_OWT)
  xn = ops.subs_ddpm_update(dl, lse, x.cuda(), t.cuda(), dt, MASK_OWT, u=u.cuda())
  tt = t.squeeze(-1)[:, None, None]
  q = logp.exp() * (tt - (tt - dt)); q[:, :, MASK_OWT] = (tt - dt)[:, :, 0]
  r = q / (1e-10 - (u + 1e-10).log())
  assert _near_tie_only(r, xn.cpu(), ref)
  assert torch.equal(xn.cpu()[x != MASK_OWT], x[x != MASK_OWT])
  # philox variant against oracle with the materialised draw
  up = ops.philox_uniform(B * L, V_OWT, 99, 5)
  _, refp = orc.ddpm_caching_update(None, x, t, dt, up.cpu().view(B, L, V_OWT),
                                    MASK_OWT, p_x0=logp.exp())
  xp = ops.subs_ddpm_update(dl, lse, x.cuda(), t.cuda(), dt, MASK_OWT, seed=99,
                            offset=5)
  rp = q / (1e-10 - (up.cpu().view(B, L, V_OWT) + 1e-10).log())
  assert _near_tie_only(rp, xp.cpu(), refp)


@pytest.mark.parametrize('fast', [False, True])
def test_fused_update_computes_lse_for_fresh_samples(ops, fast):
  """`fresh` samples get their row lse computed inside the update kernel:
  identical tokens and identical lse to the two-kernel sequence."""
  torch.manual_seed(11)
  B, L = 3, 5
  dl = ops.alloc_logits(B, L, V_OWT, torch.bfloat16)
  dl.copy_((torch.randn(B, L, V_OWT) * 3).bfloat16())
  x = torch.full((B, L), MASK_OWT); x[1, 1] = 9
  t = torch.tensor([0.6, 0.3, 0.9]).cuda()
  lse_ref = ops.subs_row_lse(dl, x.cuda(), MASK_OWT)
  ref = ops.subs_ddpm_update(dl, lse_ref, x.cuda(), t, 0.2, MASK_OWT, seed=4, offset=8,
                             fast_math=fast)
  # sample 1 is not fresh: it must keep using the lse that is already there
  lse = torch.full_like(lse_ref, float('nan'))
  lse[L:2 * L] = lse_ref[L:2 * L]
  fresh = torch.tensor([1, 0, 1], dtype=torch.int32).cuda()
  got = ops.subs_ddpm_update(dl, lse, x.cuda(), t, 0.2, MASK_OWT, seed=4, offset=8,
                             fast_math=fast, fresh=fresh)
  assert torch.equal(got, ref)
  m = (x == MASK_OWT).view(-1).cuda()
  assert torch.equal(lse[m], lse_ref[m])


# ------------------------------------------------------------ (c1) q_xt
def test_q_xt_golden_and_philox(ops):
  g = _g('g3_qxt.npz')
  x0 = _t(g['x0']).cuda()
  xt = ops.q_xt(x0, _t(g['move_chance']).cuda(), int(g['mask']),
                u=_t(g['u']).cuda())
  assert torch.equal(xt.cpu(), _t(g['xt']))
  B, L = x0.shape
  up = ops.philox_uniform(1, B * L, 5, 9).view(B, L)
  xp = ops.q_xt(x0, _t(g['move_chance']).cuda(), int(g['mask']), seed=5, offset=9)
  ref = orc.q_xt(x0.cpu(), _t(g['move_chance']), up.cpu(), int(g['mask']))
  assert torch.equal(xp.cpu(), ref)
  # move_chance 0 -> nothing masked; 1 -> everything masked
  assert torch.equal(xp[0].cpu(), x0[0].cpu())
  assert (xp[3] == int(g['mask'])).all()


# -------------------------------------------------------- (c2) SUBS NLL
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_subs_nll_matches_oracle(ops, dtype):
  torch.manual_seed(10)
  B, L = 3, 8
  logits = (torch.randn(B, L, V_OWT) * 3).to(dtype)
  x0 = torch.randint(0, 50257, (B, L))
  t = orc.sample_t(B, torch.rand(B))
  sigma, dsigma = orc.loglinear_noise(t)
  xt = orc.q_xt(x0, 1 - torch.exp(-sigma[:, None]), torch.rand(B, L), MASK_OWT)
  xt[0, 0] = MASK_OWT; xt[0, 1] = x0[0, 1]
  attn = torch.ones(B, L); attn[2, -2:] = 0
  nlls_ref, token_ref = orc.nelbo_from_logits(logits.float(), xt, x0, t, attn,
                                              MASK_OWT)
  w = dsigma / torch.expm1(sigma)
  nll = ops.subs_nll(logits.cuda(), xt.cuda(), x0.cuda(), w.cuda(), MASK_OWT,
                     attn_mask=attn.cuda())
  tol = 1e-5 if dtype == torch.float32 else 2e-2
  err = (nll.cpu() - nlls_ref).abs() / nlls_ref.abs().clamp(min=1.0)
  assert err.max().item() <= tol
  sums = ops.nll_sums(nll, attn.cuda()).cpu()
  assert abs(sums[0].item() - nlls_ref.double().sum().item()) <= 1e-4 * max(1.0, abs(sums[0].item()))
  assert sums[1].item() == attn.sum().item()
  ppl = orc.perplexity(sums[0], sums[1])
  assert abs(ppl / torch.exp(token_ref).item() - 1) < 5e-3


def test_compact_active(ops):
  changed = torch.tensor([0, 1, 1, 0, 0, 1, 0, 1] * 9 + [1, 0, 0], dtype=torch.int32).cuda()
  B = changed.numel()
  active = torch.full((B,), -1, dtype=torch.int32, device='cuda')
  n = torch.zeros(1, dtype=torch.int32, device='cuda')
  nr = torch.zeros(1, dtype=torch.int32, device='cuda')
  ops.compact_active(changed, active, n, nr, 16)
  ref = torch.nonzero(changed.cpu()).flatten().int()
  assert n.item() == ref.numel() and nr.item() == 16 * ref.numel()
  assert torch.equal(active.cpu()[:ref.numel()], ref)


def test_errors_are_loud(ops):
  x = torch.zeros(2, 4, 8)
  with pytest.raises(RuntimeError):
    ops.subs_logprobs(x, torch.zeros(2, 4, dtype=torch.int64), 7)  # CPU tensor
  with pytest.raises(RuntimeError):
    ops.subs_logprobs(x.cuda(), torch.zeros(2, 4, dtype=torch.int64).cuda(), 8)


def test_edge_shapes(ops):
  """Empty batches, one row, tiny unaligned vocabularies, nothing / everything
  masked: every entry point must match the oracle (or be a no-op)."""
  # empty batch: no launch, no error
  e = torch.zeros(0, 8, 16, device='cuda')
  ex = torch.zeros(0, 8, dtype=torch.int64, device='cuda')
  assert ops.subs_logprobs(e, ex, 15).shape == (0, 8, 16)
  assert ops.subs_argmax(e, ex, 15).shape == (0, 8)
  assert ops.q_xt(ex, torch.zeros(0, device='cuda'), 15).shape == (0, 8)
  assert ops.ddpm_cache_update(e, ex, torch.zeros(0, device='cuda'), 0.1, 15).shape == (0, 8)
  for (B, L, Vv) in [(1, 1, 5), (2, 3, 7), (1, 5, 130), (3, 2, 1000)]:
    mask = Vv - 1
    torch.manual_seed(B * 100 + Vv)
    for frac in (0.0, 0.5, 1.0):
      logits = torch.randn(B, L, Vv) * 3
      x = torch.randint(0, Vv - 1, (B, L))
      x[torch.rand(B, L) < frac] = mask
      if frac == 1.0:
        x[:] = mask
      ref = orc.subs_parameterization(logits.clone(), x, mask)
      got = ops.subs_logprobs(logits.cuda(), x.cuda(), mask).cpu()
      torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
      assert torch.equal(ops.subs_argmax(logits.cuda(), x.cuda(), mask).cpu(), ref.argmax(-1))
      p = ref.exp()
      t = torch.rand(B, 1) * 0.8 + 0.2
      u = torch.rand(B, L, Vv)
      _, xn_ref = orc.ddpm_caching_update(None, x, t, 0.1, u, mask, p_x0=p.clone())
      xn = ops.ddpm_cache_update(p.cuda(), x.cuda(), t.cuda(), 0.1, mask, u=u.cuda()).cpu()
      assert torch.equal(xn, xn_ref), (B, L, Vv, frac)
      # fused from logits + generic race with the ddpm_cache weights: same tokens
      lse = torch.zeros(B * L, device='cuda')
      fresh = torch.ones(B, dtype=torch.int32, device='cuda')
      xn2 = ops.subs_ddpm_update(logits.cuda(), lse, x.cuda(), t.cuda(), 0.1, mask, u=u.cuda(), fresh=fresh).cpu()
      tt = t.squeeze(-1)
      xn3 = ops.subs_race_update(logits.cuda(), lse, x.cuda(), (tt - (tt - 0.1)).cuda(), (tt - 0.1).cuda(),
                                 mask, u=u.cuda(), fresh=fresh).cpu()
      assert torch.equal(xn2, xn3)
      assert torch.equal(xn2[x != mask], x[x != mask])
      # q_xt and the NLL on the same tiny shapes
      mc = torch.rand(B, 1)
      uq = torch.rand(B, L)
      assert torch.equal(ops.q_xt(x.cuda(), mc.cuda(), mask, u=uq.cuda()).cpu(), orc.q_xt(x, mc, uq, mask))
