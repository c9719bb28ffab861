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


def _exact_log_ok(u):
  """True where torch's CPU fp32 log agrees with the correctly rounded log (the
  defining arithmetic of the race, include/mdlm_hip.h) at u + 1e-10: all but
  ~0.1 % of the uniforms.  Constructed near-ties use only such columns, so the
  oracle's last bit is not in question."""
  s = u + 1e-10
  return s.log() == s.double().log().float()


@pytest.mark.parametrize('aligned', [True, False])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_update_philox_matches_oracle(ops, aligned, dtype):
  """In-kernel Philox path == oracle fed the same uniforms, bit-exact, for the
  reference kernel (fast_math=False), the supplied-u path and the fast path
  (fast_math=True: approximate ranking + exact re-evaluation of near-ties)."""
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
  # fast path: identical token ids, no tolerance
  changed.zero_()
  xf = ops.ddpm_cache_update(dp, x.cuda(), t.cuda(), dt, MASK_OWT, seed=seed,
                             offset=off, fast_math=True, changed=changed)
  assert torch.equal(xf.cpu(), ref)
  assert torch.equal(changed.cpu(), (ref != x).any(-1).int())


def test_race_key_error_exhaustive(ops):
  """The error budget of the fast path (csrc/sampler.hip), measured over ALL
  2^24 uniforms the kernels can draw: the approximate pass's log2(g~) against
  log2 of the defining g = 1e-10f - RN(ln(u + 1e-10f)), and g~ against g; and
  the device's defining g against the CPU's correctly rounded one."""
  k = torch.arange(1 << 24, dtype=torch.int64)
  rbits = (k << 8).to(torch.int32).cuda()        # same 32-bit patterns
  g_fast, l2_fast, g_dev = (a.cpu() for a in ops.race_key_probe(rbits))
  u = (k.double() / 16777216.0).float()          # Philox::u01
  eps = torch.tensor(1e-10, dtype=torch.float32)
  g_ref = eps - (u + eps).double().log().float()
  assert int((g_ref != g_dev).sum()) == 0, 'defining g differs from RN(fp64 log)'
  l2 = g_dev.double().log2()
  assert -24.1 < l2.min().item() and l2.max().item() < 4.6
  e = 2.0 ** -24
  ratio = (l2_fast.double() - l2).abs() / (e * l2.abs().clamp(min=1.0))
  assert ratio.max().item() <= 5.0, ratio.max().item()      # budget: 5 e max(1, |log2 g|)
  rel = (g_fast.double() - g_dev.double()).abs() / g_dev.double()
  assert rel.max().item() <= 6.0 * e, rel.max().item() / e   # budget: 6 e
  # prefilter premise: g > 2^(-1 - n), n = leading bits of r equal to its top bit
  r = (k << 8)
  n_ones = 32 - torch.log2((((1 << 32) - 1) - r).clamp(min=1).double()).floor().long() - 1
  n_zero = 32 - torch.log2(r.clamp(min=1).double()).floor().long() - 1
  n = torch.where(r >= (1 << 31), n_ones, n_zero).clamp(max=32)
  assert bool((g_dev.double() > torch.exp2(-1.0 - n.double())).all())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('from_logits', [False, True])
def test_update_fast_is_bit_exact_full_vocab(ops, dtype, from_logits):
  """>= 64 full-vocabulary rows x 3 seeds x two unmasking regimes: the fast path,
  the reference kernel and the oracle (fed the kernel's own fp32 p_x0 when the
  entry point starts from logits) pick identical tokens."""
  torch.manual_seed(21)
  B, L = 8, 8
  logits = (torch.randn(B, L, V_OWT) * 3).to(dtype)
  x = torch.full((B, L), MASK_OWT)
  x[1, 2] = 5; x[6, 0] = 50256
  dl = ops.alloc_logits(B, L, V_OWT, dtype); dl.copy_(logits)
  lse = ops.subs_row_lse(dl, x.cuda(), MASK_OWT)
  if from_logits:
    p_dev = ops.subs_logprobs(dl, x.cuda(), MASK_OWT, out_dtype=torch.float32, as_probs=True)
  else:
    p_dev = ops.subs_logprobs(dl, x.cuda(), MASK_OWT, as_probs=True)   # p_x0 in `dtype`
  p_cpu = p_dev.float().cpu()
  dt = (1 - 1e-5) / 1000
  n_exact = torch.zeros(1, dtype=torch.int64, device='cuda')
  for t in (torch.linspace(1.0, 0.05, B), torch.full((B,), 2 * dt), torch.full((B,), 0.5 * dt)):
    for seed in (1, 2, 3):
      u = ops.philox_uniform(B * L, V_OWT, seed, 9).cpu().view(B, L, V_OWT)
      _, ref = orc.ddpm_caching_update(None, x, t, dt, u, MASK_OWT, p_x0=p_cpu.clone())
      for fast in (False, True):
        if from_logits:
          xn = ops.subs_ddpm_update(dl, lse, x.cuda(), t.cuda(), dt, MASK_OWT, seed=seed,
                                    offset=9, fast_math=fast, n_exact=n_exact)
        else:
          xn = ops.ddpm_cache_update(p_dev, x.cuda(), t.cuda(), dt, MASK_OWT, seed=seed,
                                     offset=9, fast_math=fast, n_exact=n_exact)
        assert torch.equal(xn.cpu(), ref), (float(t[0]), seed, fast)


@pytest.mark.parametrize('from_logits', [False, True])
def test_update_constructed_near_ties(ops, from_logits):
  """Rows whose two best columns are built 0, +-1, +-2 fp32 ulps apart in q/g
  (same random stream the kernel draws): the approximate pass cannot rank them,
  the re-evaluation must, and exact ties must go to the first index.  fp32
  inputs, so the construction is exact; t = 2 dt (mask-key prefilter active) and
  t = dt / 2 (q_mask < 0: no prefilter)."""
  R = 96
  dt = (1 - 1e-5) / 1000
  seed, off = 77, 3
  u = ops.philox_uniform(R, V_OWT, seed, off).cpu()
  eps = torch.tensor(1e-10, dtype=torch.float32)
  g = eps - (u + eps).log()
  ok = _exact_log_ok(u)
  ok[:, MASK_OWT] = False
  gs = torch.where(ok, g, torch.full_like(g, float('inf')))
  two = gs.topk(2, dim=-1, largest=False)
  a, b = two.indices[:, 0], two.indices[:, 1]
  ga, gb = two.values[:, 0], two.values[:, 1]
  rows = torch.arange(R)
  x = torch.full((R, 1), MASK_OWT)
  t = torch.where(rows % 2 == 0, torch.tensor(2 * dt), torch.tensor(0.5 * dt)).float()
  kk = (rows // 2) % 5 - 2                      # ulp offsets -2..2
  def bump(v, k):
    for _ in range(2):
      v = torch.where(k > 0, torch.nextafter(v, torch.full_like(v, float('inf'))), v)
      v = torch.where(k < 0, torch.nextafter(v, torch.full_like(v, -float('inf'))), v)
      k = k - k.sign()
    return v
  if from_logits:
    logits = torch.full((R, V_OWT), -25.0) + torch.rand(R, V_OWT)
    logits[rows, a] = 0.0
    logits[rows, b] = bump((gb / ga).log(), kk)
    dl = ops.alloc_logits(R, 1, V_OWT, torch.float32); dl.copy_(logits.view(R, 1, V_OWT))
    lse = ops.subs_row_lse(dl, x.cuda(), MASK_OWT)
    p_cpu = ops.subs_logprobs(dl, x.cuda(), MASK_OWT, as_probs=True).cpu()
    run = lambda fast, nx: ops.subs_ddpm_update(dl, lse, x.cuda(), t.cuda(), dt, MASK_OWT,
                                                seed=seed, offset=off, fast_math=fast, n_exact=nx)
  else:
    p = torch.full((R, V_OWT), 1e-12)
    p[rows, a] = 0.2
    p[rows, b] = bump(0.2 * gb / ga, kk)
    dp = ops.alloc_logits(R, 1, V_OWT, torch.float32); dp.copy_(p.view(R, 1, V_OWT))
    p_cpu = p.view(R, 1, V_OWT)
    run = lambda fast, nx: ops.ddpm_cache_update(dp, x.cuda(), t.cuda(), dt, MASK_OWT,
                                                 seed=seed, offset=off, fast_math=fast, n_exact=nx)
  _, ref = orc.ddpm_caching_update(None, x, t, dt, u.view(R, 1, V_OWT), MASK_OWT,
                                   p_x0=p_cpu.clone())
  # the construction worked: the winner is one of the two columns, both occur,
  # and the two keys are within a few ulps
  assert bool(((ref[:, 0] == a) | (ref[:, 0] == b)).all())
  assert (ref[:, 0] == a).any() and (ref[:, 0] == b).any()
  nx = torch.zeros(1, dtype=torch.int64, device='cuda')
  assert torch.equal(run(False, nx).cpu(), ref)
  assert nx.item() == 0
  assert torch.equal(run(True, nx).cpu(), ref)
  assert nx.item() == R, nx.item()   # every such row took the re-evaluation


@pytest.mark.parametrize('std', [2.0, 8.0])
def test_update_fast_equals_reference_kernel_full_size(ops, std):
  """BASELINE configs[1] size (64 x 1024 rows x 50,258 columns, bf16 logits): the
  fast path and the reference kernel emit identical tokens on the same stream,
  early (dt/t ~ 1e-3), late (dt/t = 0.5) and at the last step (q_mask ~ 0).
  std 2: a near-uniform output layer (the synthetic bench weights); std 8: a peaked
  one (what a trained checkpoint's logits look like), where most 8-column vectors
  cannot reach the mask key for ANY draw and the fast path skips their random
  numbers altogether."""
  torch.manual_seed(3)
  B, L = 64, 1024
  dl = ops.alloc_logits(B, L, V_OWT); dl.normal_(0, std)
  x = torch.full((B, L), MASK_OWT, dtype=torch.int64, device='cuda')
  x[::3, ::5] = 7
  lse = ops.subs_row_lse(dl, x, MASK_OWT)
  dt = (1 - 1e-5) / 1000
  nx = torch.zeros(1, dtype=torch.int64, device='cuda')
  for i, t in enumerate((torch.linspace(1.0, 0.01, B), torch.full((B,), 2 * dt),
                         torch.full((B,), dt + 1e-5))):
    xf = ops.subs_ddpm_update(dl, lse, x, t.cuda(), dt, MASK_OWT, seed=5, offset=i,
                              fast_math=True, n_exact=nx)
    xe = ops.subs_ddpm_update(dl, lse, x, t.cuda(), dt, MASK_OWT, seed=5, offset=i)
    assert torch.equal(xf, xe), i
    if i == 0:
      assert int((xe != x).sum()) > 20          # the step does unmask tokens
  assert (0 if std > 4 else 1) <= nx.item() < 200   # near-ties exist at this size, and stay rare


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
  """(a)+(b) fused: logits -> x_next.  The kernel's p_x0 = expf(logit - lse) is
  within 1e-5 (relative) of the oracle's SUBS -> exp, and given that p_x0 the
  race is the oracle's, token for token (supplied u and in-kernel Philox)."""
  torch.manual_seed(9)
  B, L = 2, 6
  logits = (torch.randn(B, L, V_OWT) * 3).to(dtype)
  x = torch.full((B, L), MASK_OWT)
  x[0, 2] = 42; x[1, 0] = 7
  t = torch.tensor([[0.8], [0.05]])
  dt = (1 - 1e-5) / 1000
  u = torch.rand(B, L, V_OWT)
  logp = orc.subs_parameterization(logits.float().clone(), x, MASK_OWT)
  dl = ops.alloc_logits(B, L, V_OWT, dtype); dl.copy_(logits)
  lse = ops.subs_row_lse(dl, x.cuda(), MASK_OWT)
  p_dev = ops.subs_logprobs(dl, x.cuda(), MASK_OWT, out_dtype=torch.float32,
                            as_probs=True).cpu()
  m = x == MASK_OWT
  rel = (p_dev[m] - logp.exp()[m]).abs() / logp.exp()[m].clamp(min=1e-30)
  assert rel[:, :MASK_OWT].max().item() <= 1e-5
  _, ref = orc.ddpm_caching_update(None, x, t, dt, u, MASK_OWT, p_x0=p_dev.clone())
  xn = ops.subs_ddpm_update(dl, lse, x.cuda(), t.cuda(), dt, MASK_OWT, u=u.cuda())
  assert torch.equal(xn.cpu(), ref)
  assert torch.equal(xn.cpu()[x != MASK_OWT], x[x != MASK_OWT])
  # the oracle's own p_x0 gives the same tokens on these rows (no near-tie among them)
  _, ref2 = orc.ddpm_caching_update(None, x, t, dt, u, MASK_OWT, p_x0=logp.exp())
  assert torch.equal(ref2, ref)
  # philox variants against the oracle with the materialised draw
  up = ops.philox_uniform(B * L, V_OWT, 99, 5)
  _, refp = orc.ddpm_caching_update(None, x, t, dt, up.cpu().view(B, L, V_OWT),
                                    MASK_OWT, p_x0=p_dev.clone())
  for fast in (False, True):
    xp = ops.subs_ddpm_update(dl, lse, x.cuda(), t.cuda(), dt, MASK_OWT, seed=99,
                              offset=5, fast_math=fast)
    assert torch.equal(xp.cpu(), refp), fast


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
