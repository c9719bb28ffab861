"""GPU tests of the `Diffusion` drop-in: sampling loop, cache policies, semi-AR
and NELBO evaluation, against the CPU oracle.

End-to-end bit-exactness through a bf16 backbone is not defined (different
GEMM reduction order), so parity is pinned at the sampler boundary: the oracle
runs the reference loop on the *GPU backbone's logits* and the *same Philox
uniforms*; every token of every step must then match."""
import math

import pytest
import torch

import mdlm_oracle as orc

pytestmark = pytest.mark.gpu

V = 300          # synthetic tokenizer vocab 299 + appended mask token
MASK = V - 1


@pytest.fixture(scope='module')
def ops():
  assert torch.cuda.is_available()
  from mdlm_amd import ops as _ops
  return _ops


def make_model(overrides=(), seed=5, vocab=V):
  from mdlm_amd.config import load_config
  from mdlm_amd.diffusion import Diffusion
  from mdlm_amd.tokenizer import SyntheticTokenizer
  # default config: sampling.fast_math=true (the two-pass race, same token ids)
  cfg = load_config(['model=tiny', 'loader.eval_batch_size=3', 'sampling.steps=24',
                     *overrides])
  model = Diffusion(cfg, SyntheticTokenizer(vocab_size=vocab - 1, eos_token_id=7)).cuda()
  m = cfg.model
  p = orc.synthetic_dit_params(m.hidden_size, m.n_heads, m.n_blocks, m.cond_dim, vocab, seed=seed)
  sd = dict(p); sd['rotary_emb.inv_freq'] = model.backbone.rotary_emb.inv_freq
  model.backbone.load_state_dict(sd)
  model.ema = None
  # these tests stand for the B = 64 loop: keep its 256x256 GEMM tiles at B = 1, 2 (DIT.run would
  # pick 128x128 tiles for so few samples; same bits, checked in the first test below)
  model.backbone.small_tile_limit = 0
  return model, cfg, p


class GpuBackedOracle(orc.OracleDiffusion):
  """Oracle loop whose backbone is the GPU model and whose uniforms are the
  Philox stream the kernels consume (offset advances as in Diffusion._draw)."""

  def __init__(self, model, ops, *a, **k):
    super().__init__(*a, **k)
    self.model, self.ops = model, ops
    self.offset = 0
    self.seed = model._rng_seed

  def backbone(self, x, sigma):
    self.nfe += 1
    return self.model.backbone(x.cuda(), sigma.cuda()).float().cpu()

  def draw(self, B, L, Vv):
    u = self.ops.philox_uniform(B * L, Vv, self.seed, self.offset).cpu().view(B, L, Vv)
    self.offset += B * L * ((Vv + 3) // 4)
    return u


def test_attributes_and_api_surface():
  model, cfg, _ = make_model()
  assert model.mask_index == V - 1 and model.vocab_size == V
  assert model.neg_infinity == -1000000.0 and model.sampler == 'ddpm_cache'
  assert model.T == 0 and model.time_conditioning is False
  assert model.parameterization == 'subs'
  for name in ('forward', '_subs_parameterization', '_process_sigma', 'q_xt',
               '_sample_prior', '_ddpm_caching_update', '_sample',
               'restore_model_and_sample', 'sample_subs_guidance',
               'restore_model_and_semi_ar_sample', '_sample_t',
               '_forward_pass_diffusion', '_loss', '_compute_loss', 'validation_step'):
    assert callable(getattr(model, name)), name
  assert torch.equal(model._sample_prior(2, 5), torch.full((2, 5), V - 1))
  # the backbone state dict has the reference's key layout
  keys = set(model.backbone.state_dict())
  assert {'vocab_embed.embedding', 'sigma_map.mlp.0.weight', 'rotary_emb.inv_freq',
          'blocks.0.attn_qkv.weight', 'blocks.1.adaLN_modulation.bias',
          'output_layer.linear.weight', 'output_layer.norm_final.weight'} <= keys


def test_forward_is_subs_logprobs(ops):
  model, cfg, p = make_model()
  torch.manual_seed(0)
  B, L = 2, cfg.model.length
  x = torch.randint(0, V - 1, (B, L)); x[:, ::2] = MASK
  lp = model.forward(x.cuda(), torch.rand(B, 1).cuda()).float().cpu()  # sigma is zeroed
  ref = orc.subs_parameterization(
    orc.dit_forward(p, x, torch.zeros(B), cfg.model.n_heads, cfg.model.n_blocks), x, MASK)
  m = x == MASK
  assert (lp[m][:, :MASK] - ref[m][:, :MASK]).abs().max().item() < 6e-2
  assert (lp[m][:, MASK] < -9e5).all()
  # unmasked rows: exactly the reference pattern
  assert torch.equal(lp[~m], ref[~m].to(torch.bfloat16).float())


def test_ddpm_caching_update_api(ops):
  """API-level method: (p_x0, x_next), cache reuse, supplied u, carry-over."""
  model, cfg, p = make_model(['sampling.p_x0_dtype=fp32'])
  B, L = 2, cfg.model.length
  torch.manual_seed(1)
  x = torch.full((B, L), MASK); x[0, :10] = torch.randint(0, V - 1, (10,))
  t = torch.full((B, 1), 0.6)
  u = torch.rand(B, L, V)
  p_x0, xn = model._ddpm_caching_update(x.cuda(), t.cuda(), 0.05, u=u.cuda())
  assert p_x0.shape == (B, L, V) and p_x0.dtype == torch.float32
  # oracle on the returned probabilities and the same u: bit-exact tokens
  _, ref = orc.ddpm_caching_update(None, x, t, 0.05, u, MASK, p_x0=p_x0.cpu().clone())
  assert torch.equal(xn.cpu(), ref)
  assert torch.equal(xn.cpu()[x != MASK], x[x != MASK])
  # probabilities are SUBS(backbone).exp()
  s = p_x0.sum(-1).cpu()
  assert (s - 1).abs().max().item() < 1e-3
  # cache hit: the backbone is not called again, the same p_x0 object is returned
  p2, xn2 = model._ddpm_caching_update(x.cuda(), t.cuda(), 0.05, p_x0=p_x0, u=u.cuda())
  assert p2 is p_x0 and torch.equal(xn2, xn)


@pytest.mark.parametrize('policy', ['sample', 'batch'])
def test_sample_trajectory_matches_oracle(ops, policy):
  """Fused sampler == oracle reference loop on GPU logits + Philox uniforms."""
  model, cfg, p = make_model([f'sampling.cache_policy={policy}'])
  B, L, T = 3, cfg.model.length, 24
  model.seed_rng(123)
  out = model._sample(num_steps=T).cpu()
  nfe = int(model.last_nfe.item())
  od = GpuBackedOracle(model, ops, p, cfg.model.n_heads, cfg.model.n_blocks, V, MASK, L)
  od.seed = 123
  od.rand = lambda *shape: od.draw(*shape)
  ref = od._sample(B, T)
  assert torch.equal(out, ref)
  assert (out != MASK).all()
  if policy == 'batch':
    # reference policy: all-or-nothing (the reference always pays the final
    # noise-removal forward; the cache makes it free when nothing changed)
    assert nfe in (od.nfe * B, (od.nfe - 1) * B)
  else:
    assert nfe <= od.nfe * B            # per-sample cache never does more work


def test_cache_policies_and_reference_loop_agree(ops):
  """per-sample cache == all-or-nothing cache == no cache == the literal
  reference loop through the API-level `_ddpm_caching_update`."""
  outs = {}
  for name, ov in {'sample': ['sampling.cache_policy=sample'],
                   'batch': ['sampling.cache_policy=batch'],
                   'none': ['sampling.cache_policy=none'],
                   'allrows': ['sampling.masked_rows_only=false'],
                   'loop': ['sampling.fused=false', 'sampling.p_x0_dtype=fp32']}.items():
    model, cfg, _ = make_model(ov)
    model.seed_rng(77)
    outs[name] = (model._sample(num_steps=24).cpu(), int(model.last_nfe.item()))
  for name in ('batch', 'none', 'loop', 'allrows'):
    assert torch.equal(outs['sample'][0], outs[name][0]), name
  assert outs['allrows'][1] == outs['sample'][1]
  assert outs['sample'][1] <= outs['batch'][1] <= outs['none'][1]
  assert outs['loop'][1] - outs['batch'][1] in (0, 3)  # final forward may be cached
  assert outs['none'][1] == 3 * 25


def test_fast_and_reference_race_emit_the_same_tokens(ops):
  """`sampling.fast_math` only chooses HOW the race is evaluated (approximate
  ranking + exact re-evaluation of near-ties vs every element in the defining
  arithmetic): whole trajectories are identical, token for token, and so is the
  number of backbone evaluations."""
  outs = []
  for fm in ('true', 'false'):
    model, cfg, _ = make_model([f'sampling.fast_math={fm}', 'loader.eval_batch_size=8'])
    model.seed_rng(5)
    outs.append((model._sample(num_steps=32).cpu(), int(model.last_nfe.item())))
  assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
  assert (outs[0][0] != MASK).all()
  assert 0.5 * 32 < outs[0][1] / 8 <= 33


def test_stream_groups_are_result_identical(ops):
  """`sampling.stream_groups` runs the backbone forwards of disjoint sample
  ranges on separate HIP streams (per-group active lists, row lists and
  workspaces, joined before the update kernel): same tokens, same NFE, also in
  the semi-AR driver that shares the cache refresh (same count of steps in which
  a token changed)."""
  outs = []
  for g in (1, 2, 4):
    model, cfg, _ = make_model([f'sampling.stream_groups={g}', 'loader.eval_batch_size=8'])
    model.seed_rng(11)
    x = model._sample(num_steps=40).cpu()
    nfe = int(model.last_nfe.item())
    assert len(model._state['groups']) == g
    model.seed_rng(12)
    n_chg, inter, _ = model.sample_subs_guidance(n_samples=8, stride_length=64, num_strides=1, dt=1 / 16)
    outs.append((x, nfe, inter, n_chg))
  for o in outs[1:]:
    assert torch.equal(outs[0][0], o[0]) and outs[0][1] == o[1]
    assert outs[0][2] == o[2] and outs[0][3] == o[3]
  assert 2 < outs[0][3] <= 2 * 17
  assert (outs[0][0] != MASK).all()
  # the other cache-refresh variants under two groups: every row projected; row lse in the update kernel
  for ov in ('sampling.masked_rows_only=false', 'sampling.lse_from_gemm=false'):
    model, cfg, _ = make_model(['sampling.stream_groups=2', 'loader.eval_batch_size=8', ov])
    model.seed_rng(11)
    x = model._sample(num_steps=40).cpu()
    assert len(model._state['groups']) == 2
    assert torch.equal(outs[0][0], x) and outs[0][1] == int(model.last_nfe.item()), ov
  # a batch that does not split into 256-row multiples falls back to one group
  model, cfg, _ = make_model(['sampling.stream_groups=2', 'loader.eval_batch_size=3'])
  model.seed_rng(11)
  model._sample(num_steps=8)
  assert len(model._state['groups']) == 1


def test_result_identical_knobs_agree_on_small_shapes(ops):
  """Every combination of the result-identical knobs (cache policy, stream groups,
  masked-row projection, row lse from the GEMM epilogue, race evaluation) emits the
  tokens of the plainest configuration -- no cache, one stream, every row
  projected, lse inside the update kernel, reference race -- on shapes where a
  sample is shorter than a GEMM tile and the batch does not split evenly."""
  import itertools
  for B, L, steps in ((8, 128, 48), (5, 128, 30), (6, 256, 40)):
    combos = [dict(cache_policy='none', stream_groups=1, masked_rows_only='false',
                   lse_from_gemm='false', fast_math='false')]
    for cp, g, mr, lg in itertools.product(('sample', 'batch'), (1, 2, 3), ('true', 'false'),
                                           ('true', 'false')):
      combos.append(dict(cache_policy=cp, stream_groups=g, masked_rows_only=mr, lse_from_gemm=lg,
                         fast_math='true'))
    base = None
    for c in combos:
      ov = [f'sampling.{k}={v}' for k, v in c.items()]
      model, cfg, _ = make_model(ov + [f'loader.eval_batch_size={B}', f'model.length={L}'])
      model.seed_rng(21)
      x = model._sample(num_steps=steps).cpu()
      if base is None:
        base = x
      assert torch.equal(base, x), (B, L, c)


def test_state_reuse_across_calls(ops):
  """One model runs sampling, semi-AR generation, the loss, the API-level cached
  update and forward in sequence (they share the sampler / eval state, the stream
  groups and the DiT workspaces); every result equals that of a fresh model given
  the same Philox seed."""
  B, L = 8, 128
  ov = [f'loader.eval_batch_size={B}', f'model.length={L}']
  torch.manual_seed(0)
  x0 = torch.randint(0, V - 1, (B, L)).cuda()
  tt = (torch.rand(B) * 0.98 + 0.01).cuda()
  u = torch.rand(B, L).cuda()
  xt = x0.clone(); xt[torch.rand(B, L).cuda() < 0.5] = MASK

  def calls(model):
    def sample40():
      model.seed_rng(1); return model._sample(num_steps=40).cpu()
    def semi():
      model.seed_rng(2)
      return model.sample_subs_guidance(n_samples=B, stride_length=64, num_strides=1, dt=1 / 16)[1]
    def sample24():
      model.seed_rng(3); return model._sample(num_steps=24).cpu()
    def loss():
      return model._loss(x0, None, t=tt, u=u).nlls.float().cpu()
    def api():
      model.seed_rng(4)
      p, xn = model._ddpm_caching_update(xt, torch.full((B, 1), 0.5).cuda(), 0.01)
      return model._ddpm_caching_update(xn, torch.full((B, 1), 0.49).cuda(), 0.01, p_x0=p)[1].cpu()
    def fwd():
      return model.forward(xt, torch.zeros(B).cuda()).float().cpu()
    return dict(sample40=sample40, semi=semi, sample24=sample24, loss=loss, api=api, fwd=fwd)

  ref = {}
  for name in ('sample40', 'semi', 'sample24', 'loss', 'api', 'fwd'):
    ref[name] = calls(make_model(ov)[0])[name]()
  fns = calls(make_model(ov)[0])
  for name in ('sample40', 'semi', 'loss', 'sample24', 'api', 'fwd', 'semi', 'sample40', 'api', 'loss',
               'sample24', 'fwd', 'sample40'):
    out = fns[name]()
    if not torch.is_tensor(out):
      assert out == ref[name], name
    elif out.dtype == torch.float32:
      assert (out - ref[name]).abs().max() < 1e-5, name
    else:
      assert torch.equal(out, ref[name]), name


def test_time_conditioning_path(ops):
  model, cfg, p = make_model(['time_conditioning=true'])
  model.seed_rng(9)
  out = model._sample(num_steps=8).cpu()
  assert (out != MASK).all()
  assert int(model.last_nfe.item()) == 3 * 9  # no caching with time conditioning
  od = GpuBackedOracle(model, ops, p, cfg.model.n_heads, cfg.model.n_blocks, V, MASK,
                       cfg.model.length, time_conditioning=True)
  od.seed = 9
  od.rand = lambda *shape: od.draw(*shape)
  assert torch.equal(out, od._sample(3, 8))


def test_semi_ar_matches_oracle(ops):
  model, cfg, p = make_model()
  L = cfg.model.length
  model.seed_rng(31)
  steps, texts, seqlens = model.sample_subs_guidance(
    n_samples=2, stride_length=L // 2, num_strides=2, dt=1 / 8)
  toks = torch.from_numpy(model.last_tokens)
  assert toks.shape == (2, L + 2 * (L // 2))
  od = GpuBackedOracle(model, ops, p, cfg.model.n_heads, cfg.model.n_blocks, V, MASK, L)
  od.seed = 31
  od.rand = lambda *shape: od.draw(*shape)
  ref_steps, ref_toks = od.sample_subs_guidance(2, L // 2, 2, dt=1 / 8)
  assert torch.equal(toks, ref_toks)
  assert steps == ref_steps
  assert len(texts) == 3 and len(seqlens) == 2
  # sequence_lengths: tokens before the first EOS (diffusion.py:989-991)
  for i in range(2):
    row = toks[i, 1:].tolist()
    exp = row.index(7) if 7 in row else len(row)
    assert int(seqlens[i]) == exp


def test_semi_ar_with_a_real_local_tokenizer(ops, tmp_path):
  """main.py:94-107 end to end with a REAL tokenizer loaded from a local
  directory (dataloader.py:488-522; a byte-level BPE in the GPT-2 file format,
  trained in the test because the hub is unreachable): mask_index = vocab_size
  (diffusion.py:85-90), semi-AR sampling, `batch_decode` of the sampled ids and the
  EOS-based sequence lengths (diffusion.py:989-995)."""
  from test_host import _local_gpt2_style_tokenizer
  from mdlm_amd.config import load_config
  from mdlm_amd.diffusion import Diffusion
  from mdlm_amd.tokenizer import get_tokenizer
  d = str(tmp_path / 'tok')
  _local_gpt2_style_tokenizer(d)
  cfg = load_config(['model=tiny', 'loader.eval_batch_size=2', 'sampling.steps=8',
                     f'data.tokenizer_name_or_path={d}'])
  tok = get_tokenizer(cfg)
  model = Diffusion(cfg, tok).cuda()
  Vt = tok.vocab_size + 1
  assert model.mask_index == tok.vocab_size and model.vocab_size == Vt
  m = cfg.model
  p = orc.synthetic_dit_params(m.hidden_size, m.n_heads, m.n_blocks, m.cond_dim, Vt, seed=4)
  sd = dict(p); sd['rotary_emb.inv_freq'] = model.backbone.rotary_emb.inv_freq
  model.backbone.load_state_dict(sd)
  model.ema = None
  L = m.length
  steps, texts, seqlens = model.restore_model_and_semi_ar_sample(
    stride_length=L // 2, num_strides=1, dt=1 / 8)
  toks = model.last_tokens
  assert toks.shape == (2, L + L // 2) and (toks != model.mask_index).all()
  assert texts[-1] == tok.batch_decode(toks)           # the decoded text is the sampled ids'
  assert all(isinstance(t, str) and len(t) > 0 for t in texts[-1])
  eos = tok.eos_token_id
  for i in range(2):
    row = toks[i, 1:].tolist()
    assert int(seqlens[i]) == (row.index(eos) if eos in row else len(row))


def test_loss_matches_oracle(ops):
  model, cfg, p = make_model()
  B, L = 4, cfg.model.length
  torch.manual_seed(3)
  x0 = torch.randint(0, V - 1, (B, L))
  attn = torch.ones(B, L); attn[1, -5:] = 0
  t = orc.sample_t(B, torch.rand(B))
  u = torch.rand(B, L)
  loss = model._loss(x0.cuda(), attn.cuda(), t=t.cuda(), u=u.cuda())
  od = orc.OracleDiffusion(p, cfg.model.n_heads, cfg.model.n_blocks, V, MASK, L)
  tok_ref, nll_ref, _ = od._loss(x0, attn, t=t, u_xt=u)
  err = (loss.nlls.cpu() - nll_ref).abs() / nll_ref.abs().clamp(min=1.0)
  assert err.max().item() < 5e-2
  assert abs(loss.loss.item() / tok_ref.item() - 1) < 5e-3       # ppl within 0.5 %
  assert torch.equal(loss.token_mask.cpu(), attn)
  # metrics accumulate in fp64 on the device
  model.valid_metrics.reset()
  model._compute_loss({'input_ids': x0.cuda(), 'attention_mask': attn.cuda()}, 'val')
  res = model.valid_metrics.compute(sync_dist=False)
  assert set(res) == {'val/nll', 'val/bpd', 'val/ppl'}
  assert abs(res['val/bpd'] - res['val/nll'] / math.log(2)) < 1e-9
  assert abs(res['val/ppl'] - math.exp(res['val/nll'])) < 1e-6 * res['val/ppl']


@pytest.mark.parametrize('sampler', ['ddpm', 'analytic'])
@pytest.mark.parametrize('tcond', [False, True])
def test_other_samplers_match_oracle(ops, sampler, tcond):
  """`sampling.predictor` in {ddpm, analytic} (diffusion.py:612-637, 721-798):
  the generic race kernel == the oracle's literal restatement (score ->
  staggered score -> transition -> _sample_categorical) on the GPU backbone's
  logits and the same Philox uniforms, token for token."""
  model, cfg, p = make_model([f'sampling.predictor={sampler}',
                              f'time_conditioning={str(tcond).lower()}'])
  B, L, T = 3, cfg.model.length, 12
  model.seed_rng(77)
  out = model._sample(num_steps=T).cpu()
  assert int(model.last_nfe.item()) == B * (T + 1)
  od = GpuBackedOracle(model, ops, p, cfg.model.n_heads, cfg.model.n_blocks, V, MASK, L,
                       time_conditioning=tcond, sampler=sampler)
  od.seed = 77
  od.rand = lambda *shape: od.draw(*shape)
  ref = od._sample(B, T)
  assert torch.equal(out, ref)
  assert (out != MASK).all()


def test_loss_variants_agree_on_small_shapes(ops):
  """NELBO path: masked-rows projection and row lse from the GEMM epilogue against
  the plain path (every row projected, lse inside subs_nll), supplied t and u, on
  shapes with samples shorter than a tile and row counts that are not multiples
  of 256 (those fall back to the plain path); a second call reuses the eval state."""
  import itertools
  for B, L in ((4, 128), (3, 128), (5, 256)):
    torch.manual_seed(B * 1000 + L)
    x0 = torch.randint(0, V - 1, (B, L)).cuda()
    tt = (torch.rand(B) * 0.98 + 0.01).cuda()
    u = torch.rand(B, L).cuda()
    am = torch.ones(B, L).cuda(); am[:, L - 7:] = 0
    base = None
    for mr, lg in itertools.product(('false', 'true'), ('false', 'true')):
      model, cfg, _ = make_model([f'sampling.masked_rows_only={mr}', f'sampling.lse_from_gemm={lg}',
                                  f'loader.eval_batch_size={B}', f'model.length={L}'])
      for _ in range(2):
        out = model._loss(x0, am, t=tt, u=u)
      n = out.nlls.float().cpu()
      assert not torch.isnan(n).any()
      if base is None:
        base = n
      assert (n - base).abs().max().item() < 2e-5, (B, L, mr, lg)   # lse summation order only


def test_other_sampler_updates_with_supplied_noise(ops):
  """API-level `_ddpm_update` / `_analytic_update` / `_denoiser_update` with a
  caller-supplied uniform tensor (exact kernel path, no Philox)."""
  model, cfg, p = make_model(['time_conditioning=true'])
  B, L = 2, cfg.model.length
  od = GpuBackedOracle(model, ops, p, cfg.model.n_heads, cfg.model.n_blocks, V, MASK, L,
                       time_conditioning=True)
  torch.manual_seed(4)
  x = torch.randint(0, V - 1, (B, L)); x[torch.rand(B, L) < 0.7] = MASK
  t = torch.tensor([[0.7], [0.2]])
  dt = 0.05
  for name in ('ddpm', 'analytic', 'denoiser'):
    u = torch.rand(B, L, V)
    if name == 'ddpm':
      got = model._ddpm_update(x.cuda(), t.cuda(), dt, u=u.cuda())
      ref = orc.ddpm_update(od.forward, x, t, dt, u, MASK)
    elif name == 'analytic':
      got = model._analytic_update(x.cuda(), t.cuda(), dt, u=u.cuda())
      ref = orc.analytic_update(od.forward, x, t, dt, u, MASK, V)
    else:
      got = model._denoiser_update(x.cuda(), t.cuda(), u=u.cuda())
      ref = orc.denoiser_update(od.forward, x, t, u, MASK, V)
      assert (got != MASK).all()
    assert torch.equal(got.cpu(), ref), name
    assert torch.equal(got.cpu()[x != MASK], x[x != MASK])


def test_d3pm_loss_matches_oracle(ops):
  """T > 0 (diffusion.py:329-358, 849-853): discrete-time SUBS loss, including
  the NaN rows the reference produces at t == 1/T."""
  Tn = 8
  model, cfg, p = make_model([f'T={Tn}'])
  B, L = 8, cfg.model.length
  torch.manual_seed(6)
  x0 = torch.randint(0, V - 1, (B, L))
  attn = torch.ones(B, L); attn[2, -4:] = 0
  t = orc.sample_t(B, torch.rand(B))
  u = torch.rand(B, L)
  loss = model._loss(x0.cuda(), attn.cuda(), t=t.cuda(), u=u.cuda())
  od = orc.OracleDiffusion(p, cfg.model.n_heads, cfg.model.n_blocks, V, MASK, L, T=Tn)
  tok_ref, nll_ref, _ = od._loss(x0, attn, t=t, u_xt=u)
  got = loss.nlls.cpu()
  assert torch.equal(got.isnan(), nll_ref.isnan())
  assert nll_ref.isnan().any() and not nll_ref.isnan().all()   # both kinds of rows are covered
  ok = ~nll_ref.isnan()
  err = (got[ok] - nll_ref[ok]).abs() / nll_ref[ok].abs().clamp(min=1.0)
  assert err.max().item() < 5e-2
  assert math.isnan(loss.loss.item()) == math.isnan(tok_ref.item())
  # the kernel alone, fp32 inputs from the oracle: tight tolerance
  lp = torch.randn(B, L).abs()
  xt = torch.where(torch.rand(B, L) < 0.5, torch.full((B, L), MASK), x0)
  tt = orc.discretize_t(t, Tn)
  mo = torch.full((B, L, V), -1e6); mo.scatter_(-1, x0[:, :, None], -lp[:, :, None])
  ref = orc.d3pm_loss(mo, xt, x0, tt, Tn, MASK)
  out = ops.d3pm_loss(lp.cuda(), xt.cuda(), tt.cuda(), Tn, MASK).cpu()
  torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6, equal_nan=True)


def test_out_of_scope_is_loud():
  from mdlm_amd.config import load_config
  from mdlm_amd.diffusion import Diffusion
  from mdlm_amd.tokenizer import SyntheticTokenizer
  tok = SyntheticTokenizer(vocab_size=V - 1)
  with pytest.raises(NotImplementedError):
    Diffusion(load_config(['model=tiny', 'parameterization=sedd']), tok)
  with pytest.raises(NotImplementedError):
    Diffusion(load_config(['model=tiny', 'training.importance_sampling=true']), tok)
  with pytest.raises(ValueError):
    Diffusion(load_config(['model=tiny', 'backbone=dimamba']), tok)
  model, cfg, _ = make_model()
  model.cpu()
  with pytest.raises(RuntimeError):
    model.forward(torch.zeros(1, cfg.model.length, dtype=torch.int64), torch.zeros(1))


def test_full_size_rows_properties(ops):
  """BASELINE-size vocabulary (V = 50,258, L = 1024) through the fused sampler
  kernels: size-independent properties only (the oracle would take minutes)."""
  B, L, Vf, mask = 2, 1024, 50258, 50257
  torch.manual_seed(4)
  logits = ops.alloc_logits(B, L, Vf)
  logits.copy_((torch.randn(B, L, Vf, device='cuda') * 3).bfloat16())
  x = torch.full((B, L), mask, device='cuda')
  x[0, :100] = torch.randint(0, 50257, (100,), device='cuda')
  lse = ops.subs_row_lse(logits, x, mask)
  # lse agrees with torch on the device for masked rows
  sh = logits.float().clone(); sh[:, :, mask] -= 1e6
  ref = torch.logsumexp(sh, -1).view(-1)
  m = (x == mask).view(-1)
  assert (lse[m] - ref[m]).abs().max().item() < 1e-3
  t = torch.full((B,), 0.5, device='cuda')
  changed = torch.zeros(B, dtype=torch.int32, device='cuda')
  xn = ops.subs_ddpm_update(logits, lse, x, t, 0.25, mask, seed=1, offset=0, changed=changed)
  assert torch.equal(xn[0, :100], x[0, :100])                   # carry-over
  frac = (xn[x == mask] != mask).float().mean().item()
  assert abs(frac - 0.5) < 0.05                                   # P(unmask) = dt / t
  assert (changed == 1).all()
  # idempotence of noise removal: argmax of an unmasked sequence is itself
  am = ops.subs_argmax(logits, xn, mask)
  assert torch.equal(am[xn != mask], xn[xn != mask])
  assert (am != mask).all()
  am2 = ops.subs_argmax(logits, am, mask)
  assert torch.equal(am2, am)


def test_lazy_unmask_kernels_statistics(ops):
  """Opt-in two-stage step (no reference counterpart; distribution-identical to
  diffusion.py:592-610): stage 1 unmasks with probability dt/t, never touches
  unmasked positions and flags the samples it changed; stage 2 draws tokens
  from softmax(logits) over v != mask."""
  B, L, Vv, mask = 64, 1024, 24, 23
  torch.manual_seed(0)
  x = torch.randint(0, Vv - 1, (B, L)); x[torch.rand(B, L) < 0.8] = mask
  x[5] = 3                                   # a sample with nothing masked
  t = torch.full((B,), 0.5); dt = 0.1       # P(unmask) = 0.2
  xc = x.cuda()
  x_next = torch.empty_like(xc)
  rows = torch.zeros(B * L + 256, dtype=torch.int32, device='cuda')
  n = torch.zeros(2, dtype=torch.int32, device='cuda')
  changed = torch.zeros(B, dtype=torch.int32, device='cuda')
  ops.lazy_select(xc, t.cuda(), dt, mask, x_next, rows, n, changed, seed=11, offset=5)
  n_pad, n_sel = n.tolist()
  assert n_pad % 256 == 0 and 0 <= n_pad - n_sel < 256
  assert torch.equal(x_next, xc)
  sel = rows[:n_sel].long().cpu()
  assert (sel[1:] > sel[:-1]).all()                      # ascending, unique
  assert (x.view(-1)[sel] == mask).all()                 # only masked positions
  n_masked = int((x == mask).sum())
  p = 0.2
  assert abs(n_sel - p * n_masked) < 5 * math.sqrt(n_masked * p * (1 - p))
  exp_changed = torch.zeros(B, dtype=torch.int32); exp_changed[torch.unique(sel // L)] = 1
  assert torch.equal(changed.cpu(), exp_changed) and changed[5].item() == 0
  # t <= dt: everything unmasks
  ops.lazy_select(xc, torch.full((B,), 0.1).cuda(), 0.1, mask, x_next, rows, n, changed, seed=1)
  assert n.tolist()[1] == n_masked
  # stage 2: 32768 rows sharing one logits row -> empirical token distribution
  R = 32768
  lg = torch.randn(Vv) * 1.5
  logits = lg.repeat(R, 1).to(torch.bfloat16).cuda()
  rows2 = torch.arange(R, dtype=torch.int32, device='cuda')
  n2 = torch.tensor([R, R], dtype=torch.int32, device='cuda')
  out = torch.full((R,), -1, dtype=torch.int64, device='cuda')
  ops.lazy_race(logits, rows2, n2, out, Vv, mask, seed=3, offset=0)
  cnt = torch.bincount(out.cpu(), minlength=Vv).double()
  assert cnt[mask] == 0 and cnt.sum() == R
  pr = torch.softmax(logits[0, :mask].float().cpu().double(), -1)
  chi2 = (((cnt[:mask] - R * pr) ** 2) / (R * pr)).sum().item()
  assert chi2 < 60.0        # 22 dof: P(chi2 > 60) ~ 2e-5


def test_lazy_unmask_sampler(ops):
  """End to end: no mask left, hidden-state cache NFE matches the exact
  sampler's on average, unmask times follow the same law (mean step ~ T/2)."""
  model, cfg, p = make_model(['sampling.lazy_unmask=true', 'loader.eval_batch_size=4',
                              'sampling.fast_math=true'])
  model.seed_rng(5)
  out = model._sample(num_steps=32).cpu()
  assert out.shape == (4, cfg.model.length) and (out != MASK).all()
  nfe_lazy = int(model.last_nfe.item())
  exact, _, _ = make_model(['loader.eval_batch_size=4', 'sampling.fast_math=true'])
  exact.seed_rng(5)
  out2 = exact._sample(num_steps=32).cpu()
  nfe_exact = int(exact.last_nfe.item())
  assert abs(nfe_lazy - nfe_exact) <= 0.25 * nfe_exact
  # token marginals of the two samplers agree in distribution (same model):
  # compare the mean token id as a crude two-sample statistic
  a, b = out.float().mean().item(), out2.float().mean().item()
  assert abs(a - b) < 0.25 * V


def test_lazy_step_matches_reference_distribution(ops):
  """One `_lazy_step` from a fixed state, repeated: the unmask rate is dt/t and
  the drawn tokens follow p_x0 OF THEIR OWN POSITION (checks the row gather /
  scatter of the hidden-state cache): the mean of log p_x0[r, x_next[r]] over
  unmasked positions must equal its expectation sum_v p log p."""
  model, cfg, p = make_model(['sampling.lazy_unmask=true', 'loader.eval_batch_size=4',
                              'sampling.fast_math=true'])
  B, L = 4, cfg.model.length
  torch.manual_seed(8)
  x = torch.randint(0, V - 1, (B, L)); x[torch.rand(B, L) < 0.6] = MASK
  xc = x.cuda()
  logp = model.forward(xc, torch.zeros(B, device='cuda')).float().cpu()   # SUBS log-probs
  masked = (x == MASK)
  pm = logp[masked][:, :MASK].double()                 # (n_masked, V-1)
  ent = (pm.exp() * pm).sum(-1)                        # E[log p(token)] per position
  var = (pm.exp() * pm * pm).sum(-1) - ent ** 2
  st = model._sampler_state(B, L)
  t = torch.full((B,), 0.5, device='cuda'); dt = 0.25   # P(unmask) = 0.5
  model.seed_rng(99)
  n_un, s_lp, e_lp, v_lp = 0, 0.0, 0.0, 0.0
  reps = 64
  for _ in range(reps):
    st['changed'].fill_(1)                             # every sample refreshes its hidden state
    xn = torch.empty_like(xc)
    model._lazy_step(st, xc, xn, t, dt)
    xn = xn.cpu()
    assert torch.equal(xn[~masked], x[~masked])        # carry-over
    un = masked & (xn != MASK)
    n_un += int(un.sum())
    s_lp += logp[un].gather(-1, xn[un][:, None]).double().sum().item()
    sel = un[masked]
    e_lp += ent[sel].sum().item(); v_lp += var[sel].sum().item()
  # row mapping, deterministically: dense logits row i of the last step == the
  # full forward's logits at position sel[i] (row-independent kernels: bit-exact)
  full = model.backbone(xc, torch.zeros(B, device='cuda')).view(B * L, -1)
  n_pad, n_sel = st['n_sel'].tolist()
  assert n_sel > 0
  sel = st['sel'][:n_sel].long()
  assert torch.equal(st['cache2d'][:n_sel], full[sel])
  n_m = int(masked.sum()) * reps
  assert abs(n_un - 0.5 * n_m) < 5 * math.sqrt(n_m * 0.25)
  assert abs(s_lp - e_lp) < 5 * math.sqrt(v_lp) + 1e-6 * abs(e_lp) + 0.05 * n_un * 2e-2


def test_checkpoint_ingestion(ops, tmp_path):
  """Reference Lightning `.ckpt` layout (state_dict['backbone.*'] + ckpt['ema']
  shadow params in backbone.parameters() order, diffusion.py:168-181,
  ema.py:79-97) and a plain safetensors backbone file, through the loaders
  `main.py` uses (weights_only=True / safetensors: nothing from the file is
  executed).  Sampling must then run on the EMA weights, as
  `restore_model_and_sample` does in the reference (700-719)."""
  import main as cli
  from safetensors.torch import save_file
  model, cfg, p = make_model(['training.ema=0.9999'])
  src, _, _ = make_model(seed=9)
  sd = {f'backbone.{k}': v.detach().cpu().clone() for k, v in src.backbone.state_dict().items()}
  shadow = [q.detach().cpu().clone() * 0.5 for q in src.backbone.parameters()]
  ckpt = {'state_dict': sd, 'ema': {'decay': 0.9999, 'num_updates': 10, 'shadow_params': shadow}}
  path = str(tmp_path / 'ref.ckpt')
  torch.save(ckpt, path)
  from mdlm_amd.config import load_config
  from mdlm_amd.diffusion import Diffusion
  from mdlm_amd.tokenizer import SyntheticTokenizer
  m2 = Diffusion(load_config(['model=tiny', 'loader.eval_batch_size=4', 'training.ema=0.9999']),
                 SyntheticTokenizer(vocab_size=V - 1, eos_token_id=7)).cuda()
  cli._load_backbone_weights(m2, path)
  for (k, a), b in zip(m2.backbone.state_dict().items(), src.backbone.state_dict().values()):
    assert torch.equal(a.cpu(), b.cpu()), k
  assert m2.ema is not None
  for a, b in zip(m2.ema.shadow_params, shadow):
    assert torch.equal(a.cpu(), b)
  # forward uses the raw weights; sampling swaps the EMA weights in and back out
  x = torch.randint(0, V - 1, (4, cfg.model.length)).cuda()
  before = m2.backbone(x, torch.zeros(4).cuda()).clone()
  out = m2.restore_model_and_sample(num_steps=8)
  assert out.shape == (4, cfg.model.length) and (out != MASK).all()
  assert torch.equal(m2.backbone(x, torch.zeros(4).cuda()), before)
  for a, b in zip(m2.backbone.parameters(), src.backbone.parameters()):
    assert torch.equal(a, b)
  # safetensors with bare backbone keys
  st_path = str(tmp_path / 'bb.safetensors')
  save_file({k: v.detach().cpu().contiguous() for k, v in src.backbone.state_dict().items()}, st_path)
  m3 = Diffusion(load_config(['model=tiny', 'loader.eval_batch_size=4']),
                 SyntheticTokenizer(vocab_size=V - 1, eos_token_id=7)).cuda()
  cli._load_backbone_weights(m3, st_path)
  assert torch.equal(m3.backbone(x, torch.zeros(4).cuda()), before)
  # no `ema` entry in the file: sampling must run on the LOADED weights, not on
  # the constructor-time shadow (random init with a zero output layer)
  assert m3.ema is not None and not m3._ema_restored
  m3.seed_rng(123)
  with_ema = m3.restore_model_and_sample(num_steps=8)
  m3.ema = None
  m3.seed_rng(123)
  assert torch.equal(with_ema, m3.restore_model_and_sample(num_steps=8))


def _full_size_model(B, overrides=()):
  """DiT-small (12 blocks, D = 768, 12 heads), L = 1024, V = 50,258 with the
  oracle's seeded synthetic weights: BASELINE configs[1] at its real size."""
  from mdlm_amd.config import load_config
  from mdlm_amd.diffusion import Diffusion
  from mdlm_amd.tokenizer import SyntheticTokenizer
  Vf = 50258
  cfg = load_config(['model=small', f'loader.eval_batch_size={B}', 'sampling.steps=1000', *overrides])
  model = Diffusion(cfg, SyntheticTokenizer(vocab_size=Vf - 1)).cuda()
  m = cfg.model
  p = orc.synthetic_dit_params(m.hidden_size, m.n_heads, m.n_blocks, m.cond_dim, Vf, seed=1)
  sd = dict(p); sd['rotary_emb.inv_freq'] = model.backbone.rotary_emb.inv_freq
  model.backbone.load_state_dict(sd)
  model.ema = None
  # these tests stand for the B = 64 loop: keep its 256x256 GEMM tiles at B = 1, 2 (DIT.run would
  # pick 128x128 tiles for so few samples; same bits, checked in the first test below)
  model.backbone.small_tile_limit = 0
  return model, cfg, p


def test_full_size_configs1_backbone_and_fused_step(ops):
  """BASELINE configs[1] at its real size, B = 1: DiT-small (12 blocks, D = 768,
  12 heads), L = 1024, V = 50,258 (reference models/dit.py:359-370,
  diffusion.py:592-610).  (1) 12-block logits against the fp32 oracle; (2) one
  fused sampler step from a half-masked x: the masked-row gather, the tail on
  the gathered rows, the 197th (82-column) vocabulary tile and the row scatter
  into the cache give bit-identical logits to the plain forward; (3) the step's
  tokens equal the oracle race on those logits and the same Philox uniforms."""
  Vf, Mf = 50258, 50257
  model, cfg, p = _full_size_model(1)
  m = cfg.model
  L = m.length
  torch.manual_seed(4)
  x = torch.randint(0, Vf - 1, (1, L))
  x[torch.rand(1, L) < 0.5] = Mf
  msk = (x == Mf)[0]
  assert 400 < int(msk.sum()) < 620
  # (1) backbone vs oracle
  xc = x.cuda()
  logits = model.backbone(xc, torch.zeros(1, device='cuda')).clone()    # (1, L, V) bf16
  model.backbone.small_tile_limit = 128      # the default: 128x128 tiles at B = 1, same bits
  assert torch.equal(model.backbone(xc, torch.zeros(1, device='cuda')), logits)
  model.backbone.small_tile_limit = 0
  with torch.no_grad():
    ref = orc.dit_forward(p, x, torch.zeros(1), m.n_heads, m.n_blocks)   # fp32
  got = logits.float().cpu()
  rel = ((got - ref).norm() / ref.norm()).item()
  assert rel < 2e-2, rel
  # log-probs of the masked rows: bf16 logits of magnitude <= 16 carry <= 0.03 of
  # rounding each, 12 blocks of bf16 GEMMs add to it: bound 0.25 absolute, 2e-2 rms
  lp_got = torch.log_softmax(got[0, msk][:, :Mf], -1)
  lp_ref = torch.log_softmax(ref[0, msk][:, :Mf], -1)
  d = (lp_got - lp_ref).abs()
  assert d.max().item() < 0.25 and d.pow(2).mean().sqrt().item() < 2e-2, (d.max().item(), d.pow(2).mean().sqrt().item())
  # (2) one fused step (per-sample cache, masked rows only, tail on gathered rows)
  st = model._sampler_state(1, L)
  xa, xb = st['xa'], st['xb']
  xa.copy_(xc); st['changed'].fill_(1); st['nfe'].zero_()
  model.seed_rng(99)
  t = torch.full((1,), 0.05, device='cuda')
  dt = (1 - 1e-5) / 1000
  model._fused_step(st, xa, xb, t, dt)
  cache = st['cache']
  assert torch.equal(cache[0][msk.cuda()], logits[0][msk.cuda()])
  assert int(st['nfe'].item()) == 1
  # (3) tokens vs the oracle race on the kernel's own fp32 p_x0 and the same uniforms
  p_dev = ops.subs_logprobs(cache, xa, Mf, out_dtype=torch.float32, as_probs=True).cpu()
  u = ops.philox_uniform(L, Vf, 99, 0).cpu().view(1, L, Vf)
  _, xr = orc.ddpm_caching_update(None, x, t.cpu(), dt, u, Mf, p_x0=p_dev)
  assert torch.equal(xb.cpu(), xr)
  n_un = int((xr != x).sum())
  assert 1 <= n_un <= 40 and int(st['changed'].item()) == 1
  assert torch.equal(xr[x != Mf], x[x != Mf])


def test_full_size_trajectory_modes_agree(ops):
  """BASELINE configs[1] shapes (DiT-small, L = 1024, V = 50,258) over a whole
  short trajectory: the default path (fast race, per-sample cache, masked-row
  projection, row lse from the GEMM epilogue) emits the same token ids as the
  reference race kernel (every element in the defining arithmetic) and as the
  reference's all-or-nothing cache policy (diffusion.py:682-685) on the same
  Philox stream."""
  import bench
  from mdlm_amd.config import load_config
  B, T = 2, 48
  outs = []
  for ov in ([], ['sampling.fast_math=false'], ['sampling.cache_policy=batch'],
             ['sampling.stream_groups=2']):
    cfg = load_config(['model=small', f'loader.eval_batch_size={B}', f'sampling.steps={T}', *ov])
    model = bench.build_model(cfg, seed=1)
    model.seed_rng(77)
    outs.append(model.restore_model_and_sample(num_steps=T).cpu())
    assert (outs[-1] != model.mask_index).all()
    del model
    torch.cuda.empty_cache()
  assert torch.equal(outs[0], outs[1])
  assert torch.equal(outs[0], outs[2])
  assert torch.equal(outs[0], outs[3])


def test_full_size_default_loop_matches_oracle_loop(ops):
  """The DEFAULT sampling loop at BASELINE configs[1]'s real size against the
  ORACLE loop (not against another HIP mode): DiT-small, L = 1024, V = 50,258,
  B = 2, several denoising steps late in the schedule from a mostly unmasked
  start, so that in some steps a sample's tokens do not change and its cached
  logits are reused while the other sample is re-evaluated.  Everything the
  default configuration combines is live: fast race, per-sample cache, masked-row
  projection with row scatter into the cache, row lse from the vocabulary GEMM's
  epilogue, two stream groups (one sample each).  The oracle runs the reference
  loop literally (diffusion.py:674-686, 592-610: all-or-nothing cache,
  `_sample_categorical` on fp32 SUBS probabilities) on the GPU backbone's
  logits and the same Philox uniforms; tokens must be equal after every step."""
  Vf, Mf = 50258, 50257
  B, n_steps = 2, 7
  model, cfg, p = _full_size_model(B)
  assert cfg.sampling.get('stream_groups', 1) == 2 and cfg.sampling.get('fast_math', False)
  assert cfg.sampling.get('cache_policy', 'sample') == 'sample'
  m = cfg.model
  L = m.length
  g = torch.Generator().manual_seed(11)
  x0 = torch.randint(0, Vf - 1, (B, L), generator=g)
  # 14 / 9 masked positions: P(no token of a sample changes in a step) = (1 - dt/t)^n
  for b, n in enumerate((14, 9)):
    x0[b, torch.randperm(L, generator=g)[:n]] = Mf
  T = 1000
  dt = (1 - 1e-5) / T
  timesteps = torch.linspace(1, 1e-5, T + 1)
  first = T - 20                       # t = 0.02: dt / t = 0.05 per masked position and step
  st = model._sampler_state(B, L)
  assert len(model._stream_groups(st, False)) == 2
  x, x_next = st['xa'], st['xb']
  x.copy_(x0.cuda()); st['changed'].fill_(1); st['nfe'].zero_()
  model.seed_rng(321)
  od = GpuBackedOracle(model, ops, p, m.n_heads, m.n_blocks, Vf, Mf, L)
  od.seed = 321
  od.rand = lambda *shape: od.draw(*shape)
  xr, cache = x0.clone(), None
  hits = changes = 0
  for i in range(first, first + n_steps):
    t = timesteps[i] * torch.ones(B, 1)
    model._fused_step(st, x, x_next, t.squeeze(-1).cuda(), dt)
    x, x_next = x_next, x
    hits += cache is not None
    cache, xn = od._ddpm_caching_update(xr, t, dt, p_x0=cache)
    if not torch.allclose(xn, xr):                      # diffusion.py:682-685
      cache = None
      changes += 1
    xr = xn
    assert torch.equal(x.cpu(), xr), f'step {i}'
  # the run exercised both branches of the cache, and the per-sample policy did
  # strictly less backbone work than the oracle's all-or-nothing policy
  assert changes >= 2 and int((xr == Mf).sum()) < 23
  nfe = int(st['nfe'].item())
  assert B <= nfe < od.nfe * B or hits > 0, (nfe, od.nfe, hits)
  assert nfe <= od.nfe * B


def test_full_size_loss_matches_oracle(ops):
  """`_loss` (diffusion.py:896-916, 847-894) at the real size, B = 1: q_xt ->
  backbone with the vocabulary projection on the masked rows only -> row lse out
  of the GEMM epilogue -> `mdlm_subs_nll(row_lse=...)`, against the fp32 CPU
  oracle's `_loss` on the same x0, t and q_xt uniforms.  Tolerances: per-token
  NLL within 2e-2 relative (bf16 logits), token NLL (hence perplexity, its exp)
  within 0.5 %."""
  Vf, Mf = 50258, 50257
  model, cfg, p = _full_size_model(1)
  m = cfg.model
  L = m.length
  assert cfg.sampling.get('lse_from_gemm', True) and cfg.sampling.get('masked_rows_only', True)
  g = torch.Generator().manual_seed(5)
  x0 = torch.randint(0, Vf - 1, (1, L), generator=g)
  x0[0, 0] = x0[0, -1] = Vf - 2                       # BOS / EOS wrap (dataloader.py:286-296)
  attn = torch.ones(1, L); attn[0, -7:] = 0
  t = torch.tensor([0.37])
  u = torch.rand(1, L, generator=g)
  loss = model._loss(x0.cuda(), attn.cuda(), t=t.cuda(), u=u.cuda())
  assert model._eval_state is not None and int(model._eval_state['n_mrows'].item()) > 256
  od = orc.OracleDiffusion(p, m.n_heads, m.n_blocks, Vf, Mf, L)
  with torch.no_grad():
    tok_ref, nll_ref, _ = od._loss(x0, attn, t=t, u_xt=u)
  got = loss.nlls.cpu()
  live = nll_ref != 0
  assert torch.equal(got != 0, live)                   # unmasked / padded positions: exactly 0
  assert 300 < int(live.sum()) < 450
  rel = ((got - nll_ref).abs() / nll_ref.abs().clamp(min=1.0))[live]
  assert rel.max().item() < 2e-2, rel.max().item()
  # perplexity = exp(token NLL) (diffusion.py:58-65): within 0.5 %
  assert abs(math.exp(loss.loss.item() - tok_ref.item()) - 1) < 5e-3, (loss.loss.item(), tok_ref.item())
