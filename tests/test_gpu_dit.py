"""GPU parity tests for the DiT-side kernels (d): LN+modulate, bf16 MFMA GEMM
with its epilogues, rotary, attention, and the whole backbone, against the
CPU oracle (fp32).  bf16 tolerance per north_star: 2e-2."""
import math

import pytest
import torch
import torch.nn.functional as F

import mdlm_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', params=['tile256', 'tile128'])
def ops(request):
  """Every test runs with both GEMM tile configurations."""
  assert torch.cuda.is_available()
  from mdlm_amd import ops as _ops
  _ops.gemm_set_tile(request.param == 'tile128')
  yield _ops
  _ops.gemm_set_tile(False)


def _bf(t):
  return t.to(torch.bfloat16)


def _rel(a, b):
  return ((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-12)).item()


def test_small_ops(ops):
  torch.manual_seed(0)
  # timestep embedding (dit.py:162-184)
  t = torch.tensor([0.0, 0.3, 2.0, 7.5])
  te = ops.timestep_embedding(t.cuda(), 256).cpu()
  ref = orc.timestep_embedding(t, 256)
  assert (te - ref).abs().max().item() < 1e-5
  assert torch.equal(te[0], torch.cat([torch.ones(128), torch.zeros(128)]))
  # conditioning linears
  x = torch.randn(3, 40); W = torch.randn(17, 40); b = torch.randn(17)
  y = ops.linear_f32(x.cuda(), W.cuda(), b.cuda(), act_in=1, act_out=1).cpu()
  ref = F.silu(F.linear(F.silu(x), W, b))
  assert (y - ref).abs().max().item() < 1e-4
  # embedding gather with a sample map and a device-side live-row count
  table = torch.randn(50, 64)
  idx = torch.randint(0, 50, (4, 8))
  out = torch.zeros(32, 64).cuda()
  amap = torch.tensor([2, 0, 3, 1], dtype=torch.int32).cuda()
  nrows = torch.tensor([16], dtype=torch.int32).cuda()
  ops.embed_rows(table.cuda(), idx.cuda(), out, sample_map=amap, L=8, n_rows_dev=nrows)
  ref = table[idx[[2, 0]].reshape(-1)]
  assert torch.equal(out[:16].cpu(), ref)
  assert (out[16:] == 0).all()


@pytest.mark.parametrize('D', [128, 768, 1024])
def test_ln_modulate(ops, D):
  torch.manual_seed(1)
  B, L = 3, 64
  x = torch.randn(B * L, D) * 2 + 0.5
  w = 1 + 0.1 * torch.randn(D)
  mod = torch.randn(B, 6 * D) * 0.5
  out = torch.empty(B * L, D, dtype=torch.bfloat16, device='cuda')
  dm = mod.cuda()
  ops.ln_modulate(x.cuda(), w.cuda(), dm[:, D:2 * D], dm[:, 3 * D:4 * D], out, L,
                  n_mod=B, mod_ld=6 * D)
  ln = orc.layer_norm_w(x.view(B, L, D), w)
  ref = orc.modulate(ln, mod[:, None, D:2 * D], mod[:, None, 3 * D:4 * D]).view(B * L, D)
  err = (out.float().cpu() - ref).abs()
  assert (err <= 2e-2 * ref.abs().clamp(min=1.0)).all()
  # batch-invariant conditioning (n_mod = 1)
  ops.ln_modulate(x.cuda(), w.cuda(), dm[1:2, D:2 * D], dm[1:2, 0:D], out, L,
                  n_mod=1, mod_ld=6 * D)
  ref = orc.modulate(ln, mod[1, None, None, D:2 * D], mod[1, None, None, 0:D]).view(B * L, D)
  err = (out.float().cpu() - ref).abs()
  assert (err <= 2e-2 * ref.abs().clamp(min=1.0)).all()


def _gemm_ref(A, W, bias=None):
  y = _bf(A).float() @ _bf(W).float().t()
  return y if bias is None else y + bias


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (256, 384, 128), (128, 1000, 192),
                                   (384, 2304, 768), (128, 777, 768), (512, 1000, 256),
                                   (1024, 768, 3072)])
def test_gemm_store(ops, M, N, K):
  from mdlm_amd._lib import EPI_STORE
  torch.manual_seed(2)
  A = torch.randn(M, K); W = torch.randn(N, K) * 0.1; bias = torch.randn(N)
  ref = _gemm_ref(A, W, bias)
  dA, dW = _bf(A).cuda(), _bf(W).cuda()
  C = torch.zeros(M, N, device='cuda')
  ops.gemm_bf16(dA, dW, C, EPI_STORE, bias=bias.cuda())
  assert _rel(C.cpu(), ref) < 1e-5
  assert (C.cpu() - ref).abs().max().item() < 1e-3
  # bf16 output into a padded leading dimension
  ld = (N + 127) // 128 * 128
  Cb = torch.zeros(M, ld, dtype=torch.bfloat16, device='cuda')
  ops.gemm_bf16(dA, dW, Cb[:, :N], EPI_STORE, bias=bias.cuda())
  assert ((Cb[:, :N].float().cpu() - ref).abs() <= 2e-2 * ref.abs().clamp(min=1.0)).all()
  assert (Cb[:, N:] == 0).all()  # nothing written past column N
  # unaligned ldc (N odd multiple) falls back to scalar stores
  Cu = torch.zeros(M, N, dtype=torch.bfloat16, device='cuda')
  ops.gemm_bf16(dA, dW, Cu, EPI_STORE, bias=bias.cuda())
  assert torch.equal(Cu, Cb[:, :N])


def test_gemm_identity_asymmetric(ops):
  """A = I with an asymmetric W catches row/col swaps in the C write."""
  from mdlm_amd._lib import EPI_STORE
  M = K = 128; N = 256
  A = torch.eye(M, K)
  W = (torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 5 % 7).float() % 64
  C = torch.zeros(M, N, device='cuda')
  ops.gemm_bf16(_bf(A).cuda(), _bf(W).cuda(), C, EPI_STORE)
  assert torch.equal(C.cpu(), _bf(W).float().t())


def test_gemm_live_rows_and_scatter(ops):
  from mdlm_amd._lib import EPI_STORE
  torch.manual_seed(3)
  L, B, N, K = 256, 4, 320, 64
  A = torch.randn(B * L, K); W = torch.randn(N, K)
  ref = _gemm_ref(A, W)
  C = torch.full((B * L, N), -5.0, device='cuda')
  live = torch.tensor([2 * L], dtype=torch.int32).cuda()
  amap = torch.tensor([3, 1, 0, 0], dtype=torch.int32).cuda()
  ops.gemm_bf16(_bf(A).cuda(), _bf(W).cuda(), C, EPI_STORE, rows_per_mod=L,
                m_rows_dev=live, out_sample_map=amap)
  Cc = C.cpu()
  assert _rel(Cc[3 * L:4 * L], ref[0:L]) < 1e-5      # compact sample 0 -> slot 3
  assert _rel(Cc[1 * L:2 * L], ref[L:2 * L]) < 1e-5  # compact sample 1 -> slot 1
  assert (Cc[0:L] == -5).all() and (Cc[2 * L:3 * L] == -5).all()


def test_gemm_gelu_and_gate_residual(ops):
  from mdlm_amd._lib import EPI_BIAS_GELU, EPI_GATE_RESIDUAL
  torch.manual_seed(4)
  B, L, N, K = 2, 256, 512, 128
  M = B * L
  A = torch.randn(M, K); W = torch.randn(N, K) * 0.2; bias = torch.randn(N)
  y = _gemm_ref(A, W, bias)
  Cg = torch.zeros(M, N, dtype=torch.bfloat16, device='cuda')
  ops.gemm_bf16(_bf(A).cuda(), _bf(W).cuda(), Cg, EPI_BIAS_GELU, bias=bias.cuda())
  ref = F.gelu(y, approximate='tanh')
  assert ((Cg.float().cpu() - ref).abs() <= 2e-2 * ref.abs().clamp(min=1.0)).all()
  # gate * (acc + bias) + residual, per-sample gate rows 6N apart, in place
  mod = torch.randn(B, 6 * N)
  res = torch.randn(M, N)
  X = res.clone().cuda()
  dm = mod.cuda()
  ops.gemm_bf16(_bf(A).cuda(), _bf(W).cuda(), X, EPI_GATE_RESIDUAL, bias=bias.cuda(),
                gate=dm[:, 2 * N:3 * N], residual=X, rows_per_mod=L, n_mod=B,
                mod_ld=6 * N)
  gate = mod[:, 2 * N:3 * N].repeat_interleave(L, 0)
  ref = res + gate * y
  assert (X.cpu() - ref).abs().max().item() < 2e-3
  # shared gate (n_mod = 1), no bias
  X = res.clone().cuda()
  ops.gemm_bf16(_bf(A).cuda(), _bf(W).cuda(), X, EPI_GATE_RESIDUAL,
                gate=dm[1:2, 0:N], residual=X, rows_per_mod=L, n_mod=1, mod_ld=6 * N)
  ref = res + mod[1, 0:N][None] * _gemm_ref(A, W)
  assert (X.cpu() - ref).abs().max().item() < 2e-3


def _qkv_ref(h, Wqkv, B, L, H):
  """bf16 Linear output -> rotary (bf16 cos/sin, fp32 math) -> bf16."""
  qkv = _bf(_gemm_ref(h, Wqkv)).float().view(B, L, 3, H, 64)
  cos, sin = orc.rotary_cos_sin(L, 64)
  rot = orc.apply_rotary_qk(qkv, _bf(cos).float(), _bf(sin).float())
  return _bf(rot).float()


@pytest.mark.parametrize('H', [1, 2, 3, 12])
def test_gemm_qkv_rotary_layout(ops, H):
  """q/k rotary + v^T layout; N = 3*H*64 is not a multiple of the 256-wide
  tile for H in {1, 2, 3}: wave tiles past the N edge must write nothing
  (guard bands on both sides of the output stay intact) and the result must
  not depend on the run."""
  from mdlm_amd._lib import EPI_QKV_ROTARY
  torch.manual_seed(5)
  B, L = 2, 256
  D = H * 64
  M = B * L
  h = torch.randn(M, D); Wqkv = torch.randn(3 * D, D) * 0.1
  ref = _qkv_ref(h, Wqkv, B, L, H)            # (B, L, 3, H, 64)
  cos, sin = orc.rotary_cos_sin(L, 64)
  G = 4096
  buf = torch.full((G + 3 * M * D + G,), 7.0, dtype=torch.bfloat16, device='cuda')
  out = buf[G:G + 3 * M * D]
  args = dict(rows_per_mod=L, rot_cos=_bf(cos).float().cuda(),
              rot_sin=_bf(sin).float().cuda(), n_heads=H, ldc=0)
  ops.gemm_bf16(_bf(h).cuda(), _bf(Wqkv).cuda(), out, EPI_QKV_ROTARY, **args)
  first = out.clone()
  assert (buf[:G] == 7.0).all() and (buf[G + 3 * M * D:] == 7.0).all()
  for _ in range(3):
    ops.gemm_bf16(_bf(h).cuda(), _bf(Wqkv).cuda(), out, EPI_QKV_ROTARY, **args)
    assert torch.equal(out, first)
  o = out.float().cpu()
  q = o[:M * D].view(B, H, L, 64)
  k = o[M * D:2 * M * D].view(B, H, L, 64)
  vt = o[2 * M * D:].view(B, H, 64, L)
  tol = lambda a, b: ((a - b).abs() <= 2e-2 * b.abs().clamp(min=1.0)).all()
  assert tol(q, ref[:, :, 0].permute(0, 2, 1, 3))
  assert tol(k, ref[:, :, 1].permute(0, 2, 1, 3))
  assert tol(vt, ref[:, :, 2].permute(0, 2, 3, 1))


@pytest.mark.parametrize('L', [128, 256, 1024])
def test_attention(ops, L):
  torch.manual_seed(6)
  B, H = 2, 3
  q = torch.randn(B, H, L, 64); k = torch.randn(B, H, L, 64); v = torch.randn(B, H, L, 64)
  # spike one key against one query so the running max jumps mid-sequence
  k[0, 0, L // 2 + 5] = q[0, 0, 7] * 3
  qb, kb, vb = _bf(q), _bf(k), _bf(v)
  ref = F.scaled_dot_product_attention(qb.float(), kb.float(), vb.float())
  ref = ref.transpose(1, 2).reshape(B * L, H * 64)
  out = torch.zeros(B * L, H * 64, dtype=torch.bfloat16, device='cuda')
  ops.attn_fwd(qb.cuda(), kb.cuda(), vb.transpose(2, 3).contiguous().cuda(), out, B, H, L)
  err = (out.float().cpu() - ref).abs()
  assert err.max().item() < 2e-2, err.max().item()
  assert _rel(out.float().cpu(), ref) < 1e-2
  # device-side live-sequence count
  out2 = torch.full_like(out, 9.0)
  live = torch.tensor([1], dtype=torch.int32).cuda()
  ops.attn_fwd(qb.cuda(), kb.cuda(), vb.transpose(2, 3).contiguous().cuda(), out2, B, H, L,
               n_seq_dev=live)
  assert torch.equal(out2[:L], out[:L]) and (out2[L:] == 9).all()


@pytest.mark.parametrize('L', [128, 256, 1024])
def test_attention_scale_guard(ops, L):
  """The kernels keep the first key block's max as the softmax scale and only
  re-scale when a row sum overflows.  Scores that grow along the sequence by far
  more than fp32's exponent range (~2^450 in the exp2 domain here) force that
  path repeatedly; rows that see only tiny or hugely negative scores afterwards
  exercise the other direction.  Against SDPA in fp32 on the same bf16 inputs
  (flash-attn semantics, dit.py:265-275)."""
  torch.manual_seed(8)
  B, H = 1, 2
  q = torch.randn(B, H, L, 64); k = torch.randn(B, H, L, 64); v = torch.randn(B, H, L, 64)
  ramp = torch.linspace(0, 1, L)[None, :, None]
  k[0, 0] = k[0, 0] * 0.1 + q[0, 0, :1] * ramp[0] * 16.0     # head 0: scores rise steadily for query 0
  k[0, 1] = k[0, 1] * 0.1 - q[0, 1, 3:4] * ramp[0] * 16.0    # head 1: scores fall for query 3
  qb, kb, vb = _bf(q), _bf(k), _bf(v)
  s = (qb.float() @ kb.float().transpose(-1, -2)) * 0.125 * 1.4426950408889634
  assert (s.max(-1).values - s[..., 0]).max().item() > 150     # far beyond the 2^100 guard
  ref = F.scaled_dot_product_attention(qb.float(), kb.float(), vb.float())
  ref = ref.transpose(1, 2).reshape(B * L, H * 64)
  out = torch.zeros(B * L, H * 64, dtype=torch.bfloat16, device='cuda')
  ops.attn_fwd(qb.cuda(), kb.cuda(), vb.transpose(2, 3).contiguous().cuda(), out, B, H, L)
  err = (out.float().cpu() - ref).abs()
  assert torch.isfinite(out.float()).all()
  assert err.max().item() < 3e-2, err.max().item()
  assert _rel(out.float().cpu(), ref) < 1e-2


def _make_model(hidden, heads, blocks, cond, V, seed):
  from mdlm_amd.dit import DIT
  cfg = {'model': dict(hidden_size=hidden, cond_dim=cond, n_heads=heads,
                       n_blocks=blocks, dropout=0.1, scale_by_sigma=True)}
  p = orc.synthetic_dit_params(hidden, heads, blocks, cond, V, seed=seed)
  m = DIT(cfg, vocab_size=V)
  sd = dict(p)
  sd['rotary_emb.inv_freq'] = m.rotary_emb.inv_freq
  m.load_state_dict(sd)
  return m.cuda(), p


@pytest.mark.parametrize('shape', [(128, 2, 2, 64, 300, 128), (768, 12, 2, 128, 1000, 256), (128, 2, 1, 64, 700, 512),
                                   # the widths of the reference's other DiT configs (configs/model/medium.yaml:
                                   # 1024 / 16 heads; tiny.yaml: 512 / 8 heads), two blocks each
                                   (1024, 16, 2, 128, 1000, 256), (512, 8, 2, 128, 1000, 256)])
def test_dit_forward_vs_oracle(ops, shape):
  hidden, heads, blocks, cond, V, L = shape
  model, p = _make_model(hidden, heads, blocks, cond, V, seed=3)
  torch.manual_seed(7)
  B = 2
  idx = torch.randint(0, V, (B, L))
  for sigma in (torch.zeros(B), torch.tensor([0.4, 1.7])):
    ref = orc.dit_forward(p, idx, sigma, heads, blocks)
    # DIT.run picks 128x128 GEMM tiles for launches of a few samples and 256x256 tiles
    # otherwise (small_tile_limit): both against the oracle, and bit-identical to each other
    outs = []
    for lim in (0, 1 << 30):
      model.small_tile_limit = lim
      out = model(idx.cuda(), sigma.cuda())
      assert out.shape == (B, L, V) and out.dtype == torch.bfloat16
      o = out.float().cpu()
      assert _rel(o, ref) < 2e-2, _rel(o, ref)
      # log-probs within the bf16 tolerance
      lp = torch.log_softmax(o, -1); lr = torch.log_softmax(ref, -1)
      # bf16 logits: the absolute error grows with the logit magnitude (|logit| up to 16 at the
      # widest shape: one bf16 ulp there is 0.0625)
      assert (lp - lr).abs().max().item() < (6e-2 if hidden <= 768 else 1e-1)
      assert (lp - lr).abs().mean().item() < 1e-2
      outs.append(out.clone())
    assert torch.equal(outs[0], outs[1])
    model.small_tile_limit = 128
  # empty batch: every kernel entry point is a no-op
  e = model(torch.zeros(0, L, dtype=torch.int64).cuda(), torch.zeros(0).cuda())
  assert e.shape == (0, L, V)
  # constant-conditioning shortcut (sigma zeroed) == general path
  out0 = model.run(idx.cuda(), torch.zeros(B).cuda(),
                   ops.alloc_logits(B, L, V), sigma_is_zero=True)
  out1 = model(idx.cuda(), torch.zeros(B).cuda())
  assert torch.equal(out0, out1)


def test_masked_row_projection(ops):
  """Final LN + vocab GEMM restricted to the masked rows of the active samples
  == the same rows of the all-rows run, bit for bit; other rows untouched."""
  hidden, heads, blocks, cond, V, L = 128, 2, 2, 64, 300, 256
  model, _ = _make_model(hidden, heads, blocks, cond, V, seed=6)
  torch.manual_seed(9)
  B, mask = 4, V - 1
  idx = torch.randint(0, V - 1, (B, L))
  idx[torch.rand(B, L) < 0.4] = mask
  idx[2] = mask                      # fully masked sample
  idx[3, :] = 5                      # fully unmasked (inactive below)
  idx = idx.cuda()
  z = torch.zeros(B).cuda()
  full = model.run(idx, z, ops.alloc_logits(B, L, V), sigma_is_zero=True).clone()
  active = torch.tensor([2, 0, 1, 0], dtype=torch.int32).cuda()
  n_act = torch.tensor([3], dtype=torch.int32).cuda()
  n_rows = torch.tensor([3 * L], dtype=torch.int32).cuda()
  gather = torch.full((B * L + 256,), -9, dtype=torch.int32).cuda()
  scatter = torch.full((B * L + 256,), -9, dtype=torch.int32).cuda()
  n_m = torch.zeros(1, dtype=torch.int32).cuda()
  ops.masked_row_map(idx, active, n_act, L, mask, gather, scatter, n_m)
  # reference row lists on the host
  g_ref, s_ref = [], []
  for i, b in enumerate([2, 0, 1]):
    for pos in range(L):
      if idx[b, pos].item() == mask:
        g_ref.append(i * L + pos); s_ref.append(b * L + pos)
  n = len(g_ref); npad = (n + 255) // 256 * 256
  assert n_m.item() == npad
  assert gather[:n].tolist() == g_ref and scatter[:n].tolist() == s_ref
  assert (gather[n:npad] == 0).all() and (scatter[n:npad] == -1).all()
  cache = ops.alloc_logits(B, L, V); cache.fill_(7.0)
  model.run(idx, z, cache, sigma_is_zero=True, active=active, n_active_dev=n_act,
            n_rows_dev=n_rows, n_max=B, row_map=(gather, scatter, n_m))
  m = (idx == mask); m[3] = False
  assert torch.equal(cache[m], full[m])
  assert (cache[~m] == 7.0).all()


def test_dit_active_compaction_is_batch_invariant(ops):
  """Per-sample evaluation must be bit-identical to the full-batch one."""
  hidden, heads, blocks, cond, V, L = 128, 2, 2, 64, 300, 256
  model, _ = _make_model(hidden, heads, blocks, cond, V, seed=4)
  torch.manual_seed(8)
  B = 5
  idx = torch.randint(0, V, (B, L)).cuda()
  z = torch.zeros(B).cuda()
  full = model.run(idx, z, ops.alloc_logits(B, L, V), sigma_is_zero=True).clone()
  cache = ops.alloc_logits(B, L, V); cache.fill_(0)
  active = torch.tensor([4, 1, 2, 0, 0], dtype=torch.int32).cuda()
  n_act = torch.tensor([3], dtype=torch.int32).cuda()
  n_rows = torch.tensor([3 * L], dtype=torch.int32).cuda()
  model.run(idx, z, cache, sigma_is_zero=True, active=active, n_active_dev=n_act,
            n_rows_dev=n_rows, n_max=B)
  for b in (4, 1, 2):
    assert torch.equal(cache[b], full[b])
  assert (cache[0] == 0).all() and (cache[3] == 0).all()


@pytest.mark.parametrize('N,mask', [(50258, 50257), (30522, 103), (1000, 999)])
def test_vocab_gemm_row_lse_from_epilogue(ops, N, mask):
  """the SUBS row statistics (diffusion.py:263-268) come out of
  the vocabulary GEMM's epilogue (dit.py:317-321) as (max, sum exp) pairs per
  64-column group; mdlm_lse_combine folds them.  Must equal the row lse of the
  STORED bf16 logits (mask column shifted by -1e6, i.e. excluded) to 1e-5, for
  the N-edge group (50258 = 785*64 + 18), a mask column inside the vocabulary
  (bert: 103) and a scattered / partly dropped row list."""
  torch.manual_seed(2)
  from mdlm_amd._lib import EPI_STORE
  M, K = 512, 256
  A = _bf(torch.randn(M, K) * 2).cuda()
  W = _bf(torch.randn(N, K) * 0.3).cuda()
  b = torch.randn(N).cuda()
  rows_out = 700
  perm = torch.randperm(rows_out)[:M].to(torch.int32)
  perm[5] = -1; perm[300:310] = -1          # dropped rows (padding of the row list)
  scatter = perm.cuda()
  live = torch.tensor([M], dtype=torch.int32).cuda()
  logits = torch.zeros(rows_out, ops.padded_vocab(N), dtype=torch.bfloat16).cuda()
  part = torch.empty(ops.workspace_bytes('gemm_lse_partials', M, N) // 4).cuda()
  lse = torch.full((rows_out,), float('nan')).cuda()
  ops.gemm_bf16(A, W, logits, EPI_STORE, bias=b, out_row_map=scatter, m_rows_dev=live,
                ldc=logits.stride(0), lse_partials=part, lse_skip_col=mask)
  ops.lse_combine(part, M, N, lse, out_row_map=scatter, n_rows_dev=live)
  lg = logits[:, :N].float()
  lg[:, mask] = float('-inf')
  ref = torch.logsumexp(lg, -1)
  kept = perm[perm >= 0].long()
  assert (lse[kept] - ref[kept]).abs().max().item() < 1e-5
  rest = torch.ones(rows_out, dtype=torch.bool); rest[kept] = False
  assert torch.isnan(lse[rest.cuda()]).all()   # rows not in the list are not touched
  # and against the stand-alone kernel on the same stored logits
  x = torch.full((rows_out, 1), mask, dtype=torch.int64).cuda()
  lse2 = ops.subs_row_lse(logits[:, None, :N], x, mask)
  assert (lse[kept] - lse2.view(-1)[kept]).abs().max().item() < 1e-5


def test_gemm_half_tile_tail_is_bit_identical(ops):
  """The persistent GEMM cuts the tiles of a short last round into half tiles
  (same K order per output row).  At 44 samples x 1024 rows the N = 768 GEMMs
  have 528 tiles on 256 workgroups: two full rounds + 16 tiles run as 32 half
  tiles.  Results must equal the 128x128 configuration's bit for bit (which never
  splits), for the gate/residual and the QKV + rotary epilogues, with a device-side
  row count below the launch bound."""
  from mdlm_amd._lib import EPI_GATE_RESIDUAL, EPI_QKV_ROTARY
  torch.manual_seed(11)
  L, D, H = 1024, 768, 12
  n_max, n_live = 48, 44
  M = n_max * L
  h = _bf(torch.randn(M, D)).cuda()
  x0 = torch.randn(M, D).cuda()
  Wo = _bf(torch.randn(D, D) * 0.05).cuda()
  Wq = _bf(torch.randn(3 * D, D) * 0.05).cuda()
  mod = torch.randn(1, 6 * D).cuda()
  cos = torch.randn(L, 32).cuda(); sin = torch.randn(L, 32).cuda()
  live = torch.tensor([n_live * L], dtype=torch.int32).cuda()
  res = []
  for tile in (0, 1, 16):   # 16: whole tiles only (what concurrent stream groups pass)
    x = x0.clone()
    ops.gemm_bf16(h, Wo, x, EPI_GATE_RESIDUAL, gate=mod[:, :D], residual=x, rows_per_mod=L,
                  n_mod=1, mod_ld=6 * D, m_rows_dev=live, tile=tile)
    qkv = torch.zeros(3 * M * D, dtype=torch.bfloat16).cuda()
    ops.gemm_bf16(h, Wq, qkv, EPI_QKV_ROTARY, rows_per_mod=L, rot_cos=cos, rot_sin=sin,
                  n_heads=H, m_rows_dev=live, ldc=0, tile=tile)
    res.append((x, qkv))
  for other in res[1:]:
    assert torch.equal(res[0][0], other[0])
    assert torch.equal(res[0][1], other[1])
  assert torch.equal(res[0][0][n_live * L:], x0[n_live * L:])      # rows past the live count untouched
  ref = x0[:4096] + mod[:, :D] * (h[:4096].float() @ Wo.float().T)
  assert _rel(res[0][0][:4096].cpu(), ref.cpu()) < 1e-2


def test_gemm_two_accumulator_set_kernel_is_bit_identical(ops):
  """`tile | 32`: the 128x256 kernel with two accumulator sets (csrc/gemm_pp.hip: the
  finished tile leaves from registers in units inside the next tile's K loop, W rows
  permuted in LDS for the bf16 outputs) keeps the K order per output row, so its results
  must equal the 256x256 kernel's bit for bit: gate/residual (out-projection K = 768,
  MLP-down K = 3072 with bias) and bias + GELU (models/dit.py:277-287, 225-228), with one,
  two and three tiles per workgroup and a device-side row count below the launch bound."""
  from mdlm_amd._lib import EPI_BIAS_GELU, EPI_GATE_RESIDUAL
  torch.manual_seed(12)
  L, D = 1024, 768
  for n_max, n_live in ((2, 2), (16, 13), (24, 24)):
    M = n_max * L
    h = _bf(torch.randn(M, D)).cuda()
    mh = _bf(torch.randn(M, 4 * D)).cuda()
    x0 = torch.randn(M, D).cuda()
    Wo = _bf(torch.randn(D, D) * 0.05).cuda()
    W2 = _bf(torch.randn(D, 4 * D) * 0.03).cuda(); b2 = (torch.randn(D) * 0.1).cuda()
    W1 = _bf(torch.randn(4 * D, D) * 0.05).cuda(); b1 = (torch.randn(4 * D) * 0.1).cuda()
    mod = torch.randn(1, 6 * D).cuda()
    live = torch.tensor([n_live * L], dtype=torch.int32).cuda()
    res = []
    for tile in (0, 32):
      x = x0.clone()
      ops.gemm_bf16(h, Wo, x, EPI_GATE_RESIDUAL, gate=mod[:, :D], residual=x, rows_per_mod=L,
                    n_mod=1, mod_ld=6 * D, m_rows_dev=live, tile=tile)
      ops.gemm_bf16(mh, W2, x, EPI_GATE_RESIDUAL, bias=b2, gate=mod[:, D:2 * D], residual=x,
                    rows_per_mod=L, n_mod=1, mod_ld=6 * D, m_rows_dev=live, tile=tile)
      act = torch.zeros(M, 4 * D, dtype=torch.bfloat16).cuda()
      ops.gemm_bf16(h, W1, act, EPI_BIAS_GELU, bias=b1, m_rows_dev=live, tile=tile)
      res.append((x, act))
    assert torch.equal(res[0][0], res[1][0]), (n_max, n_live)
    assert torch.equal(res[0][1], res[1][1]), (n_max, n_live)
    assert torch.equal(res[1][0][n_live * L:], x0[n_live * L:])      # rows past the live count untouched
    assert not res[1][1][n_live * L:].any()
    y = x0[:2048] + mod[:, :D] * (h[:2048].float() @ Wo.float().T)
    y = y + mod[:, D:2 * D] * (mh[:2048].float() @ W2.float().T + b2)
    assert _rel(res[1][0][:2048].cpu(), y.cpu()) < 1e-2


def test_gemm_sample_scatter_leaves_dead_rows_of_a_partly_live_tile_alone(ops):
  """STORE epilogue with out_sample_map and samples shorter than a tile (L = 128,
  256-row tiles): with 3 live samples the second tile is half dead, and its dead
  half belongs to list entry 3 -- a stale sample id.  That sample's rows must
  keep their cached contents (found by the stream-group test: a cached sample's
  logits were overwritten from a dead workspace slot)."""
  from mdlm_amd._lib import EPI_STORE
  torch.manual_seed(3)
  L, D, N, n_max = 128, 256, 512, 4
  M = n_max * L
  h = _bf(torch.randn(M, D)).cuda()
  W = _bf(torch.randn(N, D) * 0.05).cuda()
  amap = torch.tensor([5, 0, 2, 7], dtype=torch.int32).cuda()     # entry 3 (sample 7) is stale
  live = torch.tensor([3 * L], dtype=torch.int32).cuda()
  for tile in (0, 1):
    out = torch.full((8 * L, N), 9.0, dtype=torch.bfloat16).cuda()
    ops.gemm_bf16(h, W, out, EPI_STORE, rows_per_mod=L, m_rows_dev=live, out_sample_map=amap, tile=tile)
    ref = (h.float() @ W.float().T).view(n_max, L, N)
    o = out.float().view(8, L, N)
    for slot, sample in enumerate((5, 0, 2)):
      assert _rel(o[sample].cpu(), ref[slot].cpu()) < 1e-2
    for sample in (1, 3, 4, 6, 7):
      assert (o[sample] == 9.0).all(), (tile, sample)
