"""Thin torch-tensor wrappers over the C-ABI (include/mdlm_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic
runs in the hand-written HIP kernels.  Every wrapper requires CUDA (ROCm)
tensors and raises otherwise -- there is no CPU path.
"""
import torch

from . import _lib
from ._lib import BF16, F32, call

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def _stream():
  return torch.cuda.current_stream().cuda_stream


def _dev(t, name, dtype=None):
  if not isinstance(t, torch.Tensor) or not t.is_cuda:
    raise RuntimeError(
      f'{name}: expected a CUDA/ROCm tensor; mdlm_amd has no CPU fallback')
  if dtype is not None and t.dtype != dtype:
    raise RuntimeError(f'{name}: expected dtype {dtype}, got {t.dtype}')
  return t


def _rows2d(t, name):
  """View (..., V) as rows with a leading dimension; innermost contiguous."""
  _dev(t, name)
  if t.dtype not in _DT:
    raise RuntimeError(f'{name}: dtype {t.dtype} not supported (fp32/bf16)')
  if t.stride(-1) != 1:
    raise RuntimeError(f'{name}: innermost dimension must be contiguous')
  V = t.shape[-1]
  if t.dim() == 2:
    return t, t.shape[0], V, t.stride(0)
  if t.dim() == 3:
    B, L, _ = t.shape
    ld = t.stride(1)
    if B > 1 and t.stride(0) != L * ld:
      raise RuntimeError(f'{name}: rows must be uniformly strided')
    return t, B * L, V, ld
  raise RuntimeError(f'{name}: expected 2-D or 3-D tensor')


def _p(t):
  return 0 if t is None else t.data_ptr()


def padded_vocab(V, multiple=128):
  return (V + multiple - 1) // multiple * multiple


def alloc_logits(B, L, V, dtype=torch.bfloat16, device='cuda'):
  """(B, L, V) view of a buffer whose rows are padded to a multiple of 128
  elements, so every row starts 16-byte aligned (vector loads/stores)."""
  buf = torch.empty(B, L, padded_vocab(V), dtype=dtype, device=device)
  return buf[:, :, :V]


# ------------------------------------------------------------------ sampler
def subs_logprobs(logits, xt, mask_index, out=None, out_dtype=None,
                  as_probs=False):
  lg, rows, V, ld = _rows2d(logits, 'logits')
  xt = _dev(xt, 'xt', torch.int64).contiguous()
  assert xt.numel() == rows
  if out is None:
    od = out_dtype or logits.dtype
    if logits.dim() == 3:
      out = alloc_logits(logits.shape[0], logits.shape[1], V, od, logits.device)
    else:
      out = torch.empty(rows, padded_vocab(V), dtype=od,
                        device=logits.device)[:, :V]
  o, orows, oV, old = _rows2d(out, 'out')
  assert orows == rows and oV == V
  call('mdlm_subs_logprobs', lg.data_ptr(), ld, _DT[lg.dtype], xt.data_ptr(),
       o.data_ptr(), old, _DT[o.dtype], rows, V, int(mask_index),
       int(bool(as_probs)), _stream())
  return out


def subs_row_lse(logits, xt, mask_index, out=None, sample_map=None, L=0,
                 n_rows_dev=None):
  lg, rows, V, ld = _rows2d(logits, 'logits')
  xt = _dev(xt, 'xt', torch.int64).contiguous()
  if out is None:
    out = torch.zeros(rows, dtype=torch.float32, device=logits.device)
  call('mdlm_subs_row_lse', lg.data_ptr(), ld, _DT[lg.dtype], xt.data_ptr(),
       out.data_ptr(), rows, V, int(mask_index), _p(sample_map), int(L),
       _p(n_rows_dev), _stream())
  return out


def subs_argmax(logits, xt, mask_index, out=None):
  lg, rows, V, ld = _rows2d(logits, 'logits')
  xt = _dev(xt, 'xt', torch.int64).contiguous()
  if out is None:
    out = torch.empty_like(xt)
  call('mdlm_subs_argmax', lg.data_ptr(), ld, _DT[lg.dtype], xt.data_ptr(),
       out.data_ptr(), rows, V, int(mask_index), _stream())
  return out


def _t_vec(t, B):
  t = _dev(t, 't')
  if t.dim() > 1:
    t = t.squeeze(-1)  # diffusion.py:595-596
  assert t.dim() == 1 and t.shape[0] == B
  return t.to(torch.float32).contiguous()


def ddpm_cache_update(p_x0, x, t, dt, mask_index, u=None, seed=0, offset=0,
                      changed=None, fast_math=False, x_next=None, n_exact=None):
  """(b) from probabilities.  Returns x_next (and fills `changed` (B,) int32).
  fast_math: same token ids through the cheaper two-pass race (include/mdlm_hip.h);
  n_exact (optional, device int64[1]) counts the rows it re-evaluated."""
  p, rows, V, ld = _rows2d(p_x0, 'p_x0')
  x = _dev(x, 'x', torch.int64).contiguous()
  B, L = x.shape
  assert rows == B * L
  tt = _t_vec(t, B)
  if u is not None:
    u = _dev(u, 'u', torch.float32).contiguous()
    assert u.numel() == rows * V
  if x_next is None:
    x_next = torch.empty_like(x)
  call('mdlm_ddpm_cache_update', p.data_ptr(), ld, _DT[p.dtype], x.data_ptr(),
       tt.data_ptr(), float(dt), _p(u), int(seed), int(offset),
       x_next.data_ptr(), _p(changed), B, L, V, int(mask_index),
       int(bool(fast_math)), _p(n_exact), _stream())
  return x_next


def subs_ddpm_update(logits, lse, x, t, dt, mask_index, u=None, seed=0,
                     offset=0, changed=None, fast_math=False, x_next=None,
                     fresh=None, n_exact=None):
  """(a)+(b) fused from raw logits + row lse."""
  lg, rows, V, ld = _rows2d(logits, 'logits')
  x = _dev(x, 'x', torch.int64).contiguous()
  B, L = x.shape
  assert rows == B * L
  lse = _dev(lse, 'lse', torch.float32).contiguous()
  assert lse.numel() == rows
  tt = _t_vec(t, B)
  if u is not None:
    u = _dev(u, 'u', torch.float32).contiguous()
    assert u.numel() == rows * V
  if x_next is None:
    x_next = torch.empty_like(x)
  call('mdlm_subs_ddpm_update', lg.data_ptr(), ld, _DT[lg.dtype],
       lse.data_ptr(), _p(fresh), x.data_ptr(), tt.data_ptr(), float(dt), _p(u),
       int(seed), int(offset), x_next.data_ptr(), _p(changed), B, L, V,
       int(mask_index), int(bool(fast_math)), _p(n_exact), _stream())
  return x_next


def subs_race_update(logits, lse, x, q_scale, q_mask, mask_index, u=None,
                     seed=0, offset=0, changed=None, fast_math=False,
                     x_next=None, fresh=None, n_exact=None):
  """Generic masked-row race (samplers 'ddpm' / 'analytic' / denoiser)."""
  lg, rows, V, ld = _rows2d(logits, 'logits')
  x = _dev(x, 'x', torch.int64).contiguous()
  B, L = x.shape
  assert rows == B * L
  lse = _dev(lse, 'lse', torch.float32).contiguous()
  assert lse.numel() == rows
  qs = _dev(q_scale, 'q_scale').to(torch.float32).reshape(-1).contiguous()
  qm = _dev(q_mask, 'q_mask').to(torch.float32).reshape(-1).contiguous()
  assert qs.numel() == B and qm.numel() == B
  if u is not None:
    u = _dev(u, 'u', torch.float32).contiguous()
    assert u.numel() == rows * V
  if x_next is None:
    x_next = torch.empty_like(x)
  call('mdlm_subs_race_update', lg.data_ptr(), ld, _DT[lg.dtype],
       lse.data_ptr(), _p(fresh), x.data_ptr(), qs.data_ptr(), qm.data_ptr(),
       _p(u), int(seed), int(offset), x_next.data_ptr(), _p(changed), B, L, V,
       int(mask_index), int(bool(fast_math)), _p(n_exact), _stream())
  return x_next


def race_key_probe(rbits):
  """Diagnostic: (g~ and log2 g~ of the approximate pass, defining g) for
  uint32 bit patterns held in an int32 tensor."""
  rbits = _dev(rbits, 'rbits', torch.int32).contiguous()
  n = rbits.numel()
  g, a, b = (torch.empty(n, dtype=torch.float32, device=rbits.device) for _ in range(3))
  call('mdlm_race_key_probe', rbits.data_ptr(), g.data_ptr(), a.data_ptr(),
       b.data_ptr(), n, _stream())
  return g, a, b


def d3pm_loss(neg_logp_x0, xt, t, T, mask_index, attn_mask=None):
  nl = _dev(neg_logp_x0, 'neg_logp_x0', torch.float32).contiguous()
  xt = _dev(xt, 'xt', torch.int64).contiguous()
  B, L = xt.shape
  tt = _dev(t, 't').to(torch.float32).reshape(-1).contiguous()
  assert tt.numel() == B and nl.shape == xt.shape
  if attn_mask is not None:
    attn_mask = _dev(attn_mask, 'attn_mask').to(torch.float32).contiguous()
  out = torch.empty(B, L, dtype=torch.float32, device=xt.device)
  call('mdlm_d3pm_loss', nl.data_ptr(), xt.data_ptr(), tt.data_ptr(), int(T),
       _p(attn_mask), out.data_ptr(), B, L, int(mask_index), _stream())
  return out


def lazy_select(x, t, dt, mask_index, x_next, rows, n_rows, changed, seed=0,
                offset=0, pad=256):
  """Stage 1 of the opt-in two-stage step (see include/mdlm_hip.h)."""
  x = _dev(x, 'x', torch.int64).contiguous()
  B, L = x.shape
  tt = _t_vec(t, B)
  call('mdlm_lazy_select', x.data_ptr(), tt.data_ptr(), float(dt), int(seed),
       int(offset), B, L, int(mask_index), int(pad), x_next.data_ptr(),
       rows.data_ptr(), n_rows.data_ptr(), changed.data_ptr(), _stream())
  return x_next


def lazy_race(logits, rows, n_rows, x_next, V, mask_index, seed=0, offset=0):
  """Stage 2: logits (max_rows, >=V) dense rows of the listed positions."""
  _dev(logits, 'logits')
  assert logits.dim() == 2 and logits.stride(1) == 1
  call('mdlm_lazy_race', logits.data_ptr(), logits.stride(0), _DT[logits.dtype],
       rows.data_ptr(), n_rows.data_ptr(), logits.shape[0], int(seed),
       int(offset), x_next.data_ptr(), int(V), int(mask_index), _stream())
  return x_next


def scatter_rows_f32(src, dst, sample_map, L, n_rows_dev=None):
  src = _dev(src, 'src', torch.float32)
  dst = _dev(dst, 'dst', torch.float32)
  assert src.is_contiguous() and dst.is_contiguous() and src.shape[1] == dst.shape[1]
  call('mdlm_scatter_rows_f32', src.data_ptr(), dst.data_ptr(), _p(sample_map),
       int(L), src.shape[1], src.shape[0], _p(n_rows_dev), _stream())
  return dst


def gather_rows(x_src, x_dst, a_src, a_dst, gather, n_rows_dev=None, rows=None):
  """Dense copies of the listed rows of the fp32 residual stream and of the
  bf16 attention output (see include/mdlm_hip.h)."""
  D = x_src.shape[1]
  if rows is None:
    rows = x_dst.shape[0]
  call('mdlm_gather_rows', x_src.data_ptr(), x_dst.data_ptr(), a_src.data_ptr(),
       a_dst.data_ptr(), gather.data_ptr(), D, int(rows), _p(n_rows_dev), _stream())


def philox_uniform(rows, V, seed, offset=0, device='cuda'):
  out = torch.empty(rows, V, dtype=torch.float32, device=device)
  call('mdlm_philox_uniform', out.data_ptr(), rows, V, int(seed), int(offset),
       _stream())
  return out


def q_xt(x0, move_chance, mask_index, u=None, seed=0, offset=0):
  x0 = _dev(x0, 'x0', torch.int64).contiguous()
  B, L = x0.shape
  mc = _dev(move_chance, 'move_chance').to(torch.float32).reshape(-1).contiguous()
  assert mc.numel() == B
  if u is not None:
    u = _dev(u, 'u', torch.float32).contiguous()
    assert u.shape == x0.shape
  xt = torch.empty_like(x0)
  call('mdlm_q_xt', x0.data_ptr(), mc.data_ptr(), _p(u), int(seed),
       int(offset), xt.data_ptr(), B, L, int(mask_index), _stream())
  return xt


def subs_nll(logits, xt, x0, weight, mask_index, attn_mask=None, row_lse=None):
  lg, rows, V, ld = _rows2d(logits, 'logits')
  xt = _dev(xt, 'xt', torch.int64).contiguous()
  x0 = _dev(x0, 'x0', torch.int64).contiguous()
  B, L = xt.shape
  assert rows == B * L
  w = _dev(weight, 'weight').to(torch.float32).reshape(-1).contiguous()
  assert w.numel() == B
  if attn_mask is not None:
    attn_mask = _dev(attn_mask, 'attn_mask').to(torch.float32).contiguous()
  nll = torch.empty(B, L, dtype=torch.float32, device=logits.device)
  call('mdlm_subs_nll', lg.data_ptr(), ld, _DT[lg.dtype], xt.data_ptr(),
       x0.data_ptr(), w.data_ptr(), _p(attn_mask), _p(row_lse), nll.data_ptr(), B, L, V,
       int(mask_index), _stream())
  return nll


def nll_sums(nll, attn_mask=None, sums=None):
  nll = _dev(nll, 'nll', torch.float32).contiguous()
  if attn_mask is not None:
    attn_mask = _dev(attn_mask, 'attn_mask').to(torch.float32).contiguous()
  if sums is None:
    sums = torch.zeros(2, dtype=torch.float64, device=nll.device)
  call('mdlm_nll_sums', nll.data_ptr(), _p(attn_mask), sums.data_ptr(),
       nll.numel(), _stream())
  return sums


def compact_active(changed, active, n_active, n_rows, L, all_or_nothing=False,
                   nfe_accum=None, fresh=None):
  call('mdlm_compact_active', changed.data_ptr(), active.data_ptr(),
       n_active.data_ptr(), n_rows.data_ptr(), changed.numel(), int(L),
       int(bool(all_or_nothing)), _p(nfe_accum), _p(fresh), _stream())


# ---------------------------------------------------------------- DiT ops
def embed_rows(table, idx, out, sample_map=None, L=1, n_rows_dev=None):
  n_rows, D = out.shape
  call('mdlm_embed_rows', table.data_ptr(), idx.data_ptr(), _p(sample_map),
       int(L), out.data_ptr(), n_rows, D, table.shape[0], _p(n_rows_dev),
       _stream())
  return out


def timestep_embedding(t, dim):
  t = _dev(t, 't', torch.float32).contiguous()
  out = torch.empty(t.shape[0], dim, dtype=torch.float32, device=t.device)
  call('mdlm_timestep_embedding', t.data_ptr(), out.data_ptr(), t.shape[0],
       dim, _stream())
  return out


def linear_f32(x, W, bias, act_in=0, act_out=0):
  x = _dev(x, 'x', torch.float32).contiguous()
  B, K = x.shape
  N = W.shape[0]
  assert W.shape[1] == K and W.is_contiguous() and W.dtype == torch.float32
  out = torch.empty(B, N, dtype=torch.float32, device=x.device)
  call('mdlm_linear_f32', x.data_ptr(), W.data_ptr(), _p(bias), out.data_ptr(),
       B, N, K, act_in, act_out, _stream())
  return out


def ln_modulate(x, ln_w, shift, scale, out, rows_per_mod, n_mod=1, mod_ld=0,
                n_rows_dev=None, row_gather=None, rows=None):
  """shift/scale: tensors (views allowed) whose row m starts m*mod_ld elements
  after data_ptr()."""
  D = x.shape[1]
  if rows is None:
    rows = x.shape[0]
  call('mdlm_ln_modulate', x.data_ptr(), ln_w.data_ptr(), shift.data_ptr(),
       scale.data_ptr(), out.data_ptr(), int(rows), D, int(rows_per_mod),
       int(n_mod), int(mod_ld), _p(row_gather), _p(n_rows_dev), _stream())
  return out


def gemm_bf16(A, W, C, epilogue=_lib.EPI_STORE, bias=None, gate=None,
              residual=None, rows_per_mod=1, n_mod=1, mod_ld=0, rot_cos=None,
              rot_sin=None, n_heads=0, m_rows_dev=None, out_sample_map=None,
              out_row_map=None, ldc=None, M=None, tile=None, lse_partials=None,
              lse_skip_col=-1):
  """C = epilogue(A @ W^T); see include/mdlm_hip.h (mdlm_gemm_bf16).  tile: 0 =
  the library chooses, 1 = 128x128 (None: DEFAULT_GEMM_TILE)."""
  if tile is None:
    tile = DEFAULT_GEMM_TILE
  if M is None:
    M = A.shape[0]
  K = A.shape[1]
  N = W.shape[0]
  assert A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16
  assert W.shape[1] == K and W.is_contiguous() and A.stride(1) == 1
  if ldc is None:
    ldc = C.stride(-2) if C.dim() >= 2 else 0
  call('mdlm_gemm_bf16', A.data_ptr(), A.stride(0), W.data_ptr(), _p(bias),
       C.data_ptr(), int(ldc), _DT[C.dtype], int(M), N, K, int(epilogue),
       _p(gate), _p(residual), int(rows_per_mod), int(n_mod), int(mod_ld),
       _p(rot_cos), _p(rot_sin), int(n_heads), _p(m_rows_dev),
       _p(out_sample_map), _p(out_row_map), _p(lse_partials), int(lse_skip_col),
       int(tile), _stream())
  return C


def lse_combine(partials, M, N, lse, out_row_map=None, n_rows_dev=None):
  """Row lse from the partials of gemm_bf16(..., lse_partials=...)."""
  call('mdlm_lse_combine', partials.data_ptr(), int(M), int(N), _p(out_row_map),
       _p(n_rows_dev), lse.data_ptr(), _stream())
  return lse


def splitk_plan(M, N, K_total, target_tiles=512, min_k=512):
  """Number of K ranges for a weight-gradient GEMM (M, N: output; K_total:
  tokens): the smallest power of two that gives the 256 CUs >= target_tiles
  tiles while each range keeps >= min_k of K."""
  t = 256 if M % 256 == 0 and N >= 256 else 128
  tiles = (M // t) * ((N + t - 1) // t)
  s = 1
  while (tiles * s < target_tiles and K_total % (2 * s * 64) == 0
         and K_total // (2 * s) >= min_k):
    s *= 2
  return s


def gemm_bf16_splitk(A, W, out, partials, splits=None):
  """out (M, N) fp32 = A (M, K) @ W (N, K)^T with K cut into `splits` ranges
  (mdlm_gemm_bf16_splitk + mdlm_splitk_reduce); `partials` is an fp32
  workspace of >= splits*M*N elements.  `out` may hold fewer than M rows (the
  leading rows are written)."""
  M, K = A.shape
  N = W.shape[0]
  assert A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16 and W.shape[1] == K
  assert A.stride(1) == 1 and W.stride(1) == 1 and out.is_contiguous()
  if splits is None:
    splits = splitk_plan(M, N, K)
  assert partials.numel() >= splits * M * N and partials.dtype == torch.float32
  call('mdlm_gemm_bf16_splitk', A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0),
       partials.data_ptr(), int(M), int(N), int(K), int(splits), _stream())
  call('mdlm_splitk_reduce', partials.data_ptr(), out.data_ptr(), out.numel(),
       int(splits), int(M * N), _stream())
  return out


def gemm_bf16_gelu(A, W, C, aux, bias=None, grad=False):
  """grad=False: aux <- A W^T + bias, C <- gelu_tanh(aux)   (mlp.0 forward, both saved)
  grad=True : C <- bf16(A W^T) * gelu_tanh'(aux)          (d(pre) in the dX GEMM of mlp.2)"""
  M, K = A.shape
  N = W.shape[0]
  assert A.dtype == W.dtype == C.dtype == aux.dtype == torch.bfloat16
  assert W.shape[1] == K and W.is_contiguous() and A.stride(1) == 1
  assert C.shape == aux.shape == (M, N) and C.stride(0) == aux.stride(0) and C.stride(1) == 1
  call('mdlm_gemm_bf16_gelu', A.data_ptr(), A.stride(0), W.data_ptr(), _p(bias),
       C.data_ptr(), aux.data_ptr(), C.stride(0), int(M), N, K, int(bool(grad)), _stream())
  return C


def tn_splitk_plan(N, K, T, max_splits=64, min_slabs=16):
  """Token ranges for mdlm_gemm_tn_splitk: fills the 256 persistent workgroups
  evenly (tiles * splits close below a multiple of 256) while a range keeps
  >= min_slabs 64-token slabs; cost model in slab units (a tile's 256 KiB
  epilogue ~ 5 slabs; reducing one more partial plane ~ 0.35 slabs per nine
  tiles of output)."""
  tiles = (N // 256) * (K // 256)
  slabs = T // 64
  best, best_cost = 1, None
  for s in range(1, max_splits + 1):
    per = -(-slabs // s)
    if s > 1 and (per < min_slabs or (s - 1) * per >= slabs):
      continue
    rounds = -(-tiles * s // 256)
    cost = rounds * (per + 5) + 0.35 * s * tiles / 9
    if best_cost is None or cost < best_cost:
      best, best_cost = s, cost
  return best


def gemm_tn_splitk(dY, X, out, partials, bias_out=None, bias_partials=None, splits=None):
  """out (N, K) fp32 = dY (T, N)^T @ X (T, K), token-major bf16 operands read in
  place (mdlm_gemm_tn_splitk + mdlm_splitk_reduce); bias_out (N) = column sums
  of dY.  `out` may hold fewer than N rows (the leading rows are written)."""
  T, N = dY.shape
  K = X.shape[1]
  assert X.shape[0] == T and dY.dtype == torch.bfloat16 and X.dtype == torch.bfloat16
  assert dY.stride(1) == 1 and X.stride(1) == 1 and out.is_contiguous()
  if splits is None:
    splits = tn_splitk_plan(N, K, T)
  assert partials.numel() >= splits * N * K and partials.dtype == torch.float32
  if bias_out is not None:
    assert bias_partials is not None and bias_partials.numel() >= splits * N
  call('mdlm_gemm_tn_splitk', dY.data_ptr(), dY.stride(0), X.data_ptr(), X.stride(0),
       partials.data_ptr(), _p(bias_partials if bias_out is not None else None),
       int(T), int(N), int(K), int(splits), _stream())
  call('mdlm_splitk_reduce', partials.data_ptr(), out.data_ptr(), out.numel(),
       int(splits), int(N * K), _stream())
  if bias_out is not None:
    call('mdlm_splitk_reduce', bias_partials.data_ptr(), bias_out.data_ptr(),
         bias_out.numel(), int(splits), int(N), _stream())
  return out


def attn_fwd(q, k, vt, out, n_seq, H, L, n_seq_dev=None):
  call('mdlm_attn_fwd', q.data_ptr(), k.data_ptr(), vt.data_ptr(),
       out.data_ptr(), int(n_seq), int(H), int(L), _p(n_seq_dev), _stream())
  return out


def masked_row_map(x, active, n_active, L, mask_index, gather, scatter, n_out,
                   pad=256):
  call('mdlm_masked_row_map', x.data_ptr(), active.data_ptr(),
       n_active.data_ptr(), int(L), int(mask_index), int(pad),
       gather.data_ptr(), scatter.data_ptr(), n_out.data_ptr(), _stream())


def check_token_ids(ids, vocab):
  """Raise on ids outside [0, vocab): the reference's embedding lookup raises /
  device-asserts on them (dit.py:298-299), while the kernels clamp (forward) as
  a memory-safety net only.  One host read of a 2-element tensor: call it at
  API boundaries, not inside the sampling loop."""
  lo, hi = torch.aminmax(ids)
  lo, hi = int(lo), int(hi)
  if lo < 0 or hi >= vocab:
    raise IndexError(f'token id out of range: min {lo}, max {hi}, vocabulary size {vocab} '
                     '(tokenizer / checkpoint mismatch?)')


# tile argument of gemm_bf16 calls that do not pass one.  Host-side (Python)
# only: the tests flip it to run every GEMM of the backbone on the 128x128
# configuration; the library itself holds no tuning state.
DEFAULT_GEMM_TILE = 0


def gemm_set_tile(force_small):
  """Tests / scratch benches: default `tile` of later gemm_bf16 calls (bit 0:
  128x128; bits 1-2 only act in -DMDLM_GEMM_ABLATION builds)."""
  global DEFAULT_GEMM_TILE
  DEFAULT_GEMM_TILE = int(force_small)


def workspace_bytes(op, *args):
  """`mdlm_<op>_workspace_bytes(...)` of include/mdlm_hip.h (host arithmetic)."""
  n = _lib.call_i64(f'mdlm_{op}_workspace_bytes', *[int(a) for a in args])
  if n < 0:
    raise RuntimeError(f'mdlm_{op}_workspace_bytes{args}: bad shape')
  return n


def workspace(op, *args, device=None, dtype=torch.float32):
  """Workspace tensor sized by the library's query."""
  nbytes = workspace_bytes(op, *args)
  return torch.empty(nbytes // torch.empty((), dtype=dtype).element_size(),
                     dtype=dtype, device=device)


# ------------------------------------------------------- training step (f-3)
def attn_fwd_lse(q, k, vt, out, lse, n_seq, H, L):
  call('mdlm_attn_fwd_lse', q.data_ptr(), k.data_ptr(), vt.data_ptr(),
       out.data_ptr(), lse.data_ptr(), int(n_seq), int(H), int(L), _stream())
  return out


def attn_bwd(q, k, vt, o, dout, lse, delta_ws, dq, dk, dv, n_seq, H, L):
  call('mdlm_attn_bwd', q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(),
       dout.data_ptr(), lse.data_ptr(), delta_ws.data_ptr(), dq.data_ptr(),
       dk.data_ptr(), dv.data_ptr(), int(n_seq), int(H), int(L), _stream())


def qkv_grad_pack(dq, dk, dv, cos, sin, dqkv, n_seq, H, L):
  call('mdlm_qkv_grad_pack', dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
       cos.data_ptr(), sin.data_ptr(), dqkv.data_ptr(), int(n_seq), int(H),
       int(L), _stream())
  return dqkv


def ln_modulate_bwd(dh, x, ln_w, scale, dx_inout, partial, rows_per_mod, n_mod,
                    mod_ld, accumulate=True):
  rows, D = x.shape
  call('mdlm_ln_modulate_bwd', dh.data_ptr(), x.data_ptr(), ln_w.data_ptr(),
       scale.data_ptr(), dx_inout.data_ptr(), partial.data_ptr(), int(rows),
       int(D), int(rows_per_mod), int(n_mod), int(mod_ld), int(bool(accumulate)),
       _stream())


def colsum_reduce(partial, out, C, per_out, n_out):
  call('mdlm_colsum_reduce', partial.data_ptr(), out.data_ptr(), int(C),
       int(per_out), int(n_out), _stream())
  return out


def colsum_bf16(x, partial=None, out=None):
  """Column sums (fp32) of a 2-D bf16 matrix, two fixed-order stages."""
  rows, N = x.shape
  nblk = (rows + 127) // 128
  if partial is None:
    partial = torch.empty(nblk, N, dtype=torch.float32, device=x.device)
  if out is None:
    out = torch.empty(N, dtype=torch.float32, device=x.device)
  call('mdlm_colsum_bf16_partial', x.data_ptr(), x.stride(0), partial.data_ptr(),
       int(rows), int(N), _stream())
  return colsum_reduce(partial, out, N, nblk, 1)


def dropout_gate_residual_fwd(y, gate, residual, out, rows_per_mod, n_mod, mod_ld,
                              p, seed, offset):
  rows, D = y.shape
  call('mdlm_dropout_gate_residual_fwd', y.data_ptr(), gate.data_ptr(),
       residual.data_ptr(), out.data_ptr(), int(rows), int(D), int(rows_per_mod),
       int(n_mod), int(mod_ld), float(p), int(seed), int(offset), _stream())
  return out


def dropout_gate_residual_bwd(dout, y, gate, dy, partial, rows_per_mod, n_mod,
                              mod_ld, p, seed, offset):
  rows, D = y.shape
  call('mdlm_dropout_gate_residual_bwd', dout.data_ptr(), y.data_ptr(),
       gate.data_ptr(), dy.data_ptr(), partial.data_ptr(), int(rows), int(D),
       int(rows_per_mod), int(n_mod), int(mod_ld), float(p), int(seed),
       int(offset), _stream())


def gelu_fwd(pre, act):
  call('mdlm_gelu_fwd', pre.data_ptr(), act.data_ptr(), pre.numel(), _stream())
  return act


def gelu_bwd(dact, pre, dpre):
  call('mdlm_gelu_bwd', dact.data_ptr(), pre.data_ptr(), dpre.data_ptr(),
       pre.numel(), _stream())
  return dpre


def transpose_bf16(x, out=None, ld_out=None, colsum_partial=None):
  """(R, C) or (batch, R, C) bf16 -> (C, R) / (batch, C, R); the input may have
  a padded leading dimension (stride(-2) >= C).  colsum_partial: fp32
  (ceil(R/64), C) workspace that receives per-64-row column sums (2-D input)."""
  if x.dim() == 2:
    R, C = x.shape
    batch, bs_in = 1, 0
  else:
    batch, R, C = x.shape
    bs_in = x.stride(0)
  if out is None:
    shape = (C, R) if x.dim() == 2 else (batch, C, R)
    out = torch.empty(shape, dtype=torch.bfloat16, device=x.device)
  ld_o = out.stride(-2) if ld_out is None else ld_out
  bs_out = out.stride(0) if out.dim() == 3 else 0
  call('mdlm_transpose_bf16', x.data_ptr(), x.stride(-2), int(bs_in),
       out.data_ptr(), int(ld_o), int(bs_out), int(R), int(C), int(batch),
       _p(colsum_partial), _stream())
  return out


def embed_grad(dx, idx, dtable, hot_index=-1, hot_partial=None):
  """hot_index: a token id (the mask token) whose rows are summed in a fixed
  order instead of through atomics; its table row is overwritten."""
  rows, D = dx.shape
  if hot_index >= 0 and hot_partial is None:
    hot_partial = torch.empty((rows + 127) // 128, D, dtype=torch.float32, device=dx.device)
  call('mdlm_embed_grad', dx.data_ptr(), idx.data_ptr(), dtable.data_ptr(),
       int(rows), int(D), int(dtable.shape[0]), int(hot_index), _p(hot_partial),
       _stream())


def subs_nll_bwd(logits2d, xt, x0, row_weight, dlogits2d, V, mask_index):
  """logits2d / dlogits2d: (rows, >=V) views with padded leading dimension."""
  rows = logits2d.shape[0]
  call('mdlm_subs_nll_bwd', logits2d.data_ptr(), logits2d.stride(0),
       _DT[logits2d.dtype], xt.data_ptr(), x0.data_ptr(), row_weight.data_ptr(),
       dlogits2d.data_ptr(), dlogits2d.stride(0), int(rows), int(V),
       int(mask_index), _stream())


def grad_sumsq(g, partial, out):
  call('mdlm_grad_sumsq', g.data_ptr(), g.numel(), partial.data_ptr(),
       out.data_ptr(), _stream())
  return out


def adamw_ema(p, g, m, v, ema, w16, lr, beta1, beta2, eps, weight_decay, step,
              ema_one_minus_decay, sumsq=None, max_norm=0.0, grad_scale=1.0):
  call('mdlm_adamw_ema', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(),
       _p(ema), _p(w16), p.numel(), float(lr), float(beta1), float(beta2),
       float(eps), float(weight_decay), int(step), float(ema_one_minus_decay),
       _p(sumsq), float(max_norm), float(grad_scale), _stream())
