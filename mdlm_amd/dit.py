"""Host-side mirror of the reference DiT backbone (models/dit.py) whose forward
runs entirely on the hand-written HIP kernels of libmdlm_hip.so.

Same constructor, attribute and state-dict layout as `models.dit.DIT`
(dit.py:324-370): `vocab_embed.embedding`, `sigma_map.mlp.{0,2}`,
`rotary_emb.inv_freq`, `blocks.i.{norm1,attn_qkv,attn_out,norm2,mlp.0,mlp.2,
adaLN_modulation}`, `output_layer.{norm_final,linear,adaLN_modulation}` -- a
reference checkpoint's `backbone.*` entries load with `load_state_dict`.

dtype flow (dit.py under `autocast(bf16)`): residual stream
fp32, LayerNorm fp32, GEMM/attention operands bf16 with fp32 accumulation,
logits bf16.  The nn.Linear / nn.Parameter objects below only HOLD weights;
their torch forward is never called.
"""
import math

import torch
import torch.nn as nn

from . import ops
from ._lib import EPI_BIAS_GELU, EPI_GATE_RESIDUAL, EPI_QKV_ROTARY, EPI_STORE


def _cfg(config, name):
  m = config['model'] if isinstance(config, dict) else config.model
  return m[name] if isinstance(m, dict) else getattr(m, name)


class LayerNorm(nn.Module):  # dit.py:126-134 (weight only, no bias)
  def __init__(self, dim):
    super().__init__()
    self.weight = nn.Parameter(torch.ones([dim]))
    self.dim = dim


class TimestepEmbedder(nn.Module):  # dit.py:150-189
  def __init__(self, hidden_size, frequency_embedding_size=256):
    super().__init__()
    self.mlp = nn.Sequential(
      nn.Linear(frequency_embedding_size, hidden_size, bias=True),
      nn.SiLU(),
      nn.Linear(hidden_size, hidden_size, bias=True))
    self.frequency_embedding_size = frequency_embedding_size


class Rotary(nn.Module):  # dit.py:81-104
  def __init__(self, dim, base=10_000):
    super().__init__()
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    self.register_buffer('inv_freq', inv_freq)


class DDiTBlock(nn.Module):  # dit.py:214-234
  def __init__(self, dim, n_heads, cond_dim, mlp_ratio=4, dropout=0.1):
    super().__init__()
    self.n_heads = n_heads
    self.norm1 = LayerNorm(dim)
    self.attn_qkv = nn.Linear(dim, 3 * dim, bias=False)
    self.attn_out = nn.Linear(dim, dim, bias=False)
    self.norm2 = LayerNorm(dim)
    self.mlp = nn.Sequential(
      nn.Linear(dim, mlp_ratio * dim, bias=True),
      nn.GELU(approximate='tanh'),
      nn.Linear(mlp_ratio * dim, dim, bias=True))
    self.dropout = dropout
    self.adaLN_modulation = nn.Linear(cond_dim, 6 * dim, bias=True)
    self.adaLN_modulation.weight.data.zero_()
    self.adaLN_modulation.bias.data.zero_()


class EmbeddingLayer(nn.Module):  # dit.py:292-299
  def __init__(self, dim, vocab_dim):
    super().__init__()
    self.embedding = nn.Parameter(torch.empty((vocab_dim, dim)))
    torch.nn.init.kaiming_uniform_(self.embedding, a=math.sqrt(5))


class DDitFinalLayer(nn.Module):  # dit.py:302-314
  def __init__(self, hidden_size, out_channels, cond_dim):
    super().__init__()
    self.norm_final = LayerNorm(hidden_size)
    self.linear = nn.Linear(hidden_size, out_channels)
    self.linear.weight.data.zero_()
    self.linear.bias.data.zero_()
    self.adaLN_modulation = nn.Linear(cond_dim, 2 * hidden_size, bias=True)
    self.adaLN_modulation.weight.data.zero_()
    self.adaLN_modulation.bias.data.zero_()


class DIT(nn.Module):
  """`backbone(indices, sigma) -> logits (B, L, V)` on MI355X kernels."""

  def __init__(self, config, vocab_size: int):
    super().__init__()
    self.config = config
    self.vocab_size = vocab_size
    self.hidden = _cfg(config, 'hidden_size')
    self.n_heads = _cfg(config, 'n_heads')
    self.cond_dim = _cfg(config, 'cond_dim')
    self.n_blocks = _cfg(config, 'n_blocks')
    if self.hidden // self.n_heads != 64 or self.hidden % self.n_heads:
      raise ValueError('mdlm_amd DIT kernels are written for head_dim == 64 '
                       f'(got hidden {self.hidden}, heads {self.n_heads})')
    self.vocab_embed = EmbeddingLayer(self.hidden, vocab_size)
    self.sigma_map = TimestepEmbedder(self.cond_dim)
    self.rotary_emb = Rotary(self.hidden // self.n_heads)
    self.blocks = nn.ModuleList([
      DDiTBlock(self.hidden, self.n_heads, self.cond_dim,
                dropout=_cfg(config, 'dropout'))
      for _ in range(self.n_blocks)])
    self.output_layer = DDitFinalLayer(self.hidden, vocab_size, self.cond_dim)
    self.scale_by_sigma = _cfg(config, 'scale_by_sigma')
    self._prep = None   # device-side kernel operands (bf16 weights, tables)
    self.profile_events = None   # bench.py profiling pass only (see _timed)
    self.profile_sample = False
    self._ws = None     # activation workspace
    # block GEMM launches of at most this many 256x256 tiles use 128x128 tiles (run)
    self.small_tile_limit = 128
    self._ws_groups = {}

  # ------------------------------------------------------------ preparation
  def refresh(self):
    """Drop cached kernel operands (call after changing weights)."""
    self._prep = None

  def load_state_dict(self, *a, **k):
    r = super().load_state_dict(*a, **k)
    self.refresh()
    return r

  def _apply(self, fn, *a, **k):
    r = super()._apply(fn, *a, **k)
    self._prep = None
    self._ws = None
    self._ws_groups = {}
    return r

  def _prepare(self, L):
    dev = self.vocab_embed.embedding.device
    if dev.type != 'cuda':
      raise RuntimeError('mdlm_amd.DIT runs only on an MI355X (cuda device); '
                         'there is no CPU fallback')
    if self._prep is not None and self._prep['L'] == L:
      return self._prep
    bf = torch.bfloat16

    def w16(lin):
      return lin.weight.detach().to(bf).contiguous()

    def f32(t):
      return t.detach().float().contiguous()
    p = {'L': L}
    p['embed'] = f32(self.vocab_embed.embedding)
    p['blocks'] = []
    for blk in self.blocks:
      p['blocks'].append(dict(
        n1=f32(blk.norm1.weight), n2=f32(blk.norm2.weight),
        qkv=w16(blk.attn_qkv), out=w16(blk.attn_out),
        fc1=w16(blk.mlp[0]), b1=f32(blk.mlp[0].bias),
        fc2=w16(blk.mlp[2]), b2=f32(blk.mlp[2].bias),
        ada_w=f32(blk.adaLN_modulation.weight),
        ada_b=f32(blk.adaLN_modulation.bias)))
    ol = self.output_layer
    p['final'] = dict(n=f32(ol.norm_final.weight), w=w16(ol.linear),
                      b=f32(ol.linear.bias),
                      ada_w=f32(ol.adaLN_modulation.weight),
                      ada_b=f32(ol.adaLN_modulation.bias))
    sm = self.sigma_map.mlp
    p['sm'] = (f32(sm[0].weight), f32(sm[0].bias), f32(sm[2].weight),
               f32(sm[2].bias))
    # rotary tables (dit.py:90-104), cast to bf16 like dit.py:264
    t = torch.arange(L, device=dev).float()
    freqs = torch.einsum('i,j->ij', t, self.rotary_emb.inv_freq.float())
    p['cos'] = freqs.cos().to(bf).float().contiguous()
    p['sin'] = freqs.sin().to(bf).float().contiguous()
    p['cond0'] = None
    self._prep = p
    return p

  def _workspace(self, M, key=0):
    """Activation workspace; `key` names an independent copy (one per stream
    group of the sampling loop, whose forwards run concurrently)."""
    if key:
      if self._ws_groups.get(key) is None or self._ws_groups[key]['M'] < M:
        keep, self._ws = self._ws, None
        self._ws_groups[key] = self._workspace(M)
        self._ws = keep
      return self._ws_groups[key]
    if self._ws is not None and self._ws['M'] >= M:
      return self._ws
    dev = self.vocab_embed.embedding.device
    D = self.hidden
    bf = torch.bfloat16
    self._ws = dict(
      M=M,
      x=torch.empty(M, D, dtype=torch.float32, device=dev),
      h=torch.empty(M, D, dtype=bf, device=dev),
      qkv=torch.empty(3 * M * D, dtype=bf, device=dev),
      ao=torch.empty(M, D, dtype=bf, device=dev),
      mh=torch.empty(M, 4 * D, dtype=bf, device=dev),
      # dense copies of the masked rows for the tail of the last block
      xm=torch.empty(M, D, dtype=torch.float32, device=dev),
      aom=torch.empty(M, D, dtype=bf, device=dev))
    return self._ws

  # ------------------------------------------------------------ conditioning
  def _conditioning(self, p, sigma):
    """c = silu(sigma_map(sigma)) (dit.py:361) and every adaLN vector
    (dit.py:249-250, 318).  Returns a list of (n_mod, 6D) per block + (n_mod,
    2D) for the final layer."""
    te = ops.timestep_embedding(sigma.float().contiguous(),
                                self.sigma_map.frequency_embedding_size)
    w0, b0, w2, b2 = p['sm']
    hcond = ops.linear_f32(te, w0, b0, act_out=1)
    c = ops.linear_f32(hcond, w2, b2, act_out=1)
    mods = [ops.linear_f32(c, bp['ada_w'], bp['ada_b']) for bp in p['blocks']]
    mods.append(ops.linear_f32(c, p['final']['ada_w'], p['final']['ada_b']))
    return mods

  def conditioning(self, p, sigma, sigma_is_zero):
    if sigma_is_zero:
      # time_conditioning=False: sigma is zeroed (diffusion.py:307-308), so
      # the conditioning is one constant vector per model.
      if p['cond0'] is None:
        z = torch.zeros(1, device=p['embed'].device)
        p['cond0'] = self._conditioning(p, z)
      return p['cond0'], 1
    return self._conditioning(p, sigma), sigma.shape[0]

  # ------------------------------------------------------------------ forward
  def _timed(self, tag, rows, work_per_row, fn):
    """Profiling hook (bench.py, outside its timed region): HIP events on the
    launch stream around one kernel launch.  `profile_events` is a dict tag ->
    list of (start, end, rows, work/row) -- flops for the MFMA kernels, bytes
    for the HBM-bound ones; `rows` may be a device counter (cloned)."""
    ev = self.profile_events
    if ev is None or not self.profile_sample:
      return fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    ev.setdefault(tag, []).append((e0, e1, rows, work_per_row))

  def run(self, indices, sigma, logits_out, sigma_is_zero=False, active=None,
          n_active_dev=None, n_rows_dev=None, n_max=None, row_map=None,
          hidden_out=None, lse_out=None, lse_skip_col=-1, ws_key=0,
          concurrent=False):
    """Run the backbone.  `indices` (B, L) int64.  `logits_out`: (B, L, V)
    bf16/fp32 view (padded leading dimension allowed).  If `active` (int32
    device list) is given, only the listed samples are evaluated (compacted
    into rows 0..n*L of the workspace) and their logits are scattered to
    logits_out[active[i]]; n_active_dev / n_rows_dev are device counters and
    n_max the launch bound.  `row_map = (gather, scatter, n_dev)` from
    `ops.masked_row_map` restricts the final LayerNorm + vocabulary projection
    to the masked rows (the only rows whose logits are ever read).
    `hidden_out` (B*L, D) fp32: stop after the last block and write the
    residual stream of the evaluated samples to their rows of this per-sample
    cache instead of projecting to the vocabulary (`project_rows` does that
    later for the rows that need it).  `lse_out` (B*L) fp32 with a row map:
    the SUBS row lse (column `lse_skip_col` = the mask token excluded) of the
    projected rows comes out of the vocabulary GEMM's epilogue and is written to
    lse_out[scatter[row]] -- no pass over the logits.  `ws_key`: which workspace
    copy to use; `concurrent`: another forward runs on another stream at the same
    time (the GEMMs then keep whole tiles in a short last round: a half tile
    finishes sooner but costs 1.5x the CU time, which the other stream could use)."""
    B, L = indices.shape
    p = self._prepare(L)
    D, H = self.hidden, self.n_heads
    n_seq = B if n_max is None else n_max
    M = n_seq * L
    ws = self._workspace(M, ws_key)
    x, h, qkv, ao, mh = ws['x'][:M], ws['h'][:M], ws['qkv'], ws['ao'][:M], ws['mh'][:M]
    mods, n_mod = self.conditioning(p, sigma, sigma_is_zero)
    if n_mod != 1 and active is not None:
      raise RuntimeError('active-sample compaction needs batch-invariant '
                         'conditioning (time_conditioning=False)')
    ld6, ld2 = 6 * D, 2 * D
    prof = self.profile_events is not None and self.profile_sample
    rows_t = None
    if prof:  # one device copy of the live row count per forward
      rows_t = n_rows_dev.clone() if n_rows_dev is not None else M
    T = self._timed
    V = self.vocab_size
    LN_B, GEMM = 6.0 * D, 2.0 * D   # bytes per LN row; flops per row per output column
    T('embed', rows_t, 4.0 * D + 8, lambda: ops.embed_rows(
      p['embed'], indices, x, sample_map=active, L=L, n_rows_dev=n_rows_dev))
    q_ = qkv[:M * D]
    k_ = qkv[M * D:2 * M * D]
    vt_ = qkv[2 * M * D:3 * M * D]
    # With a row map (only the masked rows' logits are read) the last block
    # stops being full-width after its attention: out-projection, MLP and the
    # final layer are row-wise, so they run on the gathered masked rows only.
    tail_rows = (row_map is not None and n_mod == 1 and hidden_out is None
                 and getattr(self, 'tail_rows', True))
    # SUBS row lse of the projected rows out of the vocabulary GEMM's epilogue
    lsep = None
    if lse_out is not None:
      if row_map is None or M % 256:
        raise ValueError('lse_out needs a masked-row map and a row count % 256 == 0')
      if ws.get('lsep') is None or ws['lsep_key'] != (M, V):
        ws['lsep'] = torch.empty(ops.workspace_bytes('gemm_lse_partials', M, V) // 4,
                                 dtype=torch.float32, device=x.device)
        ws['lsep_key'] = (M, V)
      lsep = ws['lsep']
    n_blocks = len(p['blocks'])

    def gemm_tile(N, whole=False):
      """`tile` of a block GEMM with N output columns.  A launch of only a few samples
      (up to `small_tile_limit` 256x256 tiles, half of the CUs) runs on the 128x128
      configuration: four times the workgroups, 0.5-0.7x the time at 1-8 samples
      (scratch/small_tiles.py), and the same K order per output element, so the results
      are bit-identical whatever the launch size (tests/test_gpu_dit.py)."""
      t = ops.DEFAULT_GEMM_TILE
      if M % 256 == 0 and (M // 256) * ((N + 255) // 256) <= self.small_tile_limit:
        t |= 1
      return (t | 16) if (concurrent and whole) else t
    for bi, (bp, mod) in enumerate(zip(p['blocks'], mods)):
      # chunk order dit.py:249-250: shift_msa, scale_msa, gate_msa, shift_mlp,
      # scale_mlp, gate_mlp
      T('ln_modulate', rows_t, LN_B, lambda: ops.ln_modulate(
        x, bp['n1'], mod[:, 0:D], mod[:, D:2 * D], h, L, n_mod, ld6, n_rows_dev))
      T('gemm_qkv', rows_t, GEMM * 3 * D, lambda: ops.gemm_bf16(
        h, bp['qkv'], qkv, EPI_QKV_ROTARY, rows_per_mod=L, rot_cos=p['cos'],
        rot_sin=p['sin'], n_heads=H, m_rows_dev=n_rows_dev, ldc=0, M=M, tile=gemm_tile(3 * D, True)))
      T('attention', rows_t, 4.0 * L * D, lambda: ops.attn_fwd(
        q_, k_, vt_, ao, n_seq, H, L, n_seq_dev=n_active_dev))
      if tail_rows and bi == n_blocks - 1:
        gather, scatter, live = row_map
        xm, aom = ws['xm'][:M], ws['aom'][:M]
        lt = live.clone() if prof else None
        T('gather_rows', lt, 12.0 * D, lambda: ops.gather_rows(
          x, xm, ao, aom, gather, n_rows_dev=live, rows=M))
        T('gemm_out', lt, GEMM * D, lambda: ops.gemm_bf16(
          aom, bp['out'], xm, EPI_GATE_RESIDUAL, gate=mod[:, 2 * D:3 * D],
          residual=xm, rows_per_mod=L, n_mod=1, mod_ld=ld6, m_rows_dev=live, M=M, tile=gemm_tile(D, True)))
        T('ln_modulate', lt, LN_B, lambda: ops.ln_modulate(
          xm, bp['n2'], mod[:, 3 * D:4 * D], mod[:, 4 * D:5 * D], h, L, 1, ld6, live))
        T('gemm_fc1', lt, GEMM * 4 * D, lambda: ops.gemm_bf16(
          h, bp['fc1'], mh, EPI_BIAS_GELU, bias=bp['b1'], m_rows_dev=live, M=M, tile=gemm_tile(4 * D)))
        T('gemm_fc2', lt, GEMM * 4 * D, lambda: ops.gemm_bf16(
          mh, bp['fc2'], xm, EPI_GATE_RESIDUAL, bias=bp['b2'],
          gate=mod[:, 5 * D:6 * D], residual=xm, rows_per_mod=L, n_mod=1,
          mod_ld=ld6, m_rows_dev=live, M=M, tile=gemm_tile(D, True)))
        fm = mods[-1]
        T('ln_modulate', lt, LN_B, lambda: ops.ln_modulate(
          xm, p['final']['n'], fm[:, 0:D], fm[:, D:2 * D], h, L, 1, ld2, live))
        T('gemm_vocab', lt, GEMM * V, lambda: ops.gemm_bf16(
          h, p['final']['w'], logits_out, EPI_STORE, bias=p['final']['b'],
          m_rows_dev=live, out_row_map=scatter, ldc=logits_out.stride(-2), M=M,
          lse_partials=lsep, lse_skip_col=lse_skip_col))
        if lsep is not None:
          T('lse_combine', lt, 8.0 * ((V + 63) // 64), lambda: ops.lse_combine(
            lsep, M, V, lse_out, out_row_map=scatter, n_rows_dev=live))
        return logits_out
      T('gemm_out', rows_t, GEMM * D, lambda: ops.gemm_bf16(
        ao, bp['out'], x, EPI_GATE_RESIDUAL, gate=mod[:, 2 * D:3 * D],
        residual=x, rows_per_mod=L, n_mod=n_mod, mod_ld=ld6,
        m_rows_dev=n_rows_dev, M=M, tile=gemm_tile(D, True)))
      T('ln_modulate', rows_t, LN_B, lambda: ops.ln_modulate(
        x, bp['n2'], mod[:, 3 * D:4 * D], mod[:, 4 * D:5 * D], h, L, n_mod, ld6,
        n_rows_dev))
      T('gemm_fc1', rows_t, GEMM * 4 * D, lambda: ops.gemm_bf16(
        h, bp['fc1'], mh, EPI_BIAS_GELU, bias=bp['b1'], m_rows_dev=n_rows_dev, M=M, tile=gemm_tile(4 * D)))
      T('gemm_fc2', rows_t, GEMM * 4 * D, lambda: ops.gemm_bf16(
        mh, bp['fc2'], x, EPI_GATE_RESIDUAL, bias=bp['b2'],
        gate=mod[:, 5 * D:6 * D], residual=x, rows_per_mod=L,
        n_mod=n_mod, mod_ld=ld6, m_rows_dev=n_rows_dev, M=M, tile=gemm_tile(D, True)))
    if hidden_out is not None:
      ops.scatter_rows_f32(x, hidden_out, active, L, n_rows_dev=n_rows_dev)
      return None
    fm = mods[-1]
    assert logits_out.shape[-1] == self.vocab_size
    if row_map is not None:
      gather, scatter, live = row_map
      lt = live.clone() if prof else None
      T('ln_modulate', lt, LN_B, lambda: ops.ln_modulate(
        x, p['final']['n'], fm[:, 0:D], fm[:, D:2 * D], h, L, n_mod, ld2,
        n_rows_dev=live, row_gather=gather, rows=M))
      T('gemm_vocab', lt, GEMM * V, lambda: ops.gemm_bf16(
        h, p['final']['w'], logits_out, EPI_STORE, bias=p['final']['b'],
        m_rows_dev=live, out_row_map=scatter, ldc=logits_out.stride(-2), M=M,
        lse_partials=lsep, lse_skip_col=lse_skip_col))
      if lsep is not None:
        T('lse_combine', lt, 8.0 * ((V + 63) // 64), lambda: ops.lse_combine(
          lsep, M, V, lse_out, out_row_map=scatter, n_rows_dev=live))
    else:
      T('ln_modulate', rows_t, LN_B, lambda: ops.ln_modulate(
        x, p['final']['n'], fm[:, 0:D], fm[:, D:2 * D], h, L, n_mod, ld2, n_rows_dev))
      T('gemm_vocab', rows_t, GEMM * V, lambda: ops.gemm_bf16(
        h, p['final']['w'], logits_out, EPI_STORE, bias=p['final']['b'],
        rows_per_mod=L, m_rows_dev=n_rows_dev, out_sample_map=active,
        ldc=logits_out.stride(-2), M=M))
    return logits_out

  def project_rows(self, hidden, gather, n_rows_dev, logits_out, L,
                   out_row_map=None):
    """Output layer (dit.py:317-321, sigma = 0 conditioning) for the listed rows
    of a per-sample hidden-state cache: LN + modulate gathers `hidden[gather[i]]`
    into dense rows, the vocabulary GEMM writes logits row i to `logits_out[i]`
    (or to row `out_row_map[i]`).  n_rows_dev: device count, a multiple of 256."""
    p = self._prepare(L)
    D, V = self.hidden, self.vocab_size
    mods, n_mod = self.conditioning(p, None, True)
    fm = mods[-1]
    M = hidden.shape[0]
    h = self._workspace(M)['h'][:M]
    ops.ln_modulate(hidden, p['final']['n'], fm[:, 0:D], fm[:, D:2 * D], h, L, n_mod,
                    2 * D, n_rows_dev=n_rows_dev, row_gather=gather, rows=M)
    ops.gemm_bf16(h, p['final']['w'], logits_out, EPI_STORE, bias=p['final']['b'],
                  m_rows_dev=n_rows_dev, out_row_map=out_row_map,
                  ldc=logits_out.stride(-2), M=M)
    return logits_out

  def forward(self, indices, sigma, out=None, out_dtype=torch.bfloat16):
    """dit.py:359-370: logits (B, L, V) (bf16 like the reference under
    autocast).  The result is a view of a row-padded buffer."""
    if not indices.is_cuda:
      raise RuntimeError('mdlm_amd.DIT.forward: expected CUDA/ROCm tensors; '
                         'there is no CPU fallback')
    B, L = indices.shape
    if out is None:
      out = ops.alloc_logits(B, L, self.vocab_size, out_dtype, indices.device)
    return self.run(indices.contiguous(), sigma, out)
