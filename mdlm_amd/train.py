"""Training step of the DiT/SUBS model on MI355X kernels.

Mirrors the reference's `training_step` (diffusion.py:390-397) -> `_loss`
(896-916) with the backbone in train mode (dropout 0.1 inside
bias_dropout_add_scale, dit.py:20-34), Lightning's `gradient_clip_val: 1.0`
(configs/config.yaml:69), `torch.optim.AdamW` (diffusion.py:449-459), the
constant-with-warmup schedule (configs/lr_scheduler/constant_warmup.yaml) and
the EMA update after every optimizer step (diffusion.py:254-259,
models/ema.py:29-49).  Where the reference relies on torch autograd, every
backward here is a hand-written HIP kernel (mdlm_amd/csrc/train_ops.hip,
attn_bwd.hip) or mdlm_gemm_bf16 on transposed operands.  The only autograd
left is the conditioning network (sigma_map + adaLN Linear on a (n_mod, 128)
vector: a few kFLOP), which receives d(mod) from the kernels.

Memory layout: every parameter is a view of ONE flat fp32 buffer; gradients,
Adam moments, EMA shadow and the bf16 GEMM copies are flat buffers with the
same offsets, so clip + AdamW + EMA + bf16 refresh is one launch over 170 M
elements and a DDP bucket is a contiguous slice.  Data parallelism: one
process per GPU; the gradient slice of a block is all-reduced (RCCL over xGMI,
sum; the 1/world scale is folded into the optimizer kernel) on a side stream as
soon as that block's backward has finished, overlapping the remaining blocks.
"""
import math

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import dist as mdist
from . import ops
from ._lib import EPI_QKV_ROTARY, EPI_STORE
from .shadow import ALIGN as _ALIGN   # elements: every parameter view starts 256-byte aligned
from .shadow import flat_layout


def constant_warmup_lr(base_lr, step, num_warmup_steps):
  """transformers.get_constant_schedule_with_warmup: lr of optimizer step
  number `step` (0-based)."""
  if step < num_warmup_steps:
    return base_lr * float(step) / float(max(1.0, num_warmup_steps))
  return base_lr


class GradBuckets:
  """Contiguous slices of the flat gradient buffer, all-reduced asynchronously
  (sum) in the order their backward finishes.  Works on any device/backend
  (RCCL on the GPU, gloo in the CPU tests)."""

  def __init__(self, flat, slices, group=None):
    self.flat, self.slices, self.group = flat, slices, group
    self.work = []
    self.hold = False    # no_sync: a micro-batch that is not the last of its step
    self.carry = None    # gradients of the earlier micro-batches of this step
    self.stream = torch.cuda.Stream() if flat.is_cuda else None

  @property
  def enabled(self):
    return dist.is_initialized() and dist.get_world_size(self.group) > 1

  def launch(self, i):
    a, b = self.slices[i]
    if b <= a or self.hold:
      return
    if self.carry is not None:
      self.flat[a:b] += self.carry[a:b]
    if not self.enabled:
      return
    if self.stream is not None and dist.get_backend(self.group) != 'nccl':
      # rehearsal of N ranks on one card over gloo: staged through host memory,
      # synchronously (gloo's own device path stalls on large slices when both
      # ranks share a GPU); the RCCL path below is the product path
      host = self.flat[a:b].cpu()
      dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
      self.flat[a:b].copy_(host)
    elif self.stream is not None:
      self.stream.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(self.stream):
        self.work.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM,
                                         group=self.group, async_op=True))
    else:
      self.work.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM,
                                       group=self.group, async_op=True))

  def wait_all(self):
    for w in self.work:
      w.wait()
    self.work = []
    if self.stream is not None:
      torch.cuda.current_stream().wait_stream(self.stream)


class Trainer:
  def __init__(self, model, lr=None, betas=None, eps=None, weight_decay=None,
               max_norm=1.0, num_warmup_steps=2500, dropout=None, group=None):
    cfg = model.config
    bb = model.backbone
    dev = bb.vocab_embed.embedding.device
    if dev.type != 'cuda':
      raise RuntimeError('mdlm_amd.Trainer needs the model on a ROCm device; '
                         'there is no CPU fallback')
    if model.T > 0:
      # forward_backward uses the continuous-time weight dsigma / expm1(sigma)
      # (diffusion.py:893-894); the discrete-time branch (849-853, 329-358) is
      # evaluated by Diffusion._loss only
      raise NotImplementedError('Trainer: T > 0 (discrete-time loss) is not implemented; '
                                'training would optimise a different loss from validation')
    self.model, self.bb, self.dev = model, bb, dev
    self.lr = cfg.optim.lr if lr is None else lr
    self.betas = (cfg.optim.beta1, cfg.optim.beta2) if betas is None else betas
    self.eps = cfg.optim.eps if eps is None else eps
    self.weight_decay = cfg.optim.weight_decay if weight_decay is None else weight_decay
    self.max_norm = max_norm
    self.num_warmup_steps = num_warmup_steps
    self.p_drop = float(bb.blocks[0].dropout if dropout is None else dropout)
    self.ema_decay = float(cfg.training.ema)
    # Two counters, as in the reference: `step_count` is the optimizer's (Adam bias
    # correction, lr warm-up; Lightning's global_step), `ema_updates` is
    # models/ema.py's num_updates (decay warm-up).  They differ after a restart
    # from a weights + EMA checkpoint without optimizer state.
    self.step_count = 0
    self.ema_updates = 0
    self.group = group
    self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    # ---- flat buffers ------------------------------------------------------
    named = list(bb.named_parameters())
    self.names = [n for n, _ in named]
    table, off = flat_layout([p.shape for _, p in named])
    self.offsets = {n: e for (n, _), e in zip(named, table)}
    self.n_flat = off
    z = lambda dt=torch.float32: torch.zeros(off, dtype=dt, device=dev)
    self.P, self.G, self.M1, self.M2 = z(), z(), z(), z()
    self.W16 = z(torch.bfloat16)
    for n, p in named:
      v = self._view(self.P, n)
      v.copy_(p.data)
      p.data = v
      p.grad = self._view(self.G, n)
    self.W16.copy_(self.P)
    self.E = None
    if self.ema_decay > 0:
      ema = getattr(model, 'ema', None)
      if ema is not None and ema.n_flat == self.n_flat:
        # the model's shadow weights have this layout already: their buffer IS the
        # optimizer kernel's EMA operand.  An EMA restored by on_load_checkpoint
        # (diffusion.py:168-170) is kept and its update count continues the decay
        # warm-up of models/ema.py:40-44; a constructor-time one restarts from the
        # current weights.
        self.E = ema.to(dev).flat
        if getattr(model, '_ema_restored', False):
          self.ema_updates = int(ema.num_updates)
        else:
          self.E.copy_(self.P)
          ema.num_updates = 0
      else:
        self.E = self.P.clone()
    bb.refresh()
    # DDP buckets in the order the backward completes them: output layer,
    # blocks L-1..0, then everything before block 0 (embedding, sigma_map)
    def span(prefix):
      ks = [n for n in self.names if n.startswith(prefix)]
      a = self.offsets[ks[0]][0]
      o, n_, _ = self.offsets[ks[-1]]
      return a, (o + n_ + _ALIGN - 1) // _ALIGN * _ALIGN
    self.bucket_ids = {}
    slices = [span('output_layer.')]
    for i in reversed(range(bb.n_blocks)):
      self.bucket_ids[i] = len(slices)
      slices.append(span(f'blocks.{i}.'))
    slices.append((0, span('blocks.0.')[0]))
    self.buckets = GradBuckets(self.G, slices, group)
    self._ws = None
    self.A, self.micro = None, 0   # accumulation carry (allocated on first use)
    self.last_lr = None
    self._sumsq = torch.zeros(1, device=dev)
    self._sumsq_part = ops.workspace('grad_sumsq', device=dev)
    self.last_grad_norm = None

  # ---------------------------------------------------------------- helpers
  def _view(self, flat, name):
    o, n, shape = self.offsets[name]
    return flat[o:o + n].view(shape)

  def grads(self):
    return {n: self._view(self.G, n) for n in self.names}

  def ema_state_dict(self):
    """models/ema.py:89-92 layout."""
    return dict(decay=self.ema_decay, num_updates=self.ema_updates,
                shadow_params=[self._view(self.E, n) for n in self.names])

  # -------------------------------------------------------------- checkpoints
  def state_dict(self):
    """Everything a resumed run needs beyond the weights: Adam moments, EMA
    shadow, optimizer-step count and the Philox stream position (the reference
    gets these from Lightning's checkpoint: optimizer_states, loops, and the
    `ema` entry written by on_save_checkpoint, diffusion.py:172-181)."""
    # the Philox key is stored rank-free: every rank of a resumed DDP run loads
    # rank 0's file and must get back ITS OWN stream (seed + rank), not rank 0's
    rank = mdist.env_rank()[0]
    return dict(P=self.P, M1=self.M1, M2=self.M2, E=self.E, step_count=self.step_count,
                ema_updates=self.ema_updates,
                rng_seed=self.model._rng_seed - rank, rng_offset=self.model._rng_offset,
                n_flat=self.n_flat)

  def load_state_dict(self, sd):
    if int(sd['n_flat']) != self.n_flat:
      raise ValueError(f"trainer state holds {sd['n_flat']} flat elements, model has {self.n_flat}")
    self.P.copy_(sd['P']); self.M1.copy_(sd['M1']); self.M2.copy_(sd['M2'])
    if self.E is not None and sd.get('E') is not None:
      self.E.copy_(sd['E'])
    self.W16.copy_(self.P)
    self.step_count = int(sd['step_count'])
    self.ema_updates = int(sd.get('ema_updates', self.step_count))
    self.model.seed_rng(mdist.rank_seed(int(sd['rng_seed']), mdist.env_rank()[0]),
                        int(sd['rng_offset']))
    if getattr(self.model, 'ema', None) is not None:
      self.model.ema.num_updates = self.ema_updates
    self.bb.refresh()

  def save_checkpoint(self, path):
    """Reference (Lightning) layout: `state_dict` with `backbone.*` keys, `ema`
    as models/ema.py:89-92, `global_step`; plus `trainer_state` (this build's
    flat optimizer state).  Tensors and plain containers only, so the file
    loads with torch.load(weights_only=True)."""
    import os
    sd = {'backbone.' + k: v.detach().clone() for k, v in self.bb.state_dict().items()}
    ckpt = {'state_dict': sd, 'global_step': self.step_count,
            'trainer_state': {k: (v.detach().clone() if torch.is_tensor(v) else v)
                              for k, v in self.state_dict().items()}}
    if self.E is not None:
      ckpt['ema'] = dict(decay=self.ema_decay, num_updates=self.ema_updates,
                         shadow_params=[self._view(self.E, n).detach().clone() for n in self.names])
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + '.tmp'
    torch.save(ckpt, tmp)
    os.replace(tmp, path)

  def load_checkpoint(self, path):
    ckpt = torch.load(path, map_location=self.dev, weights_only=True)
    if 'trainer_state' in ckpt:
      self.load_state_dict(ckpt['trainer_state'])
    else:
      # a reference checkpoint: weights (+ EMA) only.  There are no Adam moments
      # in it, so the optimizer starts fresh -- step count, bias correction and lr
      # warm-up at 0 -- and only the EMA's own update count carries over.
      bbsd = {k[len('backbone.'):]: v for k, v in ckpt['state_dict'].items()
              if k.startswith('backbone.')}
      for n in self.names:
        self._view(self.P, n).copy_(bbsd[n])
      self.W16.copy_(self.P)
      if self.E is not None:
        if 'ema' in ckpt:
          for n, sp in zip(self.names, ckpt['ema']['shadow_params']):
            self._view(self.E, n).copy_(sp)
          self.ema_updates = int(ckpt['ema'].get('num_updates', 0) or 0)
        else:
          self.E.copy_(self.P)
          self.ema_updates = 0
        if getattr(self.model, 'ema', None) is not None:
          self.model.ema.num_updates = self.ema_updates
      self.M1.zero_(); self.M2.zero_()
      self.step_count = 0
      self.bb.refresh()
      return 0
    return ckpt.get('global_step', self.step_count)

  def _workspace(self, B, L):
    key = (B, L)
    if self._ws is not None and self._ws['key'] == key:
      return self._ws
    bb, dev = self.bb, self.dev
    D, H, V = bb.hidden, bb.n_heads, bb.vocab_size
    M, Vp = B * L, (V + 255) // 256 * 256   # 256-row tiles for the dW_vocab GEMM
    bf = torch.bfloat16
    e = lambda *s, dt=bf: torch.empty(*s, dtype=dt, device=dev)
    blocks = []
    for _ in range(bb.n_blocks):
      blocks.append(dict(
        x_in=e(M, D, dt=torch.float32), h1=e(M, D), qkv=e(3 * M * D), ao=e(M, D),
        lse=e(B * H * L, dt=torch.float32), y1=e(M, D), x_mid=e(M, D, dt=torch.float32),
        h2=e(M, D), pre=e(M, 4 * D), act=e(M, 4 * D), y2=e(M, D), drop=[0, 0]))
    inv_freq = bb.rotary_emb.inv_freq.float()
    fr = torch.einsum('i,j->ij', torch.arange(L, device=dev).float(), inv_freq)
    nblk = M // 32
    Mp = (M + 255) // 256 * 256   # masked-row lists are padded to 256
    # split-K partial sums of the weight-gradient GEMMs (largest plan)
    shapes = ((3 * D, D), (D, D), (4 * D, D), (D, 4 * D), (Vp, D))
    n_part = max(max(ops.splitk_plan(n, k, M), ops.tn_splitk_plan(n, k, Mp) if n % 256 == 0 and k % 256 == 0 else 1,
                     ops.tn_splitk_plan(n, k, Mp // 2) if n % 256 == 0 and k % 256 == 0 else 1) * n * k
                 for n, k in shapes)
    self._ws = dict(
      skp=e(n_part, dt=torch.float32), skb=e(64 * max(Vp, 4 * D), dt=torch.float32),
      key=key, blocks=blocks, x_last=e(M, D, dt=torch.float32), hf=e(Mp, D),
      logits=torch.zeros(Mp, Vp, dtype=bf, device=dev), dlogits=e(Mp, Vp),
      samples=torch.arange(B, dtype=torch.int32, device=dev),
      n_samples=torch.tensor([B], dtype=torch.int32, device=dev),
      gather=torch.zeros(ops.workspace_bytes('masked_row_map', Mp, 256) // 4,
                         dtype=torch.int32, device=dev),
      scatter=torch.zeros(ops.workspace_bytes('masked_row_map', Mp, 256) // 4,
                          dtype=torch.int32, device=dev),
      n_out=torch.zeros(1, dtype=torch.int32, device=dev),
      hot=ops.workspace('embed_grad', M, D, device=dev).view(-1, D),
      cos=fr.cos().to(bf).float().contiguous(), sin=fr.sin().to(bf).float().contiguous(),
      dx=e(M, D, dt=torch.float32), d_md=e(M, D), d_m4=e(M, 4 * D), d_m4b=e(M, 4 * D),
      d_qkv=e(M, 3 * D), dq=e(M * D), dk=e(M * D), dv=e(M * D),
      delta=ops.workspace('attn_bwd', B, H, L, device=dev),
      tA=e(max(Vp, 4 * D) * Mp), tB=e(4 * D * Mp), wT=e(max(Vp * D, 4 * D * D)),
      wvT=torch.zeros(D, Vp, dtype=bf, device=dev),
      part3=ops.workspace('ln_modulate_bwd', M, D, device=dev).view(nblk, 3, D),
      part1=ops.workspace('dropout_gate_residual_bwd', M, D, device=dev).view(nblk, D),
      bias_v=e(Vp, dt=torch.float32),
      partc=e((Mp + 63) // 64, max(Vp, 4 * D), dt=torch.float32))
    return self._ws

  def _w16(self, name):
    return self._view(self.W16, name)

  def _p32(self, name):
    return self._view(self.P, name)

  def _g(self, name):
    return self._view(self.G, name)

  def _conditioning(self, n_mod, sigma):
    """c = silu(sigma_map(timestep_embedding(sigma))) and the adaLN vectors
    (dit.py:361, 249-250, 318) in fp32 torch autograd (tiny)."""
    bb = self.bb
    te = ops.timestep_embedding(sigma.float().contiguous(),
                                bb.sigma_map.frequency_embedding_size)
    m = bb.sigma_map.mlp
    c = F.silu(F.linear(F.silu(F.linear(te, m[0].weight, m[0].bias)), m[2].weight, m[2].bias))
    mods = [F.linear(c, b.adaLN_modulation.weight, b.adaLN_modulation.bias) for b in bb.blocks]
    mods.append(F.linear(c, bb.output_layer.adaLN_modulation.weight,
                         bb.output_layer.adaLN_modulation.bias))
    return mods

  def _transpose(self, x, buf, colsum_out=None, ws=None):
    """x^T into `buf`; with colsum_out also the fp32 column sums of x (the bias
    gradient), whose first stage rides on the transpose's pass over x."""
    R, C = x.shape
    out = buf[:R * C].view(C, R)
    if colsum_out is None:
      return ops.transpose_bf16(x, out=out)
    nb = (R + 63) // 64
    part = ws['partc'].view(-1)[:nb * C]
    ops.transpose_bf16(x, out=out, colsum_partial=part)
    ops.colsum_reduce(part, colsum_out, C, nb, 1)
    return out

  def _wt(self, name, ws):
    w = self._w16(name)
    return self._transpose(w, ws['wT'])

  def _weight_grad(self, dy, a, dw, db, ws):
    """dw = dy^T a (+ db = column sums of dy).  Shapes that tile by 256 read the
    token-major operands in place (mdlm_gemm_tn_splitk); others go through
    transposed copies and the K-contiguous split-K GEMM."""
    T, N = dy.shape
    K = a.shape[1]
    if N % 256 == 0 and K % 256 == 0 and T % 64 == 0:
      s = ops.tn_splitk_plan(N, K, T)
      if s * N * K <= ws['skp'].numel() and s * N <= ws['skb'].numel():
        ops.gemm_tn_splitk(dy, a, dw, ws['skp'], bias_out=db, bias_partials=ws['skb'], splits=s)
        return
    dyT = self._transpose(dy, ws['tA'], db, ws)
    aT = self._transpose(a, ws['tB'])
    ops.gemm_bf16_splitk(dyT, aT, dw, ws['skp'])

  def _linear_bwd(self, dy, a, wname, bname, ws, da_out):
    """y = a W^T (+ b): G[W] = dy^T a, G[b] = colsum(dy), da = dy W."""
    self._weight_grad(dy, a, self._g(wname), None if bname is None else self._g(bname), ws)
    if da_out is not None:
      ops.gemm_bf16(dy, self._wt(wname, ws), da_out, EPI_STORE)
    return da_out

  # ----------------------------------------------------------- forward/backward
  def forward_backward(self, x0, attention_mask=None, t=None, u_xt=None,
                       accumulate=False, last=True):
    """Loss of one micro-batch and its gradients into the flat buffer G
    (overwritten).  Returns (token_nll device scalar, nlls (B, L)).

    Gradient accumulation (configs/config.yaml:68): micro-batches with
    `last=False` keep their gradients local (DDP no_sync) in a carry buffer;
    `accumulate=True` adds that carry to this micro-batch's gradients, bucket
    by bucket, just before each bucket's all-reduce.  The 1/accum mean is
    folded into the optimizer kernel's gradient scale."""
    self.buckets.hold = not last
    self.buckets.carry = self.A if accumulate else None
    if not accumulate:
      self.micro = 0
    self.micro += 1
    model, bb = self.model, self.bb
    B, L = x0.shape
    D, H, V = bb.hidden, bb.n_heads, bb.vocab_size
    M = B * L
    if M % 128 or L % 64 or D % 128:
      raise ValueError('Trainer: B*L must be a multiple of 128, L of 64, hidden of 128')
    ws = self._workspace(B, L)
    lo_hi = torch.stack(torch.aminmax(x0)).to(torch.int64)   # device; read with n_pad below
    if attention_mask is None:
      attention_mask = torch.ones(B, L, dtype=torch.float32, device=self.dev)
    attention_mask = attention_mask.float()
    # ---- noising (diffusion.py:847-869) -------------------------------------
    if t is None:
      t = model._sample_t(B, self.dev)
    sigma, dsigma = model.noise(t)
    move_chance = 1 - torch.exp(-sigma[:, None])
    xt = model.q_xt(x0, move_chance, u=u_xt)
    ops.masked_row_map(xt, ws['samples'], ws['n_samples'], L, model.mask_index,
                       ws['gather'], ws['scatter'], ws['n_out'], pad=256)
    # the only host read of the step (GPU queue still empty): the padded masked-row
    # count and the id range of x0 (ids outside the vocabulary raise, as the
    # reference's embedding lookup does, dit.py:298-299; nothing has consumed them yet
    # except q_xt / the row map, which only compare them with mask_index)
    n_pad, lo, hi = torch.cat([ws['n_out'].to(torch.int64), lo_hi]).tolist()
    if lo < 0 or hi >= V:
      raise IndexError(f'token id out of range: min {lo}, max {hi}, vocabulary size {V} '
                       '(tokenizer / checkpoint mismatch?)')
    sig_in = model._process_sigma(sigma[:, None])
    n_mod = B if model.time_conditioning else 1
    with torch.enable_grad():
      mods = self._conditioning(n_mod, sig_in[:n_mod])
    md = [m.detach() for m in mods]
    ld6, ld2 = 6 * D, 2 * D
    p = self.p_drop
    seed = model._rng_seed
    # ---- forward, activations saved -----------------------------------------
    x = ws['blocks'][0]['x_in']
    ops.embed_rows(self._p32('vocab_embed.embedding'), xt, x)
    for i, a in enumerate(ws['blocks']):
      pre_ = f'blocks.{i}.'
      mod = md[i]
      x = a['x_in']
      ops.ln_modulate(x, self._p32(pre_ + 'norm1.weight'), mod[:, 0:D], mod[:, D:2 * D],
                      a['h1'], L, n_mod, ld6)
      ops.gemm_bf16(a['h1'], self._w16(pre_ + 'attn_qkv.weight'), a['qkv'], EPI_QKV_ROTARY,
                    rows_per_mod=L, rot_cos=ws['cos'], rot_sin=ws['sin'], n_heads=H, ldc=0, M=M)
      q_, k_, vt_ = a['qkv'][:M * D], a['qkv'][M * D:2 * M * D], a['qkv'][2 * M * D:]
      ops.attn_fwd_lse(q_, k_, vt_, a['ao'], a['lse'], B, H, L)
      ops.gemm_bf16(a['ao'], self._w16(pre_ + 'attn_out.weight'), a['y1'], EPI_STORE)
      a['drop'][0] = model._draw(M * D // 4)[1]
      ops.dropout_gate_residual_fwd(a['y1'], mod[:, 2 * D:3 * D], x, a['x_mid'], L, n_mod,
                                    ld6, p, seed, a['drop'][0])
      ops.ln_modulate(a['x_mid'], self._p32(pre_ + 'norm2.weight'), mod[:, 3 * D:4 * D],
                      mod[:, 4 * D:5 * D], a['h2'], L, n_mod, ld6)
      # mlp.0 + GELU in one GEMM: the epilogue writes the pre-activation and the activation
      ops.gemm_bf16_gelu(a['h2'], self._w16(pre_ + 'mlp.0.weight'), a['act'], a['pre'],
                         bias=self._p32(pre_ + 'mlp.0.bias'))
      ops.gemm_bf16(a['act'], self._w16(pre_ + 'mlp.2.weight'), a['y2'], EPI_STORE,
                    bias=self._p32(pre_ + 'mlp.2.bias'))
      a['drop'][1] = model._draw(M * D // 4)[1]
      x_next = ws['blocks'][i + 1]['x_in'] if i + 1 < bb.n_blocks else ws['x_last']
      ops.dropout_gate_residual_fwd(a['y2'], mod[:, 5 * D:6 * D], a['x_mid'], x_next, L, n_mod,
                                    ld6, p, seed, a['drop'][1])
    # ---- output layer and loss on the MASKED rows only ---------------------------
    # SUBS gives an unmasked position log p = 0 whatever its logits
    # (diffusion.py:270-277), so only masked rows carry loss or gradient: the
    # final LN/modulate gathers them into a dense (n_pad, D) matrix (n_pad: count
    # rounded up to 256, known since q_xt) and the vocabulary projection, its
    # softmax backward and both of its backward GEMMs run on those rows alone.
    modf = md[-1]
    hfc, lgc, dlc = ws['hf'][:n_pad], ws['logits'][:n_pad], ws['dlogits'][:n_pad]
    weight = dsigma / torch.expm1(sigma)
    nlls = torch.zeros(B, L, dtype=torch.float32, device=self.dev)
    if n_pad:
      sc = ws['scatter'][:n_pad].long()
      valid = sc >= 0
      scc = sc.clamp(min=0)
      x0_c = x0.reshape(-1)[scc].contiguous()
      xt_c = torch.where(valid, torch.full_like(x0_c, model.mask_index), x0_c)
      w_c = ((weight[:, None] * attention_mask).reshape(-1)[scc] * valid).contiguous()
      ops.ln_modulate(ws['x_last'], self._p32('output_layer.norm_final.weight'), modf[:, 0:D],
                      modf[:, D:2 * D], hfc, L, n_mod, ld2, n_rows_dev=ws['n_out'],
                      row_gather=ws['gather'], rows=n_pad)
      ops.gemm_bf16(hfc, self._w16('output_layer.linear.weight'), lgc[:, :V], EPI_STORE,
                    bias=self._p32('output_layer.linear.bias'), ldc=lgc.stride(0))
      # ---- loss (diffusion.py:884-916) ----------------------------------------
      nll_c = ops.subs_nll(lgc[:, :V], xt_c[:, None], x0_c[:, None], w_c,
                           model.mask_index)
      nlls.view(-1).index_add_(0, scc, nll_c.view(-1) * valid)
    sums = ops.nll_sums(nlls, attention_mask)
    token_nll = (sums[0] / sums[1]).float()
    # ---- backward --------------------------------------------------------------
    self.G.zero_()
    ws['d_md'].zero_()
    if n_pad:
      row_w = (w_c / attention_mask.sum()).contiguous()
      ops.subs_nll_bwd(lgc[:, :V], xt_c, x0_c, row_w, dlc[:, :V], V, model.mask_index)
      # logits = hf Wv^T + bv (rows of dWv beyond V stay in the split-K partials)
      self._weight_grad(dlc, hfc, self._g('output_layer.linear.weight'), ws['bias_v'], ws)
      self._g('output_layer.linear.bias').copy_(ws['bias_v'][:V])
      ops.transpose_bf16(self._w16('output_layer.linear.weight'), out=ws['wvT'][:, :V],
                         ld_out=ws['wvT'].stride(0))
      # d(hf) rows go back to their positions; unmasked rows keep the zero
      ops.gemm_bf16(dlc, ws['wvT'], ws['d_md'], EPI_STORE, out_row_map=ws['scatter'])
    dx = ws['dx']
    red3 = torch.empty(n_mod, 3, D, device=self.dev)
    red1 = torch.empty(n_mod, D, device=self.dev)
    ops.ln_modulate_bwd(ws['d_md'], ws['x_last'], self._p32('output_layer.norm_final.weight'),
                        modf[:, D:2 * D], dx, ws['part3'], L, n_mod, ld2, accumulate=False)
    ops.colsum_reduce(ws['part3'], red3, 3 * D, (M // 32) // n_mod, n_mod)
    self._g('output_layer.norm_final.weight').copy_(red3[:, 0].sum(0))
    dmods = [None] * (bb.n_blocks + 1)
    dmods[-1] = torch.cat([red3[:, 2], red3[:, 1]], dim=1)
    with torch.enable_grad():
      torch.autograd.backward([mods[-1]], [dmods[-1]], retain_graph=True)
    self.buckets.launch(0)
    for i in reversed(range(bb.n_blocks)):
      a = ws['blocks'][i]
      pre_ = f'blocks.{i}.'
      mod = md[i]
      dm = torch.empty(n_mod, 6 * D, device=self.dev)
      # x_out = x_mid + gate_mlp * dropout(y2)
      ops.dropout_gate_residual_bwd(dx, a['y2'], mod[:, 5 * D:6 * D], ws['d_md'], ws['part1'],
                                    L, n_mod, ld6, p, seed, a['drop'][1])
      ops.colsum_reduce(ws['part1'], red1, D, (M // 32) // n_mod, n_mod)
      dm[:, 5 * D:] = red1
      # dW2 / db2, then d(pre) = (dY W2) * gelu'(pre) straight out of the dX GEMM
      self._linear_bwd(ws['d_md'], a['act'], pre_ + 'mlp.2.weight', pre_ + 'mlp.2.bias', ws, None)
      ops.gemm_bf16_gelu(ws['d_md'], self._wt(pre_ + 'mlp.2.weight', ws), ws['d_m4b'], a['pre'],
                         grad=True)
      self._linear_bwd(ws['d_m4b'], a['h2'], pre_ + 'mlp.0.weight', pre_ + 'mlp.0.bias', ws, ws['d_md'])
      ops.ln_modulate_bwd(ws['d_md'], a['x_mid'], self._p32(pre_ + 'norm2.weight'),
                          mod[:, 4 * D:5 * D], dx, ws['part3'], L, n_mod, ld6, accumulate=True)
      ops.colsum_reduce(ws['part3'], red3, 3 * D, (M // 32) // n_mod, n_mod)
      self._g(pre_ + 'norm2.weight').copy_(red3[:, 0].sum(0))
      dm[:, 3 * D:4 * D] = red3[:, 2]
      dm[:, 4 * D:5 * D] = red3[:, 1]
      # x_mid = x_in + gate_msa * dropout(y1)
      ops.dropout_gate_residual_bwd(dx, a['y1'], mod[:, 2 * D:3 * D], ws['d_md'], ws['part1'],
                                    L, n_mod, ld6, p, seed, a['drop'][0])
      ops.colsum_reduce(ws['part1'], red1, D, (M // 32) // n_mod, n_mod)
      dm[:, 2 * D:3 * D] = red1
      dao = ws['d_m4'].view(-1)[:M * D].view(M, D)
      self._linear_bwd(ws['d_md'], a['ao'], pre_ + 'attn_out.weight', None, ws, dao)
      q_, k_, vt_ = a['qkv'][:M * D], a['qkv'][M * D:2 * M * D], a['qkv'][2 * M * D:]
      ops.attn_bwd(q_, k_, vt_, a['ao'], dao, a['lse'], ws['delta'], ws['dq'], ws['dk'],
                   ws['dv'], B, H, L)
      ops.qkv_grad_pack(ws['dq'], ws['dk'], ws['dv'], ws['cos'], ws['sin'], ws['d_qkv'], B, H, L)
      self._linear_bwd(ws['d_qkv'], a['h1'], pre_ + 'attn_qkv.weight', None, ws, ws['d_md'])
      ops.ln_modulate_bwd(ws['d_md'], a['x_in'], self._p32(pre_ + 'norm1.weight'),
                          mod[:, D:2 * D], dx, ws['part3'], L, n_mod, ld6, accumulate=True)
      ops.colsum_reduce(ws['part3'], red3, 3 * D, (M // 32) // n_mod, n_mod)
      self._g(pre_ + 'norm1.weight').copy_(red3[:, 0].sum(0))
      dm[:, 0:D] = red3[:, 2]
      dm[:, D:2 * D] = red3[:, 1]
      dmods[i] = dm
      with torch.enable_grad():
        torch.autograd.backward([mods[i]], [dm], retain_graph=i > 0)
      self.buckets.launch(self.bucket_ids[i])
    ops.embed_grad(dx, xt.view(-1), self._g('vocab_embed.embedding'),
                   hot_index=model.mask_index, hot_partial=ws['hot'])
    self.buckets.launch(len(self.buckets.slices) - 1)
    if not last:
      if self.A is None:
        self.A = torch.zeros_like(self.G)
      if accumulate:
        self.A += self.G
      else:
        self.A.copy_(self.G)
    return token_nll, nlls

  # ----------------------------------------------------------------- optimizer
  def optimizer_step(self):
    """clip_grad_norm_ + AdamW + lr schedule + EMA (diffusion.py:254-259,
    449-471) and the bf16 refresh of the GEMM weights, one launch."""
    self.buckets.wait_all()
    lr = constant_warmup_lr(self.lr, self.step_count, self.num_warmup_steps)
    self.step_count += 1
    ops.grad_sumsq(self.G, self._sumsq_part, self._sumsq)
    scale = 1.0 / (self.world * max(1, self.micro))
    self.last_lr = lr
    self.last_grad_norm = self._sumsq  # device: sum of squares of the summed grads
    n = self.step_count
    omd = 0.0
    if self.E is not None:   # models/ema.py:40-45 on the EMA's own counter
      self.ema_updates += 1
      ne = self.ema_updates
      omd = 1.0 - min(self.ema_decay, (1 + ne) / (10 + ne))
    ops.adamw_ema(self.P, self.G, self.M1, self.M2, self.E, self.W16, lr, self.betas[0],
                  self.betas[1], self.eps, self.weight_decay, n, omd, sumsq=self._sumsq,
                  max_norm=self.max_norm or 0.0, grad_scale=scale)
    if getattr(self.model, 'ema', None) is not None:
      self.model.ema.num_updates = self.ema_updates
    self.bb.refresh()
    return lr

  def grad_norm(self):
    return math.sqrt(float(self._sumsq.item())) / (self.world * max(1, self.micro))

  def training_step(self, batch, accumulate=False, step=True):
    """diffusion.py:390-397 + optimizer_step (254-259): returns the loss.
    `accumulate`: add to the gradients of the previous micro-batches;
    `step=False`: not the last micro-batch of this optimizer step."""
    loss, nlls = self.forward_backward(batch['input_ids'], batch.get('attention_mask'),
                                       accumulate=accumulate, last=step)
    self.model.train_metrics.update(nlls, batch.get('attention_mask')
                                    if batch.get('attention_mask') is not None
                                    else torch.ones_like(nlls))
    if step:
      self.optimizer_step()
    return loss
