"""GPU parity tests for the training-step kernels: each
hand-written backward against torch autograd of the CPU oracle's fp32 op on the
same seeded inputs.  bf16 tolerance 2e-2 (relative L2), fp32 paths 1e-4."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

import mdlm_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
  assert torch.cuda.is_available()
  from mdlm_amd import ops as _ops
  return _ops


def _bf(t):
  return t.to(torch.bfloat16)


def _rel(a, b):
  a, b = a.detach().double().cpu(), b.detach().double().cpu()
  return ((a - b).norm() / b.norm().clamp(min=1e-12)).item()


def test_ln_modulate_bwd(ops):
  torch.manual_seed(0)
  M, D, L = 256, 768, 128
  x = torch.randn(M, D) * 2 + 0.3
  w = 1 + 0.1 * torch.randn(D)
  mod = torch.randn(M // L, 2 * D) * 0.3         # shift | scale
  dh = _bf(torch.randn(M, D))
  xr, wr, mr = (t.clone().requires_grad_(True) for t in (x, w, mod))
  h = orc.modulate(orc.layer_norm_w(xr.view(M // L, L, D), wr),
                   mr[:, None, :D], mr[:, None, D:])
  h.backward(dh.float().view(M // L, L, D))
  dx0 = torch.randn(M, D)
  dx = dx0.clone().cuda()
  nblk = M // 32
  partial = torch.empty(nblk, 3, D, device='cuda')
  modc = mod.cuda()
  ops.ln_modulate_bwd(dh.cuda(), x.cuda(), w.cuda(), modc[:, D:], dx, partial, L,
                      M // L, 2 * D, accumulate=True)
  assert _rel(dx.cpu() - dx0, xr.grad) < 1e-4
  red = torch.empty(M // L, 3, D, device='cuda')
  ops.colsum_reduce(partial, red, 3 * D, L // 32, M // L)
  assert _rel(red[:, 0].sum(0), wr.grad) < 1e-4
  assert _rel(red[:, 1], mr.grad[:, D:]) < 1e-4
  assert _rel(red[:, 2], mr.grad[:, :D]) < 1e-4
  # overwrite mode
  dx2 = torch.full((M, D), 7.0, device='cuda')
  ops.ln_modulate_bwd(dh.cuda(), x.cuda(), w.cuda(), modc[:, D:], dx2, partial, L,
                      M // L, 2 * D, accumulate=False)
  assert _rel(dx2, xr.grad) < 1e-4


@pytest.mark.parametrize('p', [0.0, 0.1])
def test_dropout_gate_residual(ops, p):
  torch.manual_seed(1)
  M, D, L = 256, 768, 128
  y = _bf(torch.randn(M, D))
  gate = torch.randn(M // L, D)
  res = torch.randn(M, D)
  dout = torch.randn(M, D)
  seed, offset = 1234, 77
  u = ops.philox_uniform(M, D, seed, offset)
  keep = (u.cpu() >= p) if p > 0 else None
  if p > 0:
    assert abs(keep.float().mean().item() - (1 - p)) < 0.01
  yr = y.float().requires_grad_(True)
  gr = gate.clone().requires_grad_(True)
  out_ref = res + (gr[:, None, :] * orc._dropout(yr, keep, p).view(M // L, L, D)).view(M, D)
  out_ref.backward(dout)
  out = torch.empty(M, D, device='cuda')
  ops.dropout_gate_residual_fwd(y.cuda(), gate.cuda(), res.cuda(), out, L, M // L, D,
                                p, seed, offset)
  assert (out.cpu() - out_ref.detach()).abs().max().item() < 1e-5
  dy = torch.empty(M, D, dtype=torch.bfloat16, device='cuda')
  partial = torch.empty(M // 32, D, device='cuda')
  ops.dropout_gate_residual_bwd(dout.cuda(), y.cuda(), gate.cuda(), dy, partial, L,
                                M // L, D, p, seed, offset)
  assert _rel(dy.float(), yr.grad) < 5e-3
  dgate = torch.empty(M // L, D, device='cuda')
  ops.colsum_reduce(partial, dgate, D, L // 32, M // L)
  assert _rel(dgate, gr.grad) < 1e-4


def test_gelu_transpose_colsum_embed(ops):
  torch.manual_seed(2)
  pre = _bf(torch.randn(256, 3072) * 2)
  dact = _bf(torch.randn(256, 3072))
  pr = pre.float().requires_grad_(True)
  a = F.gelu(pr, approximate='tanh')
  a.backward(dact.float())
  act = torch.empty_like(pre, device='cuda')
  ops.gelu_fwd(pre.cuda(), act)
  assert _rel(act.float(), a) < 5e-3
  dpre = torch.empty_like(pre, device='cuda')
  ops.gelu_bwd(dact.cuda(), pre.cuda(), dpre)
  assert _rel(dpre.float(), pr.grad) < 5e-3
  # transpose: plain, padded source rows, batched
  x = _bf(torch.randn(200, 330))
  assert torch.equal(ops.transpose_bf16(x.cuda()).cpu(), x.t().contiguous())
  xp = _bf(torch.randn(130, 384)).cuda()
  assert torch.equal(ops.transpose_bf16(xp[:, :322]).cpu(), xp[:, :322].t().contiguous().cpu())
  xb = _bf(torch.randn(6, 64, 128)).cuda()
  assert torch.equal(ops.transpose_bf16(xb).cpu(), xb.transpose(1, 2).contiguous().cpu())
  # 16-byte path: R, C, leading dimensions multiples of 8 (ragged 64x64 tiles, padded rows)
  for R, C, ld in ((200, 328, 328), (1024, 768, 768), (136, 72, 96)):
    xv = _bf(torch.randn(R, ld)).cuda()
    assert torch.equal(ops.transpose_bf16(xv[:, :C]).cpu(), xv[:, :C].t().contiguous().cpu()), (R, C)
  xb = _bf(torch.randn(3, 72, 200)).cuda()
  assert torch.equal(ops.transpose_bf16(xb).cpu(), xb.transpose(1, 2).contiguous().cpu())
  # column sums (bias gradient) riding on the transpose
  xv = _bf(torch.randn(1000, 776)).cuda()
  part = torch.full((16, 776), 7.0, device='cuda')
  xt_ = ops.transpose_bf16(xv, colsum_partial=part)
  assert torch.equal(xt_.cpu(), xv.t().contiguous().cpu())
  cs = ops.colsum_reduce(part, torch.empty(776, device='cuda'), 776, 16, 1)
  assert _rel(cs, xv.float().sum(0)) < 1e-5
  # column sums
  z = _bf(torch.randn(1000, 770))
  assert _rel(ops.colsum_bf16(z.cuda()), z.float().sum(0)) < 1e-5
  # embedding gradient (atomics: order-dependent rounding only)
  idx = torch.randint(0, 50, (512,))
  dx = torch.randn(512, 64)
  ref = torch.zeros(50, 64).index_add_(0, idx, dx)
  dt = torch.zeros(50, 64, device='cuda')
  ops.embed_grad(dx.cuda(), idx.cuda(), dt)
  assert (dt.cpu() - ref).abs().max().item() < 1e-4
  # hot row (the mask token): summed in a fixed order, written not added
  idx[::2] = 7
  ref = torch.zeros(50, 64).index_add_(0, idx, dx)
  dt = torch.zeros(50, 64, device='cuda'); dt[7] = 123.0
  ops.embed_grad(dx.cuda(), idx.cuda(), dt, hot_index=7)
  assert (dt.cpu() - ref).abs().max().item() < 1e-4
  dt2 = torch.zeros(50, 64, device='cuda')
  ops.embed_grad(dx.cuda(), idx.cuda(), dt2, hot_index=7)
  assert torch.equal(dt2[7], dt[7])


def test_qkv_grad_pack(ops):
  torch.manual_seed(3)
  B, H, L, d = 2, 3, 128, 64
  cos, sin = orc.rotary_cos_sin(L, d)
  cos, sin = _bf(cos).float(), _bf(sin).float()
  g = _bf(torch.randn(B, L, 3, H, d))            # gradient w.r.t. the rotated qkv
  qkv = torch.randn(B, L, 3, H, d, requires_grad=True)
  orc.apply_rotary_qk(qkv, cos, sin).backward(g.float())
  hm = lambda i: g[:, :, i].permute(0, 2, 1, 3).contiguous().cuda()   # (B,H,L,d)
  out = torch.empty(B * L, 3 * H * d, dtype=torch.bfloat16, device='cuda')
  ops.qkv_grad_pack(hm(0), hm(1), hm(2), cos.cuda(), sin.cuda(), out, B, H, L)
  assert _rel(out.float(), qkv.grad.reshape(B * L, 3 * H * d)) < 5e-3


@pytest.mark.parametrize('L', [128, 256])
def test_attention_backward(ops, L):
  torch.manual_seed(4)
  B, H, d = 2, 3, 64
  q, k, v = (_bf(torch.randn(B, H, L, d)) for _ in range(3))
  do = _bf(torch.randn(B, L, H * d))
  qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
  o_ref = F.scaled_dot_product_attention(qr, kr, vr).transpose(1, 2).reshape(B, L, H * d)
  o_ref.backward(do.float())
  qc, kc = q.cuda(), k.cuda()
  vt = v.transpose(2, 3).contiguous().cuda()
  o = torch.empty(B * L, H * d, dtype=torch.bfloat16, device='cuda')
  lse = torch.empty(B, H, L, device='cuda')
  ops.attn_fwd_lse(qc, kc, vt, o, lse, B, H, L)
  assert _rel(o.float().view(B, L, H * d), o_ref) < 1e-2
  s = (qr @ kr.transpose(2, 3)).detach() * 0.125
  lse_ref = torch.logsumexp(s, -1) * math.log2(math.e)
  assert (lse.cpu() - lse_ref).abs().max().item() < 2e-2
  # without lse the plain entry point gives the same output
  o2 = torch.empty_like(o)
  ops.attn_fwd(qc, kc, vt, o2, B, H, L)
  assert torch.equal(o, o2)
  dq, dk, dv = (torch.empty(B, H, L, d, dtype=torch.bfloat16, device='cuda') for _ in range(3))
  ws = torch.empty(B * H * L, device='cuda')
  ops.attn_bwd(qc, kc, vt, o, do.view(B * L, H * d).cuda(), lse, ws, dq, dk, dv, B, H, L)
  assert _rel(dv.float(), vr.grad) < 2e-2
  assert _rel(dq.float(), qr.grad) < 2e-2
  assert _rel(dk.float(), kr.grad) < 2e-2
  dq2, dk2, dv2 = (torch.empty_like(dq) for _ in range(3))
  ops.attn_bwd(qc, kc, vt, o, do.view(B * L, H * d).cuda(), lse, ws, dq2, dk2, dv2, B, H, L)
  assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)


@pytest.mark.parametrize('V,mask,dtype', [(30522, 103, torch.bfloat16), (1000, 999, torch.float32)])
def test_subs_nll_bwd(ops, V, mask, dtype):
  torch.manual_seed(5)
  B, L = 2, 16
  ldp = ops.padded_vocab(V)
  logits = (torch.randn(B, L, V) * 3).to(dtype)
  x0 = torch.randint(0, V, (B, L))
  x0[x0 == mask] = 0
  xt = x0.clone()
  xt[torch.rand(B, L) < 0.6] = mask
  t = torch.tensor([0.3, 0.8])
  attn = torch.ones(B, L); attn[1, -3:] = 0
  lr = logits.float().requires_grad_(True)
  nlls, token_nll = orc.nelbo_from_logits(lr, xt, x0, t, attn, mask)
  token_nll.backward()
  sigma, dsigma = orc.loglinear_noise(t)
  row_w = ((dsigma / torch.expm1(sigma))[:, None] * attn / attn.sum()).reshape(-1)
  lg = torch.zeros(B * L, ldp, dtype=dtype, device='cuda')
  lg[:, :V] = logits.view(B * L, V).cuda()
  dl = torch.full((B * L, ldp), 9.0, dtype=torch.bfloat16, device='cuda')
  ops.subs_nll_bwd(lg[:, :V], xt.view(-1).cuda(), x0.view(-1).cuda(), row_w.cuda(),
                   dl[:, :V], V, mask)
  assert (dl[:, V:] == 0).all()
  assert (dl[:, mask] == 0).all()
  assert (dl[(xt.view(-1) != mask).cuda()] == 0).all()
  assert _rel(dl[:, :V].float(), lr.grad.view(B * L, V)) < 5e-3


def test_adamw_ema_clip(ops):
  torch.manual_seed(6)
  n = 100003
  p = torch.randn(n); g = torch.randn(n) * 0.01
  st = dict(p=p.clone(), m=torch.zeros(n), v=torch.zeros(n), s=p.clone())
  pc, mc, vc, sc = p.clone().cuda(), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda'), p.clone().cuda()
  w16 = torch.empty(n, dtype=torch.bfloat16, device='cuda')
  part = torch.empty(1024, device='cuda'); ss = torch.empty(1, device='cuda')
  for step in range(1, 4):
    gi = g * step
    total, coef = orc.clip_coef({'g': gi}, 1.0)
    orc.adamw_step(st['p'], gi * coef, st['m'], st['v'], step, 3e-4)
    orc.ema_update(st['s'], st['p'], 0.9999, step)
    gc = gi.cuda()
    ops.grad_sumsq(gc, part, ss)
    assert abs(math.sqrt(ss.item()) - total.item()) < 1e-4 * total.item()
    ops.adamw_ema(pc, gc, mc, vc, sc, w16, 3e-4, 0.9, 0.999, 1e-8, 0.0, step,
                  1.0 - orc.ema_decay(0.9999, step), sumsq=ss, max_norm=1.0)
    assert (pc.cpu() - st['p']).abs().max().item() < 2e-6
    assert (sc.cpu() - st['s']).abs().max().item() < 2e-6
    assert _rel(mc, st['m']) < 1e-4 and _rel(vc, st['v']) < 1e-4
    assert torch.equal(w16.cpu(), pc.cpu().to(torch.bfloat16))


# ----------------------------------------------------------- whole training step
V_T, MASK_T = 300, 103     # mask id inside the vocabulary, like bert-base-uncased (LM1B)


def _make_trainer(p_drop, time_conditioning=False, seed=5, extra=()):
  from mdlm_amd.config import load_config
  from mdlm_amd.diffusion import Diffusion
  from mdlm_amd.tokenizer import SyntheticTokenizer
  from mdlm_amd.train import Trainer
  cfg = load_config(['model=tiny', 'loader.eval_batch_size=4',
                     f'time_conditioning={time_conditioning}', *extra])
  tok = SyntheticTokenizer(vocab_size=V_T, eos_token_id=7, mask_token_id=MASK_T)
  model = Diffusion(cfg, tok).cuda()
  assert model.mask_index == MASK_T and model.vocab_size == V_T
  m = cfg.model
  p = orc.synthetic_dit_params(m.hidden_size, m.n_heads, m.n_blocks, m.cond_dim, V_T, seed=seed)
  sd = dict(p); sd['rotary_emb.inv_freq'] = model.backbone.rotary_emb.inv_freq
  model.backbone.load_state_dict(sd)
  tr = Trainer(model, lr=3e-3, num_warmup_steps=2, dropout=p_drop)
  return model, cfg, p, tr


def _batch(B, L, seed=0):
  g = torch.Generator().manual_seed(seed)
  x0 = torch.randint(0, V_T, (B, L), generator=g)
  x0[x0 == MASK_T] = 0
  attn = torch.ones(B, L); attn[1, -5:] = 0
  t = orc.sample_t(B, torch.rand(B, generator=g))
  u = torch.rand(B, L, generator=g)
  return x0, attn, t, u


@pytest.mark.parametrize('p_drop', [0.0, 0.1])
def test_training_step_matches_oracle(ops, p_drop):
  model, cfg, p, tr = _make_trainer(p_drop)
  m = cfg.model
  B, L, D = 4, m.length, m.hidden_size
  x0, attn, t, u = _batch(B, L)
  loss, nlls = tr.forward_backward(x0.cuda(), attn.cuda(), t=t.cuda(), u_xt=u.cuda())
  keep = None
  if p_drop > 0:
    keep = []
    for a in tr._ws['blocks']:
      keep.append(tuple(
        (ops.philox_uniform(B * L, D, model._rng_seed, off).cpu() >= p_drop).view(B, L, D)
        for off in a['drop']))
  ref_loss, ref_nlls, ref_g = orc.loss_and_grads(p, x0, attn, t, u, m.n_heads, m.n_blocks,
                                                 MASK_T, keep, p_drop)
  assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item())
  assert _rel(nlls, ref_nlls) < 2e-2
  g = tr.grads()
  worst = max((_rel(g[k], ref_g[k]), k) for k in ref_g)
  flat = torch.cat([g[k].reshape(-1).cpu() for k in ref_g]).double()
  flat_ref = torch.cat([ref_g[k].reshape(-1) for k in ref_g]).double()
  cos = (flat @ flat_ref) / (flat.norm() * flat_ref.norm())
  assert cos.item() > 0.999, cos.item()
  assert worst[0] < 6e-2, worst
  # the optimizer consumes those gradients exactly as the oracle's AdamW does
  st = orc.new_train_state({k: v.clone() for k, v in p.items()})
  gcpu = {k: g[k].detach().cpu().clone() for k in ref_g}
  total, coef = orc.clip_coef(gcpu, 1.0)
  lr = tr.optimizer_step()
  assert lr == orc.constant_warmup_lr(3e-3, 0, 2) == 0.0
  assert abs(tr.grad_norm() - total.item()) < 1e-4 * total.item()
  # second step with a non-zero learning rate
  tr.forward_backward(x0.cuda(), attn.cuda(), t=t.cuda(), u_xt=u.cuda())
  g2 = {k: v.detach().cpu().clone() for k, v in tr.grads().items()}
  lr = tr.optimizer_step()
  assert abs(lr - 1.5e-3) < 1e-12
  for step, gg, lr_ in ((1, gcpu, 0.0), (2, g2, 1.5e-3)):
    _, cf = orc.clip_coef(gg, 1.0)
    st['step'] = step
    for k in st['params']:
      orc.adamw_step(st['params'][k], gg[k] * cf, st['m'][k], st['v'][k], step, lr_)
      orc.ema_update(st['shadow'][k], st['params'][k], 0.9999, step)
  sd = model.backbone.state_dict()
  ema = tr.ema_state_dict()
  for i, k in enumerate(tr.names):
    assert (sd[k].cpu() - st['params'][k]).abs().max().item() < 2e-6, k
    assert (ema['shadow_params'][i].cpu() - st['shadow'][k]).abs().max().item() < 2e-6, k
  assert ema['num_updates'] == 2
  # the eval path sees the new weights (bf16 copies refreshed)
  lg = model.backbone(x0.cuda(), torch.zeros(B, device='cuda')).float().cpu()
  ref_lg = orc.dit_forward(st['params'], x0, torch.zeros(B), m.n_heads, m.n_blocks)
  assert _rel(lg, ref_lg) < 2e-2


@pytest.mark.parametrize('B,L', [(3, 128), (5, 256), (16, 128)])
def test_training_gradients_other_batch_shapes(ops, B, L):
  """forward_backward against the oracle's autograd at row counts that are odd
  multiples of 128 (128x128 GEMM tiles, partial split-K ranges) and at L = 256."""
  model, cfg, p, tr = _make_trainer(0.0, extra=[f'model.length={L}', f'loader.eval_batch_size={B}'])
  m = cfg.model
  x0, attn, t, u = _batch(B, L, seed=B * 10 + L)
  loss, nlls = tr.forward_backward(x0.cuda(), attn.cuda(), t=t.cuda(), u_xt=u.cuda())
  ref_loss, ref_nlls, ref_g = orc.loss_and_grads(p, x0, attn, t, u, m.n_heads, m.n_blocks,
                                                 MASK_T, None, 0.0)
  assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item())
  assert _rel(nlls, ref_nlls) < 2e-2
  g = tr.grads()
  flat = torch.cat([g[k].reshape(-1).cpu() for k in ref_g]).double()
  flat_ref = torch.cat([ref_g[k].reshape(-1) for k in ref_g]).double()
  assert ((flat @ flat_ref) / (flat.norm() * flat_ref.norm())).item() > 0.999
  assert max(_rel(g[k], ref_g[k]) for k in ref_g) < 6e-2


def test_training_reduces_loss_and_time_conditioning(ops):
  model, cfg, p, tr = _make_trainer(0.1, time_conditioning=True)
  B, L = 4, cfg.model.length
  x0, attn, t, u = _batch(B, L, seed=3)
  # per-sample conditioning: gradient parity with dropout off is covered above;
  # here the time-conditioned path runs end to end and the loss goes down
  tr.num_warmup_steps = 0
  losses = []
  for _ in range(12):
    loss, _ = tr.forward_backward(x0.cuda(), attn.cuda(), t=t.cuda(), u_xt=u.cuda())
    tr.optimizer_step()
    losses.append(loss.item())
  assert all(math.isfinite(v) for v in losses)
  assert losses[-1] < 0.8 * losses[0], losses


def test_time_conditioned_gradients(ops):
  model, cfg, p, tr = _make_trainer(0.0, time_conditioning=True)
  m = cfg.model
  B, L = 4, m.length
  x0, attn, t, u = _batch(B, L, seed=4)
  tr.forward_backward(x0.cuda(), attn.cuda(), t=t.cuda(), u_xt=u.cuda())
  sigma, dsigma = orc.loglinear_noise(t)
  leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
  xt = orc.q_xt(x0, 1 - torch.exp(-sigma[:, None]), u, MASK_T)
  logits = orc.dit_forward(leaves, xt, sigma, m.n_heads, m.n_blocks)
  nlls, token_nll = orc.nelbo_from_logits(logits, xt, x0, t, attn, MASK_T)
  grads = torch.autograd.grad(token_nll, list(leaves.values()))
  g = tr.grads()
  for k, gr in zip(leaves, grads):
    assert _rel(g[k], gr) < 6e-2, k


def test_gradient_accumulation(ops):
  """configs/config.yaml:68 accumulate_grad_batches: the carry of the held
  micro-batches is added bucket by bucket, the mean is taken in the optimizer."""
  model, cfg, p, tr = _make_trainer(0.0)
  B, L = 4, cfg.model.length
  xa, attn, ta, ua = _batch(B, L, seed=11)
  xb, _, tb, ub = _batch(B, L, seed=12)
  tr.forward_backward(xa.cuda(), attn.cuda(), t=ta.cuda(), u_xt=ua.cuda())
  ga = {k: v.detach().clone() for k, v in tr.grads().items()}
  tr.forward_backward(xb.cuda(), attn.cuda(), t=tb.cuda(), u_xt=ub.cuda())
  gb = {k: v.detach().clone() for k, v in tr.grads().items()}
  tr.forward_backward(xa.cuda(), attn.cuda(), t=ta.cuda(), u_xt=ua.cuda(), last=False)
  tr.forward_backward(xb.cuda(), attn.cuda(), t=tb.cuda(), u_xt=ub.cuda(), accumulate=True)
  g = tr.grads()
  for k in ga:
    ref = ga[k] + gb[k]
    assert (g[k] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-9, k
  assert tr.micro == 2
  tr.optimizer_step()
  total = torch.sqrt(sum(((ga[k] + gb[k]) / 2).double().pow(2).sum() for k in ga)).item()
  assert abs(tr.grad_norm() - total) < 1e-4 * total
  # a step without accumulation resets the micro-batch count
  tr.forward_backward(xa.cuda(), attn.cuda(), t=ta.cuda(), u_xt=ua.cuda())
  assert tr.micro == 1


@pytest.mark.parametrize('M,N,K,splits', [(256, 768, 8192, 8), (384, 128, 4096, 4),
                                          (768, 768, 2048, None), (512, 300, 1024, 1)])
def test_gemm_splitk(ops, M, N, K, splits):
  """Weight-gradient GEMM: K (tokens) cut into ranges, partial sums added in
  split order; operands may be column slices of wider buffers (lda, ldw > K)."""
  g = torch.Generator().manual_seed(M + N)
  A = (torch.randn(M, K + 64, generator=g) * 0.5).to(torch.bfloat16)
  W = (torch.randn(N, K + 128, generator=g) * 0.5).to(torch.bfloat16)
  ref = A[:, :K].double() @ W[:, :K].double().T
  s = ops.splitk_plan(M, N, K) if splits is None else splits
  part = torch.empty(s * M * N, dtype=torch.float32, device='cuda')
  out = torch.empty(M, N, dtype=torch.float32, device='cuda')
  ops.gemm_bf16_splitk(A.cuda()[:, :K], W.cuda()[:, :K], out, part, splits=splits)
  err = (out.cpu().double() - ref).abs().max().item()
  assert err < 2e-4 * K ** 0.5, err
  # run-to-run determinism (fixed summation order, no atomics)
  out2 = torch.empty_like(out)
  ops.gemm_bf16_splitk(A.cuda()[:, :K], W.cuda()[:, :K], out2, part, splits=splits)
  assert torch.equal(out, out2)
  if splits is None:
    assert s == 4   # 9 tiles of 256x256; each range keeps >= 512 tokens: 2048 / 512 = 4


_DDP_WORKER = r'''
import os, sys, torch
root = sys.argv[1]; out = sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests')); sys.path.insert(0, os.path.join(root, 'oracle'))
from mdlm_amd import dist as mdist
rank, local_rank, world = mdist.init(backend='gloo')
torch.cuda.set_device(mdist.local_device(local_rank))
import test_gpu_train as T
model, cfg, p, tr = T._make_trainer(0.0)
assert tr.world == 2 and tr.buckets.enabled
B, L = 4, cfg.model.length
x0, attn, t, u = T._batch(B, L, seed=21 + rank)
for _ in range(2):   # second step has a non-zero learning rate
  tr.forward_backward(x0.cuda(), attn.cuda(), t=t.cuda(), u_xt=u.cuda())
  tr.optimizer_step()
torch.cuda.synchronize()
torch.save({'P': tr.P.cpu(), 'G': tr.G.cpu(), 'norm': tr.grad_norm()}, out + f'.{rank}')
mdist.barrier()
'''


def test_ddp_two_ranks_match_accumulation(ops, tmp_path):
  """DDP (configs/strategy/ddp.yaml): two ranks, each with its own micro-batch,
  bucketed gradient all-reduce on the side stream (gloo here, RCCL on a node)
  -> the same parameters as one process accumulating both micro-batches."""
  import socket, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  script = tmp_path / 'ddp_worker.py'
  script.write_text(_DDP_WORKER)
  out = str(tmp_path / 'res')
  procs = []
  for r in range(2):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2',
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    procs.append(subprocess.Popen([sys.executable, str(script), root, out], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
  outs = [p.communicate(timeout=110) for p in procs]
  for p, (o, e) in zip(procs, outs):
    assert p.returncode == 0, e[-3000:]
  r0, r1 = (torch.load(out + f'.{r}', weights_only=True) for r in range(2))
  assert torch.equal(r0['P'], r1['P']) and torch.equal(r0['G'], r1['G'])   # replicas stay in step
  model, cfg, p, tr = _make_trainer(0.0)
  B, L = 4, cfg.model.length
  batches = [_batch(B, L, seed=21 + r) for r in range(2)]
  for _ in range(2):
    for i, (x0, attn, t, u) in enumerate(batches):
      tr.forward_backward(x0.cuda(), attn.cuda(), t=t.cuda(), u_xt=u.cuda(),
                          accumulate=i > 0, last=i == 1)
    tr.optimizer_step()
  assert abs(tr.grad_norm() - r0['norm']) < 1e-4 * r0['norm']
  # summed gradients agree to fp32 rounding (atomics in the embedding gradient
  # are order-dependent); Adam turns a rounding-level gradient into a full-size
  # update where the true gradient cancels to ~0, hence the fraction bound
  g_ref, g_ddp = tr.G.cpu().double(), r0['G'].double()
  assert (g_ref - g_ddp).norm() < 1e-5 * g_ref.norm()
  diff = (tr.P.cpu() - r0['P']).abs()
  assert (diff > 2e-5).float().mean().item() < 1e-3, diff.max().item()


@pytest.mark.parametrize('T,N,K,splits', [(1024, 256, 512, 3), (4096, 768, 768, None),
                                          (64, 256, 256, 1), (2048, 512, 256, 7),
                                          (512, 30720, 256, 2)])   # the LM1B output layer's row count
def test_gemm_tn_splitk(ops, T, N, K, splits):
  """Weight + bias gradient straight from token-major dY (T, N) and X (T, K):
  transposing LDS reads, uneven token ranges, padded leading dimensions."""
  g = torch.Generator().manual_seed(T + N + K)
  dY = (torch.randn(T, N + 64, generator=g) * 0.5).to(torch.bfloat16)
  X = (torch.randn(T, K + 8, generator=g) * 0.5).to(torch.bfloat16)
  ref = dY[:, :N].double().T @ X[:, :K].double()
  ref_b = dY[:, :N].double().sum(0)
  s = ops.tn_splitk_plan(N, K, T) if splits is None else splits
  part = torch.empty(s * N * K, dtype=torch.float32, device='cuda')
  bpart = torch.empty(s * N, dtype=torch.float32, device='cuda')
  out = torch.empty(N, K, dtype=torch.float32, device='cuda')
  bias = torch.empty(N, dtype=torch.float32, device='cuda')
  dYc, Xc = dY.cuda()[:, :N], X.cuda()[:, :K]
  ops.gemm_tn_splitk(dYc, Xc, out, part, bias_out=bias, bias_partials=bpart, splits=splits)
  err = (out.cpu().double() - ref).abs().max().item()
  assert err < 2e-4 * T ** 0.5, err
  assert (bias.cpu().double() - ref_b).abs().max().item() < 1e-4 * T ** 0.5
  out2 = torch.empty_like(out)
  ops.gemm_tn_splitk(dYc, Xc, out2, part, splits=splits)   # no bias; bit-reproducible
  assert torch.equal(out, out2)


def test_training_gradients_token_major_path(ops):
  """hidden 256: every weight-gradient shape tiles by 256, so the Trainer takes
  the token-major GEMM (no transposed copies) for all Linears and the output layer."""
  model, cfg, p, tr = _make_trainer(0.0, extra=('model.hidden_size=256', 'model.n_heads=4'))
  m = cfg.model
  B, L = 4, m.length
  x0, attn, t, u = _batch(B, L, seed=31)
  calls = []
  orig = ops.gemm_tn_splitk
  ops.gemm_tn_splitk = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
  try:
    loss, nlls = tr.forward_backward(x0.cuda(), attn.cuda(), t=t.cuda(), u_xt=u.cuda())
  finally:
    ops.gemm_tn_splitk = orig
  assert len(calls) == 4 * m.n_blocks + 1
  ref_loss, ref_nlls, ref_g = orc.loss_and_grads(p, x0, attn, t, u, m.n_heads, m.n_blocks,
                                                 MASK_T, None, 0.0)
  assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item())
  g = tr.grads()
  worst = max((_rel(g[k], ref_g[k]), k) for k in ref_g)
  assert worst[0] < 6e-2, worst


@pytest.mark.parametrize('M,N,K', [(256, 3072, 768), (384, 520, 128)])
def test_gemm_gelu_fusions(ops, M, N, K):
  """mlp.0 forward writing pre-activation + GELU, and the GELU backward in the
  epilogue of the dX GEMM, against the separate kernels / torch autograd."""
  g = torch.Generator().manual_seed(N)
  A = _bf(torch.randn(M, K, generator=g)).cuda()
  W = _bf(torch.randn(N, K, generator=g) * 0.05).cuda()
  b = (torch.randn(N, generator=g) * 0.1).cuda()
  ld = (N + 7) // 8 * 8 + 8
  pre = torch.zeros(M, ld, dtype=torch.bfloat16, device='cuda')[:, :N]
  act = torch.zeros(M, ld, dtype=torch.bfloat16, device='cuda')[:, :N]
  ops.gemm_bf16_gelu(A, W, act, pre, bias=b)
  ref_pre = A.float() @ W.float().T + b
  assert _rel(pre.float(), ref_pre) < 1e-2
  assert _rel(act.float(), F.gelu(ref_pre, approximate='tanh')) < 1e-2
  # backward: C = bf16(A W^T) * gelu'(aux)
  x = (pre.float() * 3).requires_grad_(True)
  F.gelu(x, approximate='tanh').sum().backward()
  aux = _bf(x.detach())
  auxp = torch.zeros(M, ld, dtype=torch.bfloat16, device='cuda')[:, :N]
  auxp.copy_(aux)
  out = torch.zeros(M, ld, dtype=torch.bfloat16, device='cuda')[:, :N]
  ops.gemm_bf16_gelu(A, W, out, auxp, grad=True)
  xg = aux.float().requires_grad_(True)
  F.gelu(xg, approximate='tanh').sum().backward()
  ref = _bf(A.float() @ W.float().T).float() * xg.grad
  assert _rel(out.float(), ref) < 1e-2


def test_training_step_without_masked_tokens(ops):
  """q_xt may leave a micro-batch without a single masked position (small t):
  the loss is 0, every gradient is 0 and the optimizer step is a no-op on the weights."""
  model, cfg, p, tr = _make_trainer(0.1)
  B, L = 4, cfg.model.length
  x0, attn, t, _ = _batch(B, L, seed=41)
  u = torch.full((B, L), 0.9999)            # u >= move_chance everywhere: nothing is masked
  loss, nlls = tr.forward_backward(x0.cuda(), attn.cuda(), t=t.cuda() * 0.5, u_xt=u.cuda())
  assert loss.item() == 0.0 and float(nlls.abs().max()) == 0.0
  assert float(tr.G.abs().max()) == 0.0
  before = tr.P.clone()
  tr.step_count = 5                          # past the warm-up: lr > 0
  tr.optimizer_step()
  assert torch.equal(tr.P, before)


def test_trainer_checkpoint_roundtrip_and_resume(ops, tmp_path):
  """save -> load into a fresh Trainer -> one more step == the uninterrupted
  run (weights, Adam moments, EMA shadow, step count, Philox
  offset; the reference gets this from Lightning checkpoints + the `ema` entry of
  on_save_checkpoint / on_load_checkpoint, diffusion.py:168-181).  The file has
  the reference layout (`state_dict` with backbone.* keys + `ema`) and loads with
  weights_only=True."""
  B = 4
  model, cfg, p, tr = _make_trainer(0.1, extra=['training.ema=0.999'])
  L = cfg.model.length
  batches = [_batch(B, L, seed=s) for s in range(4)]

  def step(tr_, b):
    x0, attn, t, u = b
    tr_.forward_backward(x0.cuda(), attn.cuda(), t=t.cuda(), u_xt=u.cuda())
    tr_.optimizer_step()
  for b in batches[:3]:
    step(tr, b)
  path = str(tmp_path / 'checkpoints' / 'last.ckpt')
  tr.save_checkpoint(path)
  ckpt = torch.load(path, map_location='cpu', weights_only=True)
  assert set(ckpt) >= {'state_dict', 'ema', 'global_step', 'trainer_state'}
  assert all(k.startswith('backbone.') for k in ckpt['state_dict'])
  assert ckpt['global_step'] == 3 and ckpt['ema']['num_updates'] == 3
  assert len(ckpt['ema']['shadow_params']) == len(list(model.backbone.parameters()))
  step(tr, batches[3])   # the uninterrupted run

  model2, _, _, tr2 = _make_trainer(0.1, seed=11, extra=['training.ema=0.999'])
  assert tr2.load_checkpoint(path) == 3
  assert tr2.step_count == 3 and model2._rng_offset == ckpt['trainer_state']['rng_offset']
  step(tr2, batches[3])
  # equal up to the order of the embedding gradient's fp32 atomics (the only
  # non-deterministic sum of the step): everything else is bit-identical
  for name in ('P', 'M1', 'M2', 'E', 'W16'):
    a, b = getattr(tr, name).float(), getattr(tr2, name).float()
    assert torch.allclose(a, b, rtol=1e-3, atol=1e-6), (name, (a - b).abs().max().item())
    assert (a != b).float().mean().item() < 0.05, name
  assert tr2.step_count == tr.step_count == 4 and tr2.last_lr == tr.last_lr

  # the same file through the inference loader (main.py): weights + restored EMA,
  # and a Trainer built on that model keeps the restored EMA instead of resetting it
  import main as cli
  model3, _, _, _ = _make_trainer(0.1, seed=12, extra=['training.ema=0.999'])
  from mdlm_amd.config import load_config
  from mdlm_amd.diffusion import Diffusion
  from mdlm_amd.tokenizer import SyntheticTokenizer
  from mdlm_amd.train import Trainer
  m4 = Diffusion(load_config(['model=tiny', 'loader.eval_batch_size=4', 'training.ema=0.999']),
                 SyntheticTokenizer(vocab_size=V_T, eos_token_id=7, mask_token_id=MASK_T)).cuda()
  cli._load_backbone_weights(m4, path)
  assert m4._ema_restored and m4.ema.num_updates == 3
  tr4 = Trainer(m4, lr=3e-3, num_warmup_steps=2, dropout=0.1)
  for n, sp in zip(tr4.names, ckpt['ema']['shadow_params']):
    assert torch.equal(tr4._view(tr4.E, n).cpu(), sp), n
  # ... and CONTINUES it: the next update uses the decay of update 4 of
  # models/ema.py:40-45 (min(0.999, 5/14)), not a restarted warm-up (2/11) --
  # while the optimizer, which got no Adam moments from this loader, starts at 0
  assert tr4.ema_updates == 3 and tr4.step_count == 0
  e0 = tr4.E.clone()
  step(tr4, batches[3])
  d = min(0.999, (1 + 4) / (10 + 4))
  want = e0 - (1 - d) * (e0 - tr4.P)
  assert torch.allclose(tr4.E, want, rtol=1e-5, atol=1e-7), (tr4.E - want).abs().max().item()
  assert tr4.ema_updates == 4 and tr4.step_count == 1 and m4.ema.num_updates == 4
  assert tr4.ema_state_dict()['num_updates'] == 4

  # a weights + EMA file WITHOUT optimizer state through Trainer.load_checkpoint
  # (a reference checkpoint): only the EMA's counter carries over
  ref_like = {k: v for k, v in ckpt.items() if k != 'trainer_state'}
  ref_like['ema'] = dict(ckpt['ema'], num_updates=100000)
  path_ref = str(tmp_path / 'ref_like.ckpt')
  torch.save(ref_like, path_ref)
  model5, _, _, tr5 = _make_trainer(0.1, seed=13, extra=['training.ema=0.999'])
  step(tr5, batches[0])   # dirty Adam moments / counters that the load must reset
  assert tr5.load_checkpoint(path_ref) == 0
  assert tr5.step_count == 0 and tr5.ema_updates == 100000
  assert float(tr5.M1.abs().max()) == 0 and float(tr5.M2.abs().max()) == 0
  e0 = tr5.E.clone()
  step(tr5, batches[3])
  assert tr5.last_lr == 0.0      # optimizer step 0 of the warm-up: lr 0 (constant_warmup)
  want = e0 - (1 - 0.999) * (e0 - tr5.P)
  assert torch.allclose(tr5.E, want, rtol=1e-5, atol=1e-7)

  # every rank of a resumed DDP run loads rank 0's file; each must get back its
  # OWN Philox key (seed + rank), not rank 0's
  import os
  old = os.environ.get('RANK')
  os.environ['RANK'] = '1'
  try:
    model6, _, _, tr6 = _make_trainer(0.1, seed=11, extra=['training.ema=0.999'])
    tr6.load_checkpoint(path)
    assert model6._rng_seed == model2._rng_seed + 1
    assert model6._rng_offset == ckpt['trainer_state']['rng_offset']
  finally:
    if old is None:
      del os.environ['RANK']
    else:
      os.environ['RANK'] = old


def test_trainer_rejects_discrete_time_and_bad_ids(ops):
  """T > 0 would train on a different loss from validation (diffusion.py:849-853,
  329-358): refused.  Token ids outside the vocabulary raise (the reference's
  embedding lookup asserts) instead of being clamped silently."""
  with pytest.raises(NotImplementedError, match='T > 0'):
    _make_trainer(0.0, extra=['T=8'])
  model, cfg, p, tr = _make_trainer(0.0)
  x0, attn, t, u = _batch(4, cfg.model.length)
  x0[0, 3] = V_T + 5
  with pytest.raises(IndexError, match='token id out of range'):
    tr.forward_backward(x0.cuda(), attn.cuda(), t=t.cuda(), u_xt=u.cuda())
  with pytest.raises(IndexError, match='token id out of range'):
    model._loss(x0.cuda(), attn.cuda())
