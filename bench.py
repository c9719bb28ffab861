"""Benchmark of the MDLM ddpm_cache sampling hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full `restore_model_and_sample(num_steps=1000)` over one batch
(BASELINE.json configs[1]: DiT-small 169M, L=1024, T=1000, batch 64/GPU, bf16
operands, GPT-2-shaped vocab 50,258, seeded synthetic weights, all-mask prior).
Ranks are independent replicas (seed + rank); no data-path collective.  Rank 0
prints ONE JSON line: metric = ddpm_cache samples/sec (whole job).  The timed
region carries no instrumentation; one more batch after it is profiled with
HIP events on the launch stream around every kernel of every k-th denoising
step, giving `rooflines` (every loop kernel: algorithmic flops or bytes / launch
time / peak) and `roofline` (the dominant kernel by total time).  `cpu_baseline`
is the CPU oracle timed on this box's host cores on a bounded sample; `dist`
records the backend, the world size and an all-reduce check.
`python bench.py --gpus N` without torchrun starts its own N ranks.

The other workloads of BASELINE.json's config list ride in the same line as
sub-blocks, each with its own value / unit / config / dominant-kernel roofline:
`eval` (configs[2], NELBO tokens/s), `semi_ar` (configs[3], one batch of 2048-token
semi-AR sequences on the same model) and `train` (configs[4], 20 LM1B training
steps on its own model); `--workload train | semi_ar` runs one of them alone.  `MDLM_DIST_BACKEND=gloo` rehearses N ranks on a one-GPU box.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mdlm_amd import dist as mdist  # noqa: E402
from mdlm_amd.config import load_config  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def csrc_sha16():
  """sha256 (first 16 hex digits) over the kernel sources: stamps the PMC traffic
  figures (tools/pmc_traffic.py writes it, this file compares it), so that bytes
  measured on other kernels are never replayed as this tree's."""
  import glob
  import hashlib
  h = hashlib.sha256()
  d = os.path.join(ROOT, 'mdlm_amd', 'csrc')
  for f in sorted(glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, '*.h'))):
    h.update(os.path.basename(f).encode())
    h.update(open(f, 'rb').read())
  return h.hexdigest()[:16]


def load_traffic():
  """profiles/pmc_traffic.json if it was collected on THIS tree's kernels, else no
  per-kernel traffic (null in the rooflines) and the reason."""
  tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
  if not os.path.exists(tpath):
    return {}, {'file': None, 'used': False, 'why': 'profiles/pmc_traffic.json absent'}
  t = json.load(open(tpath))
  src = {'file': 'profiles/pmc_traffic.json', 'git_head': t.get('_git_head'),
         'collected': t.get('_date'), 'T': t.get('_T'), 'csrc_sha16': t.get('_csrc_sha16'),
         'tree_csrc_sha16': csrc_sha16()}
  src['used'] = src['csrc_sha16'] == src['tree_csrc_sha16']
  if not src['used']:
    src['why'] = 'kernel sources changed since the PMC passes: traffic dropped (null), not replayed'
    return {}, src
  return t, src


def synthetic_state_dict(model_cfg, vocab, seed=1):
  """Seeded N(0, s) weights with the reference state-dict layout,
  generated on the GPU (the oracle's CPU generator would take minutes at
  169M params and is not needed for throughput)."""
  g = torch.Generator(device='cuda').manual_seed(seed)
  D, C = model_cfg.hidden_size, model_cfg.cond_dim

  def rn(*shape, s=0.02):
    return torch.randn(*shape, generator=g, device='cuda') * s
  sd = {'vocab_embed.embedding': rn(vocab, D, s=0.15),
        'sigma_map.mlp.0.weight': rn(C, 256, s=0.05), 'sigma_map.mlp.0.bias': rn(C),
        'sigma_map.mlp.2.weight': rn(C, C, s=0.05), 'sigma_map.mlp.2.bias': rn(C)}
  for i in range(model_cfg.n_blocks):
    p = f'blocks.{i}.'
    sd[p + 'norm1.weight'] = 1 + rn(D)
    sd[p + 'attn_qkv.weight'] = rn(3 * D, D, s=0.04)
    sd[p + 'attn_out.weight'] = rn(D, D)
    sd[p + 'norm2.weight'] = 1 + rn(D)
    sd[p + 'mlp.0.weight'] = rn(4 * D, D); sd[p + 'mlp.0.bias'] = rn(4 * D)
    sd[p + 'mlp.2.weight'] = rn(D, 4 * D); sd[p + 'mlp.2.bias'] = rn(D)
    sd[p + 'adaLN_modulation.weight'] = rn(6 * D, C, s=0.05)
    sd[p + 'adaLN_modulation.bias'] = rn(6 * D, s=0.1)
  sd['output_layer.norm_final.weight'] = 1 + rn(D)
  sd['output_layer.linear.weight'] = rn(vocab, D, s=0.08)
  sd['output_layer.linear.bias'] = rn(vocab)
  sd['output_layer.adaLN_modulation.weight'] = rn(2 * D, C, s=0.05)
  sd['output_layer.adaLN_modulation.bias'] = rn(2 * D, s=0.1)
  return sd


def build_model(cfg, seed):
  from mdlm_amd.diffusion import Diffusion
  from mdlm_amd.tokenizer import get_tokenizer
  model = Diffusion(cfg, get_tokenizer(cfg)).cuda()
  sd = synthetic_state_dict(cfg.model, model.vocab_size, seed)
  sd['rotary_emb.inv_freq'] = model.backbone.rotary_emb.inv_freq
  model.backbone.load_state_dict(sd)
  model.ema = None  # synthetic weights: nothing to swap
  return model


def expected_nfe_b1(T, L, eps=1e-5, trials=64, seed=0):
  """Expected backbone calls at batch 1: a step is a cache miss iff the
  previous step unmasked a token; each masked token unmasks w.p. dt/t."""
  import numpy as np
  rng = np.random.default_rng(seed)
  ts = np.linspace(1, eps, T + 1)
  dt = (1 - eps) / T
  tot = 0
  for _ in range(trials):
    masked, miss, nfe = L, True, 0
    for i in range(T):
      nfe += miss
      k = rng.binomial(masked, min(1.0, dt / ts[i])) if masked else 0
      masked -= k
      miss = k > 0
    tot += nfe + 1  # + noise removal forward
  return tot / trials


def cpu_baseline(cfg, T, budget_s=20.0):
  """Oracle (CPU restatement of the reference, fp32, SDPA attention) on this
  box's host cores: BASELINE config 1 (B=1, L=1024, DiT-small)."""
  sys.path.insert(0, os.path.join(ROOT, 'oracle'))
  import mdlm_oracle as orc
  m = cfg.model
  V = 50258
  cores = len(os.sched_getaffinity(0))
  threads = min(cores, 16)
  torch.set_num_threads(threads)
  p = orc.synthetic_dit_params(m.hidden_size, m.n_heads, m.n_blocks, m.cond_dim, V, seed=1)
  od = orc.OracleDiffusion(p, m.n_heads, m.n_blocks, V, V - 1, m.length)
  L = m.length
  x = torch.full((1, L), V - 1, dtype=torch.int64)
  t = torch.ones(1, 1)
  dt = (1 - 1e-5) / T
  with torch.no_grad():
    od._ddpm_caching_update(x, t, dt)  # warm-up (page-in, thread pool)
    t0 = time.perf_counter(); n_miss = 0
    while n_miss < 2 or (time.perf_counter() - t0 < budget_s * 0.6 and n_miss < 8):
      p_x0, xn = od._ddpm_caching_update(x, t, dt); n_miss += 1
    t_miss = (time.perf_counter() - t0) / n_miss
    t0 = time.perf_counter(); n_hit = 0
    while n_hit < 2 or (time.perf_counter() - t0 < budget_s * 0.4 and n_hit < 16):
      od._ddpm_caching_update(x, t, dt, p_x0=p_x0); n_hit += 1
    t_hit = (time.perf_counter() - t0) / n_hit
  nfe = expected_nfe_b1(T, L)
  total = nfe * t_miss + (T + 1 - nfe) * t_hit
  return {
    'value': 1.0 / total, 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
    'sample': (f'oracle B=1 L={L} V={V}: {n_miss} cache-miss steps '
               f'({t_miss:.3f} s each) + {n_hit} cache-hit steps ({t_hit:.3f} s each), '
               f'extrapolated to T={T} with expected NFE {nfe:.0f}'),
  }


class OpTimer:
  """HIP events on the launch stream around every call of the public functions of
  `mdlm_amd.ops` (the training step's kernels), for ONE profiling step outside the
  timed region.  name -> [(start, end, work)], work = algorithmic flops of the
  GEMM / attention calls (from their operand shapes), else None."""

  GEMMS = {'gemm_bf16': lambda a, w, *r, **k: 2.0 * a.shape[0] * w.shape[0] * w.shape[1],
           'gemm_bf16_gelu': lambda a, w, *r, **k: 2.0 * a.shape[0] * w.shape[0] * w.shape[1],
           # dW = dY^T X over T tokens: (T, N) x (T, K)
           'gemm_tn_splitk': lambda dy, a, *r, **k: 2.0 * dy.shape[0] * dy.shape[1] * a.shape[1],
           'gemm_bf16_splitk': lambda a, w, *r, **k: 2.0 * a.shape[0] * w.shape[0] * a.shape[1]}

  def __init__(self):
    from mdlm_amd import ops
    self.ops, self.saved, self.events = ops, {}, {}

  def __enter__(self):
    import types
    for name, fn in vars(self.ops).items():
      if name.startswith('_') or not isinstance(fn, types.FunctionType) or fn.__module__ != self.ops.__name__:
        continue
      if name in ('workspace', 'workspace_bytes', 'check_token_ids', 'splitk_plan', 'tn_splitk_plan',
                  'padded_vocab', 'alloc_logits', 'gemm_set_tile'):
        continue
      self.saved[name] = fn
      setattr(self.ops, name, self._wrap(name, fn))
    return self

  def _wrap(self, name, fn):
    def timed(*a, **k):
      e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
      e0.record()
      r = fn(*a, **k)
      e1.record()
      work = self.GEMMS[name](*a, **k) if name in self.GEMMS else None
      self.events.setdefault(name, []).append((e0, e1, work))
      return r
    return timed

  def __exit__(self, *exc):
    for name, fn in self.saved.items():
      setattr(self.ops, name, fn)

  def summary(self, extra_work=None):
    """[{op, launches, total_ms, avg_launch_ms, (achieved, peak, frac)}], by total time."""
    torch.cuda.synchronize()
    out = []
    for name, evs in self.events.items():
      ms = [a.elapsed_time(b) for a, b, _ in evs]
      d = {'op': name, 'launches': len(ms), 'total_ms': round(sum(ms), 3),
           'avg_launch_ms': round(sum(ms) / len(ms), 4)}
      work = [w for _, _, w in evs]
      if extra_work and name in extra_work:
        work = [extra_work[name]] * len(evs)
      if all(w is not None for w in work) and sum(ms) > 0:
        ach = sum(work) / (sum(ms) * 1e-3) / 1e12
        d.update(bound='mfma', achieved=round(ach, 1), peak=PEAK_BF16_TFLOPS, unit='TFLOP/s',
                 frac=round(ach / PEAK_BF16_TFLOPS, 4),
                 algorithmic_gflop_per_launch=round(sum(work) / len(work) / 1e9, 2))
      out.append(d)
    return sorted(out, key=lambda d: -d['total_ms'])


def train_block(steps, warmup, global_batch, rank, world, dev, profile=True):
  """BASELINE.json configs[4]: LM1B MDLM training step, DiT-small, L=128, global
  batch 512 split over the ranks (strong scaling), bert-base-uncased-shaped
  vocabulary (30,522, [MASK]=103 in-vocabulary), dropout 0.1, clip 1.0, AdamW,
  EMA, DDP gradient all-reduce on RCCL overlapped with the backward.  A step =
  forward + backward + all-reduce + optimizer/EMA over one global batch."""
  from mdlm_amd.train import Trainer
  cfg = load_config(['model=small', 'data=lm1b', 'model.length=128',
                     f'loader.global_batch_size={global_batch}', f'trainer.devices={world}'])
  model = build_model(cfg, seed=1)   # same weights on every rank
  model.seed_rng(mdist.rank_seed(cfg.seed, rank))
  tr = Trainer(model, num_warmup_steps=cfg.lr_scheduler.num_warmup_steps)
  B, L, V = cfg.loader.batch_size, cfg.model.length, model.vocab_size
  g = torch.Generator(device='cuda').manual_seed(1 + rank)
  ids = torch.randint(0, V, (B, L), generator=g, device='cuda')
  ids[ids == model.mask_index] = 0
  ids[:, 0] = 101; ids[:, -1] = 102
  batch = {'input_ids': ids, 'attention_mask': torch.ones(B, L, device='cuda')}
  torch.manual_seed(1 + rank)
  for _ in range(max(1, warmup)):
    loss = tr.training_step(batch)
  torch.cuda.synchronize(); mdist.barrier(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    loss = tr.training_step(batch)
  torch.cuda.synchronize(); mdist.barrier(); torch.cuda.synchronize()
  el = mdist.max_over_ranks(time.perf_counter() - t0, dev)
  n_par = sum(p.numel() for p in model.backbone.parameters())
  # forward flops per token: blocks 2*(3+1+8)*D^2 + attention 4*L*D, vocab 2*D*V; x3 for fwd+bwd
  D, nb, H = cfg.model.hidden_size, cfg.model.n_blocks, cfg.model.n_heads
  f_tok = nb * (24 * D * D + 4 * L * D) + 2 * D * V
  out = {'metric': 'LM1B MDLM training step tokens/sec (DiT-small, L=128, global batch 512)',
         'value': round(world * B * L * steps / el, 1), 'unit': 'tokens/s', 'n_gpus': world,
         'steps': steps, 'warmup': max(1, warmup), 'ms_per_step': round(el / steps * 1e3, 2),
         'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'bf16',
         'data': 'synthetic',
         'config': {'workload': f'LM1B MDLM training step, DiT-small ({n_par/1e6:.1f}M), L={L}, '
                                f'global batch {world * B}, V={V}, dropout 0.1, clip 1.0, AdamW+EMA',
                    'batch_per_gpu': B, 'parallelism': f'ddp x{world}',
                    'grad_allreduce_mbytes': round(tr.n_flat * 4 / 1e6, 1)},
         'loss': round(float(loss), 4),
         'model_tflops': round(3 * f_tok * world * B * L * steps / el / 1e12, 1)}
  if profile:
    # one more step with HIP events around every kernel call (outside the timed region)
    with OpTimer() as ot:
      tr.training_step(batch)
    # attention: forward 4 L^2 d per (sequence, head); backward 5 matmuls = 10 L^2 d
    att = {'attn_fwd_lse': 4.0 * L * L * 64 * B * H, 'attn_bwd': 10.0 * L * L * 64 * B * H}
    ks = ot.summary(att)
    mf = [k for k in ks if k.get('bound') == 'mfma']
    if mf:
      out['roofline'] = dict(mf[0], kernel={'gemm_tn_splitk': 'gemm_tn_splitk_kernel (dW = dY^T X, token-major, + splitk_reduce)',
                                            'gemm_bf16': 'gemm_bf16_kernel<STORE / QKV_ROTARY> (forward + dX GEMMs)',
                                            'attn_bwd': 'attn_bwd_l128_kernel'}.get(mf[0]['op'], mf[0]['op']))
    out['kernels'] = ks[:12]
  del tr, model
  torch.cuda.empty_cache()
  return out


def train_bench(args):
  rank, local_rank, world = mdist.init()
  torch.cuda.set_device(mdist.local_device(local_rank))
  dev = torch.device('cuda', mdist.local_device(local_rank))
  out = train_block(args.steps if args.steps > 2 else 20, args.warmup, args.global_batch,
                    rank, world, dev)
  out['dist'] = dist_block(world, dev)
  if rank == 0:
    print(json.dumps(out), flush=True)
  mdist.barrier()


def semi_ar_block(model, cfg, T, steps, warmup, rank, world, dev, profile_stride=8):
  """BASELINE.json configs[3]: semi-autoregressive generation (README recipe:
  L=1024 windows, stride 512, 2 extra strides -> 2048 tokens per sequence,
  dt = 1/1000), batch per GPU = loader.eval_batch_size, replicas sharded by sample
  (no collective).  A step = one restore_model_and_semi_ar_sample over one batch."""
  B, L = cfg.loader.eval_batch_size, cfg.model.length
  stride, n_str = 512, 2
  run = lambda ns=n_str: model.restore_model_and_semi_ar_sample(
    stride_length=stride, num_strides=ns, dt=1 / T)
  for _ in range(warmup):
    run()
  torch.cuda.synchronize(); mdist.barrier(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    nfe_steps, _, _ = run()
  torch.cuda.synchronize(); mdist.barrier(); torch.cuda.synchronize()
  el = mdist.max_over_ranks(time.perf_counter() - t0, dev)
  n_tok = L + n_str * stride
  out = {'metric': 'semi-AR generation sequences/sec (2048 tokens, stride 512, T=1000 per window)',
         'value': round(world * B * steps / el, 4), 'unit': 'sequences/s', 'n_gpus': world,
         'steps': steps, 'warmup': warmup, 'ms_per_step': round(el / steps * 1e3, 1),
         'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16',
         'data': 'synthetic',
         'config': {'workload': f'semi-AR sample_subs_guidance, DiT-small, window L={L}, stride {stride}, '
                                f'num_strides {n_str} -> {n_tok} tokens, dt=1/{T}, batch={B}/GPU',
                    'parallelism': f'replicas x{world}'},
         'tokens_per_sec': round(world * B * n_tok * steps / el, 1),
         'sec_per_sequence_at_this_batch': round(el / steps / B, 4),
         'steps_with_change': int(nfe_steps)}
  if profile_stride > 0:
    # profiling pass over ONE window (num_strides = 0), outside the timed region
    global D_MODEL
    D_MODEL = cfg.model.hidden_size
    traffic, _ = load_traffic()
    bb = model.backbone
    bb.profile_events = {}
    model._profile = dict(stride=profile_stride, rows=[])
    run(0)
    torch.cuda.synchronize()
    events = bb.profile_events
    bb.profile_events, model._profile = None, None
    rl = kernel_rooflines(events, traffic)
    dom = max((d for t, d in rl.items() if d['bound'] != 'launch' and t not in ('gemm_out', 'gemm_fc2')),
              key=lambda d: d['total_ms'])
    out['roofline'] = dom
  return out


def semi_ar_bench(args):
  rank, local_rank, world = mdist.init()
  torch.cuda.set_device(mdist.local_device(local_rank))
  dev = torch.device('cuda', mdist.local_device(local_rank))
  cfg = load_config(['model=small', f'loader.eval_batch_size={args.batch}',
                     f'sampling.steps={args.T}', 'sampling.semi_ar=True',
                     'sampling.stride_length=512', 'sampling.num_strides=2',
                     'sampling.fast_math=true', f'trainer.devices={world}'])
  model = build_model(cfg, seed=1)
  model.seed_rng(mdist.rank_seed(cfg.seed, rank))
  out = semi_ar_block(model, cfg, args.T, args.steps, args.warmup, rank, world, dev,
                      profile_stride=args.profile_stride)
  out['dist'] = dist_block(world, dev)
  if rank == 0:
    print(json.dumps(out), flush=True)
  mdist.barrier()


def update_kernel_block(model, B, L, V):
  """The fused update kernel alone (SUBS + race from cached logits, every row masked,
  early step) on two synthetic logit distributions: std 2.2 = what the bench's random
  output layer produces; std 8 = a PEAKED distribution, what a trained checkpoint's
  logits look like -- there the fast race proves for most 8-column vectors that no draw
  can reach the mask key and skips their random numbers (same token ids).  Labelled
  secondary figures: the headline runs on the synthetic weights."""
  from mdlm_amd import ops
  st = model._sampler_state(B, L)
  cache, lse = st['cache'], st['lse']
  x = torch.full((B, L), model.mask_index, dtype=torch.int64, device='cuda')
  t = torch.full((B,), 0.9, device='cuda')
  dt = (1 - 1e-5) / 1000
  out = {}
  for name, std in (('near_uniform_std2.2', 2.2), ('peaked_std8', 8.0)):
    cache.normal_(0, std)
    ops.subs_row_lse(cache, x, model.mask_index, out=lse)
    run = lambda k: ops.subs_ddpm_update(cache, lse, x, t, dt, model.mask_index, seed=3, offset=k,
                                         x_next=st['xb'], changed=st['changed'], fast_math=True)
    for k in range(2):
      run(k)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(10):
      run(10 + k)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gbs = B * L * V * 2.0 / (ms * 1e-3) / 1e9
    out[name] = {'ms': round(ms, 4), 'achieved': round(gbs, 1), 'unit': 'GB/s', 'peak': PEAK_HBM_GBS,
                 'frac': round(gbs / PEAK_HBM_GBS, 4)}
  st['changed'].fill_(1)
  return {'kernel': 'ddpm_update_fast_kernel, 64 x 1024 masked rows x 50,258 bf16 logits, t = 0.9',
          **out}


def dist_block(world, dev):
  """What proves the collectives saw N ranks: backend, world size and a SUM
  all-reduce of (rank + 1) checked against N (N + 1) / 2 on every rank."""
  import torch.distributed as tdist
  if world == 1:
    return {'backend': None, 'world_size': 1, 'allreduce_check': True}
  t = torch.tensor([float(mdist.rank() + 1)], dtype=torch.float64, device=dev)
  mdist.all_reduce_sums(t)
  return {'backend': tdist.get_backend(), 'world_size': tdist.get_world_size(),
          'allreduce_check': bool(t.item() == world * (world + 1) / 2)}


def self_launch(args):
  """`python bench.py --gpus N` without torchrun: the parent -- before anything
  touches a GPU -- starts N fresh child processes (one rank per GPU) with
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, relays rank 0's JSON line and
  returns the worst exit code.  A stalled child dumps its Python stacks and
  exits non-zero by itself (faulthandler, armed in main())."""
  import socket
  import subprocess
  n = args.gpus
  if os.environ.get('MDLM_DIST_BACKEND', 'nccl') == 'nccl' and args.workload != 'dist_check':
    have = torch.cuda.device_count()   # counting devices does not initialise HIP
    if have < n:
      raise SystemExit(f'--gpus {n}: {have} GPU(s) visible (one RCCL rank per GPU; '
                       'MDLM_DIST_BACKEND=gloo rehearses N ranks on fewer cards)')
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  procs = []
  for r in range(n):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
               LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    procs.append(subprocess.Popen(
      [sys.executable, '-u', os.path.abspath(__file__)] + sys.argv[1:], env=env,
      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
  rc = 0
  try:
    for line in procs[0].stdout:     # rank 0 prints the one JSON line
      sys.stdout.write(line)
      sys.stdout.flush()
    for p in procs:
      rc = max(rc, abs(p.wait()))
  finally:
    for p in procs:                  # exact PIDs of our own children only
      if p.poll() is None:
        p.terminate()
  return rc


def dist_check_bench(args):
  """Launcher + collective rehearsal without compute (runs on CPU over gloo as
  well): init, barrier, all-reduce check, max-over-ranks; one JSON line."""
  rank, local_rank, world = mdist.init()
  dev = torch.device('cpu')
  if torch.cuda.is_available():
    torch.cuda.set_device(mdist.local_device(local_rank))
    dev = torch.device('cuda', mdist.local_device(local_rank))
  mdist.barrier()
  t0 = time.perf_counter()
  d = dist_block(world, dev)
  el = mdist.max_over_ranks(time.perf_counter() - t0, dev)
  if rank == 0:
    print(json.dumps({'metric': 'dist_check', 'n_gpus': world, 'dist': d,
                      'allreduce_s': round(el, 6)}), flush=True)
  mdist.barrier()
  if not d['allreduce_check']:
    raise SystemExit(3)


# per-kernel roofline bookkeeping of the sampling loop: tag -> (kernel symbol, bound)
KERNELS = {
  # 154 flop/byte: below the machine balance (2.5 PF / 8 TB/s = 312), so HBM is its roofline
  'gemm_out': ('gemm_bf16_kernel<EPI_GATE_RESIDUAL,fp32> attn_out 768->768', 'hbm'),
  'gemm_fc2': ('gemm_bf16_kernel<EPI_GATE_RESIDUAL,fp32> mlp.2 3072->768', 'mfma'),
  'gemm_fc1': ('gemm_bf16_kernel<EPI_BIAS_GELU,bf16> mlp.0 768->3072', 'mfma'),
  'gemm_qkv': ('gemm_bf16_kernel<EPI_QKV_ROTARY,bf16> 768->2304 + rotary', 'mfma'),
  'gemm_vocab': ('gemm_bf16_kernel<EPI_STORE,bf16> vocab projection 768->V, masked rows', 'mfma'),
  'attention': ('attn_fwd_pipe_kernel', 'mfma'),
  'update': ('ddpm_update_fast_kernel (SUBS + race fused from logits)', 'hbm'),
  'lse_combine': ('lse_combine_kernel', 'hbm'),
  'ln_modulate': ('ln_modulate_kernel', 'hbm'),
  'embed': ('embed_rows_kernel', 'hbm'),
  'gather_rows': ('gather_rows_kernel', 'hbm'),
  'row_lists': ('compact_active_kernel + masked_row_map_kernel', 'launch'),
  'gemm_gate_residual': ('gemm_bf16_kernel<EPI_GATE_RESIDUAL,fp32,256x256> (attn_out 768->768 and '
                         'mlp.2 3072->768: one symbol in the rocprof trace)', 'mfma'),
}


D_MODEL = 768   # set from the config in main()


def kernel_rooflines(events, traffic):
  """tag -> roofline dict from the HIP-event pairs of the profiling batch:
  achieved = sum(algorithmic work) / sum(launch time)."""
  vals = {}

  def rows_of(r):
    if torch.is_tensor(r):
      k = (r.data_ptr(), r._version)
      if k not in vals:
        vals[k] = int(r)
      return vals[k]
    return r
  out = {}
  events = dict(events)
  # one kernel symbol in a rocprof trace: both gate/residual GEMM shapes together
  events['gemm_gate_residual'] = events.get('gemm_out', []) + events.get('gemm_fc2', [])
  for tag, evs in events.items():
    if not evs:
      continue
    ms = [s.elapsed_time(e) for s, e, _, _ in evs]
    n = max(1, len(ms))
    rows = [rows_of(r) for _, _, r, _ in evs]
    work = sum(w * r for (_, _, _, w), r in zip(evs, rows))
    t = sum(ms) * 1e-3
    kernel, bound = KERNELS.get(tag, (tag, 'hbm'))
    if tag == 'gemm_out':
      # events carry flops per row (2 D^2); as an HBM-bound kernel it is priced on
      # its bytes: A bf16 (2 D) + residual read and fp32 C write (4 D each) per row
      work = work / (2.0 * D_MODEL * D_MODEL) * (10.0 * D_MODEL)
    d = {'tag': tag, 'kernel': kernel, 'bound': bound, 'launches': len(ms),
         'avg_launch_ms': round(sum(ms) / n, 4), 'total_ms': round(sum(ms), 2),
         'avg_rows_per_launch': round(sum(rows) / n, 1)}
    if bound == 'mfma':
      ach = work / t / 1e12 if t > 0 else 0.0
      d.update(achieved=round(ach, 1), peak=PEAK_BF16_TFLOPS, unit='TFLOP/s',
               frac=round(ach / PEAK_BF16_TFLOPS, 4),
               algorithmic_gflop_per_launch=round(work / n / 1e9, 2))
    elif bound == 'hbm':
      ach = work / t / 1e9 if t > 0 else 0.0
      d.update(achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit='GB/s',
               frac=round(ach / PEAK_HBM_GBS, 4),
               algorithmic_mbytes_per_launch=round(work / n / 1e6, 2))
    tr = traffic.get(tag)   # keys of profiles/pmc_traffic.json = these tags (tools/pmc_traffic.py)
    d['traffic'] = (round(tr['hbm_bytes_per_launch']
                          / tr.get('rows_per_launch', traffic['_rows_per_launch_profiled'])
                          * sum(rows) / n) if tr else None)
    if tr:
      d['traffic_unit'] = ('HBM bytes/launch: PMC FETCH_SIZE x2 + WRITE_SIZE from separate rocprofv3 '
                           '--pmc passes (profiles/pmc_traffic.json), scaled by live rows')
    out[tag] = d
  return out


def start_heartbeat(period_s=30.0):
  """A line on STDERR every `period_s` seconds for as long as the process lives, so that
  a harness that takes a silent process for a hung one sees the long timed region
  (minutes of asynchronous launches, no output by design) as alive.  A daemon thread
  that only sleeps and writes: it touches neither the GPU nor the timed loop."""
  import threading
  t0 = time.time()

  def beat():
    while True:
      time.sleep(period_s)
      print(f'[bench] running, {time.time() - t0:.0f} s', file=sys.stderr, flush=True)
  threading.Thread(target=beat, daemon=True).start()


class BlockGuard:
  """Runs the secondary blocks of the line (see main): an exception is recorded as the
  block's {"error": ...}; a block still running after its limit ends the process on every
  rank (each rank runs the same guard with the same limits), rank 0 first printing the
  line with everything measured so far -- exit code 0, the headline is valid."""

  def __init__(self, out, rank):
    self.out, self.rank = out, rank

  def _expire(self, name, limit_s):
    self.out[name] = {'error': f'no result after {limit_s} s: block abandoned, the line carries what was '
                               'measured before it'}
    if self.rank == 0:
      print(json.dumps(self.out), flush=True)
    sys.stdout.flush()
    os._exit(0)

  def run(self, name, limit_s, fn):
    import threading
    timer = threading.Timer(limit_s, self._expire, args=(name, limit_s))
    timer.daemon = True
    timer.start()
    try:
      fn()
    except Exception as e:   # noqa: BLE001 -- anything: the headline must survive
      self.out[name] = {'error': f'{type(e).__name__}: {e}'[:400]}
      print(f'[bench] block {name} failed: {type(e).__name__}: {e}', file=sys.stderr, flush=True)
    finally:
      timer.cancel()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--workload', default='sample',
                  choices=['sample', 'train', 'semi_ar', 'dist_check'])
  ap.add_argument('--global-batch', type=int, default=512)
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=2)
  ap.add_argument('--warmup', type=int, default=1)
  ap.add_argument('--batch', type=int, default=64)
  ap.add_argument('--T', type=int, default=1000)
  ap.add_argument('--model', default='small')
  ap.add_argument('--cache-policy', default='sample', choices=['sample', 'batch', 'none'])
  ap.add_argument('--race', default='fast', choices=['fast', 'reference'],
                  help='fast: approximate ranking + exact re-evaluation of near-ties; '
                       'reference: every element in the defining arithmetic.  Same tokens.')
  ap.add_argument('--profile-stride', type=int, default=8,
                  help='profiling batch (after the timed region): HIP events around '
                       'every kernel of each k-th denoising step; 0 = skip')
  ap.add_argument('--set', action='append', default=[], metavar='KEY=VALUE',
                  help='extra config override (same-box ablations), e.g. sampling.lse_from_gemm=false')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-policy-compare', action='store_true')
  ap.add_argument('--no-eval', action='store_true')
  ap.add_argument('--no-semi-ar', action='store_true')
  ap.add_argument('--no-train', action='store_true')
  args = ap.parse_args()
  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    raise SystemExit(self_launch(args))
  world_env = int(os.environ.get('WORLD_SIZE', 1))
  if int(os.environ.get('RANK', 0)) == 0:
    start_heartbeat()
  # a stuck rank dumps its Python stacks and exits non-zero (always armed for N > 1)
  dump_s = os.environ.get('MDLM_FAULT_DUMP_S') or ('1500' if world_env > 1 else '')
  if dump_s:
    import faulthandler
    faulthandler.dump_traceback_later(int(dump_s), exit=True)
  if args.workload == 'dist_check':
    return dist_check_bench(args)
  if args.workload == 'train':
    return train_bench(args)
  if args.workload == 'semi_ar':
    return semi_ar_bench(args)

  rank, local_rank, world = mdist.init()
  if world != args.gpus:
    raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
  torch.cuda.set_device(mdist.local_device(local_rank))
  dev = torch.device('cuda', mdist.local_device(local_rank))

  fast = args.race == 'fast'
  cfg = load_config([f'model={args.model}', f'loader.eval_batch_size={args.batch}',
                     f'sampling.steps={args.T}', f'sampling.cache_policy={args.cache_policy}',
                     f'sampling.fast_math={str(fast).lower()}', f'trainer.devices={world}',
                     *args.set])
  model = build_model(cfg, seed=1)
  model.seed_rng(mdist.rank_seed(cfg.seed, rank))
  B, L, V = args.batch, cfg.model.length, model.vocab_size
  D, nb = cfg.model.hidden_size, cfg.model.n_blocks
  global D_MODEL
  D_MODEL = D
  bb = model.backbone

  def sync_all():
    torch.cuda.synchronize()
    mdist.barrier()
    torch.cuda.synchronize()

  dist_info = dist_block(world, dev)
  for _ in range(args.warmup):
    model.restore_model_and_sample(num_steps=args.T)
  sync_all()
  # ---- timed region: no instrumentation of any kind ---------------------------
  t0 = time.perf_counter()
  for _ in range(args.steps):
    model.restore_model_and_sample(num_steps=args.T)
  sync_all()
  elapsed = time.perf_counter() - t0
  elapsed = mdist.max_over_ranks(elapsed, dev)
  nfe_last = int(model.last_nfe.item())
  n_exact = int(model.last_n_exact.item())
  n_groups = len((model._state or {}).get('groups') or [None])

  # ---- profiling batch (outside the timed region): HIP events on the launch
  # stream around every kernel of each k-th step, live row counts of every step
  rooflines, executed = {}, None
  if args.profile_stride > 0:
    # HBM bytes per launch from separate rocprofv3 --pmc passes, if taken on these kernels
    traffic, traffic_source = load_traffic()
    bb.profile_events = {}
    model._profile = dict(stride=args.profile_stride, rows=[])
    model.restore_model_and_sample(num_steps=args.T)
    torch.cuda.synchronize()
    events, rows = bb.profile_events, model._profile['rows']
    bb.profile_events, model._profile = None, None
    rooflines = kernel_rooflines(events, traffic)
    # executed flops: full-width blocks on the active rows; the last block's tail
    # and the vocabulary projection on the masked rows only
    n_rows = sum(int(a) for a, _ in rows)
    n_mrows = sum(int(b) for _, b in rows)
    f_head = 2.0 * D * 3 * D + 4.0 * L * D            # QKV + attention, per row
    f_tail = 2.0 * D * D + 16.0 * D * D                # out-proj + MLP, per row
    nfe_prof = int(model.last_nfe.item())
    executed = ((nb - 1) * (f_head + f_tail) * n_rows + f_head * n_rows
                + (f_tail + 2.0 * D * V) * n_mrows) * (nfe_last / max(1, nfe_prof))

  samples = world * B * args.steps
  value = samples / elapsed
  out = {
    'metric': 'ddpm_cache samples/sec (L=1024, T=1000)',
    'value': round(value, 4), 'unit': 'samples/s', 'n_gpus': world,
    'steps': args.steps, 'warmup': args.warmup,
    'ms_per_step': round(elapsed / args.steps * 1e3, 2),
    'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
    'dtype': 'bf16', 'data': 'synthetic',
    'config': {'workload': (f'mdlm-owt DiT-small ddpm_cache sampling, L={L}, T={args.T}, '
                            f'batch={B}/GPU, bf16, V={V}, noise_removal'),
               'cache_policy': args.cache_policy,
               'race': ('bit-exact; approximate ranking + exact re-evaluation of near-ties '
                        '(sampling.fast_math=true, the default)' if fast else
                        'bit-exact; every element in the defining arithmetic '
                        '(sampling.fast_math=false)'),
               'rng': 'in-kernel Philox4x32-7',
               'stream_groups': n_groups,
               'parallelism': f'replicas x{world}',
               **({'overrides': args.set} if args.set else {})},
    'dist': dist_info,
    'nfe_per_sample': round(nfe_last / B, 1),
    'race_rows_re_evaluated_exactly': n_exact,
    'ms_per_denoise_step': round(elapsed / args.steps / (args.T + 1) * 1e3, 3),
  }
  if rooflines:
    # dominant kernel = largest total time in the profiling batch
    # (by kernel symbol: the two gate/residual shapes count as one kernel)
    dom = max((d for t, d in rooflines.items()
               if d['bound'] != 'launch' and t not in ('gemm_out', 'gemm_fc2')),
              key=lambda d: d['total_ms'])
    out['roofline'] = dict(dom, traffic_source=traffic_source)
    out['rooflines'] = sorted((d for t, d in rooflines.items() if t != 'gemm_gate_residual'),
                              key=lambda d: -d['total_ms'])
    out['profile'] = {'stride': args.profile_stride,
                      'note': ('separate batch after the timed region; HIP events on the launch '
                               'stream around every kernel launch of each k-th denoising step; '
                               'the whole batch on ONE stream so that every kernel is timed alone '
                               '(the timed region overlaps the forwards of config.stream_groups '
                               'sample ranges on as many streams)')}
  # model flops charge every sample-forward 291.65 GFLOP; executed
  # flops count the vocabulary projection and the last block's tail on masked rows only
  out['model_tflops'] = round((nfe_last * 291.65e9 * args.steps) / elapsed / 1e12, 1)
  if executed is not None:
    out['executed_tflops'] = round(executed * args.steps / elapsed / 1e12, 1)

  # ---- secondary blocks.  The headline above is complete; nothing below may cost it: an
  # exception in a block becomes that block's {"error": ...}, and a block that produces
  # nothing within its time limit (a collective that never returns on some node) makes
  # every rank leave at once, rank 0 printing the line as it stands.
  guard = BlockGuard(out, rank)

  def eval_blk():
    # second half of the metric: NELBO eval tokens/s (q_xt + backbone + fused
    # SUBS-NLL + fp64 sums; sequences sharded by rank, one all-reduce at the end)
    eb = 64
    g = torch.Generator(device='cuda').manual_seed(1 + rank)
    ids = torch.randint(0, V - 1, (eb, L), generator=g, device='cuda')
    ids[:, 0] = 50256; ids[:, -1] = 50256
    batch = {'input_ids': ids, 'attention_mask': torch.ones(eb, L, device='cuda')}
    model.on_validation_epoch_start()
    model.validation_step(batch, 0)
    sync_all()
    model.valid_metrics.reset()
    n_eval = 8
    te = time.perf_counter()
    for i in range(n_eval):
      model.validation_step(batch, i)
    res = model.valid_metrics.compute(sync_dist=True)
    torch.cuda.synchronize()
    te = mdist.max_over_ranks(time.perf_counter() - te, dev)
    out['eval'] = {'metric': 'NELBO eval tokens/sec', 'value': round(world * n_eval * eb * L / te, 1),
                   'unit': 'tokens/s', 'batch_per_gpu': eb, 'batches': n_eval,
                   'val_nll': round(res['val/nll'], 4)}
  if not args.no_eval:
    guard.run('eval', 300, eval_blk)

  def semi_ar_blk():
    # BASELINE configs[3] on the same model and sampler state (already warm): one batch
    out['semi_ar'] = semi_ar_block(model, cfg, args.T, 1, 0, rank, world, dev,
                                   profile_stride=args.profile_stride)
  if not args.no_semi_ar and args.cache_policy == 'sample' and not args.set:
    guard.run('semi_ar', 600, semi_ar_blk)

  def ref_policy_blk():
    # the reference's literal all-or-nothing cache policy, for the record
    cfg.sampling.cache_policy = 'batch'
    try:
      torch.cuda.synchronize(); t1 = time.perf_counter()
      model.restore_model_and_sample(num_steps=args.T)
      torch.cuda.synchronize(); tb = time.perf_counter() - t1
      out['reference_cache_policy'] = {
        'policy': 'batch (all-or-nothing, diffusion.py:682-685)',
        'value': round(B / tb, 4), 'unit': 'samples/s',
        'nfe_per_sample': round(int(model.last_nfe.item()) / B, 1)}
    finally:
      cfg.sampling.cache_policy = args.cache_policy

  def lazy_blk():
    # opt-in two-stage step: same distribution as the reference step, not
    # bit-identical to it on a shared uniform tensor -> never the headline value
    cfg.sampling.lazy_unmask = True
    try:
      model.restore_model_and_sample(num_steps=args.T)   # warm-up (state rebuild)
      torch.cuda.synchronize(); t1 = time.perf_counter()
      model.restore_model_and_sample(num_steps=args.T)
      torch.cuda.synchronize(); tb = time.perf_counter() - t1
      out['lazy_unmask_mode'] = {
        'note': ('sampling.lazy_unmask=true: Bernoulli unmask decision, then a categorical draw and the '
                 'vocabulary projection for the unmasking rows only; distribution-identical to '
                 'diffusion.py:592-610, NOT bit-identical to the reference race on a shared uniform tensor'),
        'value': round(B / tb, 4), 'unit': 'samples/s',
        'nfe_per_sample': round(int(model.last_nfe.item()) / B, 1)}
    finally:
      cfg.sampling.lazy_unmask = False

  def update_blk():
    out['update_kernel'] = update_kernel_block(model, B, L, V)

  def peaked_blk():
    # the same sampling workload on a PEAKED output layer (output weights x 3.6: logit std ~ 8, what a
    # trained checkpoint's logits look like), where the fast race skips most of the vocabulary's random
    # numbers.  A labelled secondary figure: the headline `value` is measured on the BASELINE weights.
    w = model.backbone.output_layer.linear.weight
    with torch.no_grad():
      w.mul_(3.6)
    model.backbone.refresh()
    try:
      torch.cuda.synchronize(); t1 = time.perf_counter()
      model.restore_model_and_sample(num_steps=args.T)
      torch.cuda.synchronize(); tb = time.perf_counter() - t1
      out['peaked_output_layer'] = {
        'note': 'same workload, output_layer.linear.weight x 3.6 (logit std ~ 8); one batch, state already warm',
        'value': round(B / tb, 4), 'unit': 'samples/s',
        'nfe_per_sample': round(int(model.last_nfe.item()) / B, 1)}
    finally:
      with torch.no_grad():
        w.div_(3.6)
      model.backbone.refresh()

  if rank == 0 and world == 1 and not args.no_policy_compare:
    if args.cache_policy == 'sample':
      guard.run('reference_cache_policy', 600, ref_policy_blk)
      guard.run('lazy_unmask_mode', 600, lazy_blk)
    if B * L * V < (1 << 33):
      guard.run('update_kernel', 300, update_blk)
      guard.run('peaked_output_layer', 600, peaked_blk)
  del model
  torch.cuda.empty_cache()

  def train_blk():
    # BASELINE configs[4]: 20 LM1B training steps (its own model: V = 30,522, L = 128)
    out['train'] = train_block(20, 3, args.global_batch, rank, world, dev,
                               profile=args.profile_stride > 0)
  if not args.no_train:
    guard.run('train', 400, train_blk)

  def cpu_blk():
    out['cpu_baseline'] = cpu_baseline(cfg, args.T)
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    guard.run('cpu_baseline', 900, cpu_blk)
  if rank == 0:
    print(json.dumps(out), flush=True)
  mdist.barrier()


def semi_ar_block(model, cfg, T, steps, warmup, rank, world, dev, profile_stride=8):
  """BASELINE.json configs[3]: semi-autoregressive generation (README recipe:
  L=1024 windows, stride 512, 2 extra strides -> 2048 tokens per sequence,
  dt = 1/1000), batch per GPU = loader.eval_batch_size, replicas sharded by sample
  (no collective).  A step = one restore_model_and_semi_ar_sample over one batch."""
  B, L = cfg.loader.eval_batch_size, cfg.model.length
  stride, n_str = 512, 2
  run = lambda ns=n_str: model.restore_model_and_semi_ar_sample(
    stride_length=stride, num_strides=ns, dt=1 / T)
  for _ in range(warmup):
    run()
  torch.cuda.synchronize(); mdist.barrier(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    nfe_steps, _, _ = run()
  torch.cuda.synchronize(); mdist.barrier(); torch.cuda.synchronize()
  el = mdist.max_over_ranks(time.perf_counter() - t0, dev)
  n_tok = L + n_str * stride
  out = {'metric': 'semi-AR generation sequences/sec (2048 tokens, stride 512, T=1000 per window)',
         'value': round(world * B * steps / el, 4), 'unit': 'sequences/s', 'n_gpus': world,
         'steps': steps, 'warmup': warmup, 'ms_per_step': round(el / steps * 1e3, 1),
         'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16',
         'data': 'synthetic',
         'config': {'workload': f'semi-AR sample_subs_guidance, DiT-small, window L={L}, stride {stride}, '
                                f'num_strides {n_str} -> {n_tok} tokens, dt=1/{T}, batch={B}/GPU',
                    'parallelism': f'replicas x{world}'},
         'tokens_per_sec': round(world * B * n_tok * steps / el, 1),
         'sec_per_sequence_at_this_batch': round(el / steps / B, 4),
         'steps_with_change': int(nfe_steps)}
  if profile_stride > 0:
    # profiling pass over ONE window (num_strides = 0), outside the timed region
    global D_MODEL
    D_MODEL = cfg.model.hidden_size
    traffic, _ = load_traffic()
    bb = model.backbone
    bb.profile_events = {}
    model._profile = dict(stride=profile_stride, rows=[])
    run(0)
    torch.cuda.synchronize()
    events = bb.profile_events
    bb.profile_events, model._profile = None, None
    rl = kernel_rooflines(events, traffic)
    dom = max((d for t, d in rl.items() if d['bound'] != 'launch' and t not in ('gemm_out', 'gemm_fc2')),
              key=lambda d: d['total_ms'])
    out['roofline'] = dom
  return out


def semi_ar_bench(args):
  rank, local_rank, world = mdist.init()
  torch.cuda.set_device(mdist.local_device(local_rank))
  dev = torch.device('cuda', mdist.local_device(local_rank))
  cfg = load_config(['model=small', f'loader.eval_batch_size={args.batch}',
                     f'sampling.steps={args.T}', 'sampling.semi_ar=True',
                     'sampling.stride_length=512', 'sampling.num_strides=2',
                     'sampling.fast_math=true', f'trainer.devices={world}'])
  model = build_model(cfg, seed=1)
  model.seed_rng(mdist.rank_seed(cfg.seed, rank))
  out = semi_ar_block(model, cfg, args.T, args.steps, args.warmup, rank, world, dev,
                      profile_stride=args.profile_stride)
  out['dist'] = dist_block(world, dev)
  if rank == 0:
    print(json.dumps(out), flush=True)
  mdist.barrier()


def update_kernel_block(model, B, L, V):
  """The fused update kernel alone (SUBS + race from cached logits, every row masked,
  early step) on two synthetic logit distributions: std 2.2 = what the bench's random
  output layer produces; std 8 = a PEAKED distribution, what a trained checkpoint's
  logits look like -- there the fast race proves for most 8-column vectors that no draw
  can reach the mask key and skips their random numbers (same token ids).  Labelled
  secondary figures: the headline runs on the synthetic weights."""
  from mdlm_amd import ops
  st = model._sampler_state(B, L)
  cache, lse = st['cache'], st['lse']
  x = torch.full((B, L), model.mask_index, dtype=torch.int64, device='cuda')
  t = torch.full((B,), 0.9, device='cuda')
  dt = (1 - 1e-5) / 1000
  out = {}
  for name, std in (('near_uniform_std2.2', 2.2), ('peaked_std8', 8.0)):
    cache.normal_(0, std)
    ops.subs_row_lse(cache, x, model.mask_index, out=lse)
    run = lambda k: ops.subs_ddpm_update(cache, lse, x, t, dt, model.mask_index, seed=3, offset=k,
                                         x_next=st['xb'], changed=st['changed'], fast_math=True)
    for k in range(2):
      run(k)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(10):
      run(10 + k)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gbs = B * L * V * 2.0 / (ms * 1e-3) / 1e9
    out[name] = {'ms': round(ms, 4), 'achieved': round(gbs, 1), 'unit': 'GB/s', 'peak': PEAK_HBM_GBS,
                 'frac': round(gbs / PEAK_HBM_GBS, 4)}
  st['changed'].fill_(1)
  return {'kernel': 'ddpm_update_fast_kernel, 64 x 1024 masked rows x 50,258 bf16 logits, t = 0.9',
          **out}


def dist_block(world, dev):
  """What proves the collectives saw N ranks: backend, world size and a SUM
  all-reduce of (rank + 1) checked against N (N + 1) / 2 on every rank."""
  import torch.distributed as tdist
  if world == 1:
    return {'backend': None, 'world_size': 1, 'allreduce_check': True}
  t = torch.tensor([float(mdist.rank() + 1)], dtype=torch.float64, device=dev)
  mdist.all_reduce_sums(t)
  return {'backend': tdist.get_backend(), 'world_size': tdist.get_world_size(),
          'allreduce_check': bool(t.item() == world * (world + 1) / 2)}


def self_launch(args):
  """`python bench.py --gpus N` without torchrun: the parent -- before anything
  touches a GPU -- starts N fresh child processes (one rank per GPU) with
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, relays rank 0's JSON line and
  returns the worst exit code.  A stalled child dumps its Python stacks and
  exits non-zero by itself (faulthandler, armed in main())."""
  import socket
  import subprocess
  n = args.gpus
  if os.environ.get('MDLM_DIST_BACKEND', 'nccl') == 'nccl' and args.workload != 'dist_check':
    have = torch.cuda.device_count()   # counting devices does not initialise HIP
    if have < n:
      raise SystemExit(f'--gpus {n}: {have} GPU(s) visible (one RCCL rank per GPU; '
                       'MDLM_DIST_BACKEND=gloo rehearses N ranks on fewer cards)')
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  procs = []
  for r in range(n):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
               LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    procs.append(subprocess.Popen(
      [sys.executable, '-u', os.path.abspath(__file__)] + sys.argv[1:], env=env,
      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
  rc = 0
  try:
    for line in procs[0].stdout:     # rank 0 prints the one JSON line
      sys.stdout.write(line)
      sys.stdout.flush()
    for p in procs:
      rc = max(rc, abs(p.wait()))
  finally:
    for p in procs:                  # exact PIDs of our own children only
      if p.poll() is None:
        p.terminate()
  return rc


def dist_check_bench(args):
  """Launcher + collective rehearsal without compute (runs on CPU over gloo as
  well): init, barrier, all-reduce check, max-over-ranks; one JSON line."""
  rank, local_rank, world = mdist.init()
  dev = torch.device('cpu')
  if torch.cuda.is_available():
    torch.cuda.set_device(mdist.local_device(local_rank))
    dev = torch.device('cuda', mdist.local_device(local_rank))
  mdist.barrier()
  t0 = time.perf_counter()
  d = dist_block(world, dev)
  el = mdist.max_over_ranks(time.perf_counter() - t0, dev)
  if rank == 0:
    print(json.dumps({'metric': 'dist_check', 'n_gpus': world, 'dist': d,
                      'allreduce_s': round(el, 6)}), flush=True)
  mdist.barrier()
  if not d['allreduce_check']:
    raise SystemExit(3)


# per-kernel roofline bookkeeping of the sampling loop: tag -> (kernel symbol, bound)
KERNELS = {
  # 154 flop/byte: below the machine balance (2.5 PF / 8 TB/s = 312), so HBM is its roofline
  'gemm_out': ('gemm_bf16_kernel<EPI_GATE_RESIDUAL,fp32> attn_out 768->768', 'hbm'),
  'gemm_fc2': ('gemm_bf16_kernel<EPI_GATE_RESIDUAL,fp32> mlp.2 3072->768', 'mfma'),
  'gemm_fc1': ('gemm_bf16_kernel<EPI_BIAS_GELU,bf16> mlp.0 768->3072', 'mfma'),
  'gemm_qkv': ('gemm_bf16_kernel<EPI_QKV_ROTARY,bf16> 768->2304 + rotary', 'mfma'),
  'gemm_vocab': ('gemm_bf16_kernel<EPI_STORE,bf16> vocab projection 768->V, masked rows', 'mfma'),
  'attention': ('attn_fwd_pipe_kernel', 'mfma'),
  'update': ('ddpm_update_fast_kernel (SUBS + race fused from logits)', 'hbm'),
  'lse_combine': ('lse_combine_kernel', 'hbm'),
  'ln_modulate': ('ln_modulate_kernel', 'hbm'),
  'embed': ('embed_rows_kernel', 'hbm'),
  'gather_rows': ('gather_rows_kernel', 'hbm'),
  'row_lists': ('compact_active_kernel + masked_row_map_kernel', 'launch'),
  'gemm_gate_residual': ('gemm_bf16_kernel<EPI_GATE_RESIDUAL,fp32,256x256> (attn_out 768->768 and '
                         'mlp.2 3072->768: one symbol in the rocprof trace)', 'mfma'),
}


D_MODEL = 768   # set from the config in main()


def kernel_rooflines(events, traffic):
  """tag -> roofline dict from the HIP-event pairs of the profiling batch:
  achieved = sum(algorithmic work) / sum(launch time)."""
  vals = {}

  def rows_of(r):
    if torch.is_tensor(r):
      k = (r.data_ptr(), r._version)
      if k not in vals:
        vals[k] = int(r)
      return vals[k]
    return r
  out = {}
  events = dict(events)
  # one kernel symbol in a rocprof trace: both gate/residual GEMM shapes together
  events['gemm_gate_residual'] = events.get('gemm_out', []) + events.get('gemm_fc2', [])
  for tag, evs in events.items():
    if not evs:
      continue
    ms = [s.elapsed_time(e) for s, e, _, _ in evs]
    n = max(1, len(ms))
    rows = [rows_of(r) for _, _, r, _ in evs]
    work = sum(w * r for (_, _, _, w), r in zip(evs, rows))
    t = sum(ms) * 1e-3
    kernel, bound = KERNELS.get(tag, (tag, 'hbm'))
    if tag == 'gemm_out':
      # events carry flops per row (2 D^2); as an HBM-bound kernel it is priced on
      # its bytes: A bf16 (2 D) + residual read and fp32 C write (4 D each) per row
      work = work / (2.0 * D_MODEL * D_MODEL) * (10.0 * D_MODEL)
    d = {'tag': tag, 'kernel': kernel, 'bound': bound, 'launches': len(ms),
         'avg_launch_ms': round(sum(ms) / n, 4), 'total_ms': round(sum(ms), 2),
         'avg_rows_per_launch': round(sum(rows) / n, 1)}
    if bound == 'mfma':
      ach = work / t / 1e12 if t > 0 else 0.0
      d.update(achieved=round(ach, 1), peak=PEAK_BF16_TFLOPS, unit='TFLOP/s',
               frac=round(ach / PEAK_BF16_TFLOPS, 4),
               algorithmic_gflop_per_launch=round(work / n / 1e9, 2))
    elif bound == 'hbm':
      ach = work / t / 1e9 if t > 0 else 0.0
      d.update(achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit='GB/s',
               frac=round(ach / PEAK_HBM_GBS, 4),
               algorithmic_mbytes_per_launch=round(work / n / 1e6, 2))
    tr = traffic.get(tag)   # keys of profiles/pmc_traffic.json = these tags (tools/pmc_traffic.py)
    d['traffic'] = (round(tr['hbm_bytes_per_launch']
                          / tr.get('rows_per_launch', traffic['_rows_per_launch_profiled'])
                          * sum(rows) / n) if tr else None)
    if tr:
      d['traffic_unit'] = ('HBM bytes/launch: PMC FETCH_SIZE x2 + WRITE_SIZE from separate rocprofv3 '
                           '--pmc passes (profiles/pmc_traffic.json), scaled by live rows')
    out[tag] = d
  return out


def start_heartbeat(period_s=30.0):
  """A line on STDERR every `period_s` seconds for as long as the process lives, so that
  a harness that takes a silent process for a hung one sees the long timed region
  (minutes of asynchronous launches, no output by design) as alive.  A daemon thread
  that only sleeps and writes: it touches neither the GPU nor the timed loop."""
  import threading
  t0 = time.time()

  def beat():
    while True:
      time.sleep(period_s)
      print(f'[bench] running, {time.time() - t0:.0f} s', file=sys.stderr, flush=True)
  threading.Thread(target=beat, daemon=True).start()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--workload', default='sample',
                  choices=['sample', 'train', 'semi_ar', 'dist_check'])
  ap.add_argument('--global-batch', type=int, default=512)
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=2)
  ap.add_argument('--warmup', type=int, default=1)
  ap.add_argument('--batch', type=int, default=64)
  ap.add_argument('--T', type=int, default=1000)
  ap.add_argument('--model', default='small')
  ap.add_argument('--cache-policy', default='sample', choices=['sample', 'batch', 'none'])
  ap.add_argument('--race', default='fast', choices=['fast', 'reference'],
                  help='fast: approximate ranking + exact re-evaluation of near-ties; '
                       'reference: every element in the defining arithmetic.  Same tokens.')
  ap.add_argument('--profile-stride', type=int, default=8,
                  help='profiling batch (after the timed region): HIP events around '
                       'every kernel of each k-th denoising step; 0 = skip')
  ap.add_argument('--set', action='append', default=[], metavar='KEY=VALUE',
                  help='extra config override (same-box ablations), e.g. sampling.lse_from_gemm=false')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-policy-compare', action='store_true')
  ap.add_argument('--no-eval', action='store_true')
  ap.add_argument('--no-semi-ar', action='store_true')
  ap.add_argument('--no-train', action='store_true')
  args = ap.parse_args()
  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    raise SystemExit(self_launch(args))
  world_env = int(os.environ.get('WORLD_SIZE', 1))
  if int(os.environ.get('RANK', 0)) == 0:
    start_heartbeat()
  # a stuck rank dumps its Python stacks and exits non-zero (always armed for N > 1)
  dump_s = os.environ.get('MDLM_FAULT_DUMP_S') or ('1500' if world_env > 1 else '')
  if dump_s:
    import faulthandler
    faulthandler.dump_traceback_later(int(dump_s), exit=True)
  if args.workload == 'dist_check':
    return dist_check_bench(args)
  if args.workload == 'train':
    return train_bench(args)
  if args.workload == 'semi_ar':
    return semi_ar_bench(args)

  rank, local_rank, world = mdist.init()
  if world != args.gpus:
    raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
  torch.cuda.set_device(mdist.local_device(local_rank))
  dev = torch.device('cuda', mdist.local_device(local_rank))

  fast = args.race == 'fast'
  cfg = load_config([f'model={args.model}', f'loader.eval_batch_size={args.batch}',
                     f'sampling.steps={args.T}', f'sampling.cache_policy={args.cache_policy}',
                     f'sampling.fast_math={str(fast).lower()}', f'trainer.devices={world}',
                     *args.set])
  model = build_model(cfg, seed=1)
  model.seed_rng(mdist.rank_seed(cfg.seed, rank))
  B, L, V = args.batch, cfg.model.length, model.vocab_size
  D, nb = cfg.model.hidden_size, cfg.model.n_blocks
  global D_MODEL
  D_MODEL = D
  bb = model.backbone

  def sync_all():
    torch.cuda.synchronize()
    mdist.barrier()
    torch.cuda.synchronize()

  dist_info = dist_block(world, dev)
  for _ in range(args.warmup):
    model.restore_model_and_sample(num_steps=args.T)
  sync_all()
  # ---- timed region: no instrumentation of any kind ---------------------------
  t0 = time.perf_counter()
  for _ in range(args.steps):
    model.restore_model_and_sample(num_steps=args.T)
  sync_all()
  elapsed = time.perf_counter() - t0
  elapsed = mdist.max_over_ranks(elapsed, dev)
  nfe_last = int(model.last_nfe.item())
  n_exact = int(model.last_n_exact.item())
  n_groups = len((model._state or {}).get('groups') or [None])

  # ---- profiling batch (outside the timed region): HIP events on the launch
  # stream around every kernel of each k-th step, live row counts of every step
  rooflines, executed = {}, None
  if args.profile_stride > 0:
    # HBM bytes per launch from separate rocprofv3 --pmc passes, if taken on these kernels
    traffic, traffic_source = load_traffic()
    bb.profile_events = {}
    model._profile = dict(stride=args.profile_stride, rows=[])
    model.restore_model_and_sample(num_steps=args.T)
    torch.cuda.synchronize()
    events, rows = bb.profile_events, model._profile['rows']
    bb.profile_events, model._profile = None, None
    rooflines = kernel_rooflines(events, traffic)
    # executed flops: full-width blocks on the active rows; the last block's tail
    # and the vocabulary projection on the masked rows only
    n_rows = sum(int(a) for a, _ in rows)
    n_mrows = sum(int(b) for _, b in rows)
    f_head = 2.0 * D * 3 * D + 4.0 * L * D            # QKV + attention, per row
    f_tail = 2.0 * D * D + 16.0 * D * D                # out-proj + MLP, per row
    nfe_prof = int(model.last_nfe.item())
    executed = ((nb - 1) * (f_head + f_tail) * n_rows + f_head * n_rows
                + (f_tail + 2.0 * D * V) * n_mrows) * (nfe_last / max(1, nfe_prof))

  samples = world * B * args.steps
  value = samples / elapsed
  out = {
    'metric': 'ddpm_cache samples/sec (L=1024, T=1000)',
    'value': round(value, 4), 'unit': 'samples/s', 'n_gpus': world,
    'steps': args.steps, 'warmup': args.warmup,
    'ms_per_step': round(elapsed / args.steps * 1e3, 2),
    'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
    'dtype': 'bf16', 'data': 'synthetic',
    'config': {'workload': (f'mdlm-owt DiT-small ddpm_cache sampling, L={L}, T={args.T}, '
                            f'batch={B}/GPU, bf16, V={V}, noise_removal'),
               'cache_policy': args.cache_policy,
               'race': ('bit-exact; approximate ranking + exact re-evaluation of near-ties '
                        '(sampling.fast_math=true, the default)' if fast else
                        'bit-exact; every element in the defining arithmetic '
                        '(sampling.fast_math=false)'),
               'rng': 'in-kernel Philox4x32-7',
               'stream_groups': n_groups,
               'parallelism': f'replicas x{world}',
               **({'overrides': args.set} if args.set else {})},
    'dist': dist_info,
    'nfe_per_sample': round(nfe_last / B, 1),
    'race_rows_re_evaluated_exactly': n_exact,
    'ms_per_denoise_step': round(elapsed / args.steps / (args.T + 1) * 1e3, 3),
  }
  if rooflines:
    # dominant kernel = largest total time in the profiling batch
    # (by kernel symbol: the two gate/residual shapes count as one kernel)
    dom = max((d for t, d in rooflines.items()
               if d['bound'] != 'launch' and t not in ('gemm_out', 'gemm_fc2')),
              key=lambda d: d['total_ms'])
    out['roofline'] = dict(dom, traffic_source=traffic_source)
    out['rooflines'] = sorted((d for t, d in rooflines.items() if t != 'gemm_gate_residual'),
                              key=lambda d: -d['total_ms'])
    out['profile'] = {'stride': args.profile_stride,
                      'note': ('separate batch after the timed region; HIP events on the launch '
                               'stream around every kernel launch of each k-th denoising step; '
                               'the whole batch on ONE stream so that every kernel is timed alone '
                               '(the timed region overlaps the forwards of config.stream_groups '
                               'sample ranges on as many streams)')}
  # model flops charge every sample-forward 291.65 GFLOP; executed
  # flops count the vocabulary projection and the last block's tail on masked rows only
  out['model_tflops'] = round((nfe_last * 291.65e9 * args.steps) / elapsed / 1e12, 1)
  if executed is not None:
    out['executed_tflops'] = round(executed * args.steps / elapsed / 1e12, 1)

  if not args.no_eval:
    # second half of the metric: NELBO eval tokens/s (q_xt + backbone + fused
    # SUBS-NLL + fp64 sums; sequences sharded by rank, one all-reduce at the end)
    eb = 64
    g = torch.Generator(device='cuda').manual_seed(1 + rank)
    ids = torch.randint(0, V - 1, (eb, L), generator=g, device='cuda')
    ids[:, 0] = 50256; ids[:, -1] = 50256
    batch = {'input_ids': ids, 'attention_mask': torch.ones(eb, L, device='cuda')}
    model.on_validation_epoch_start()
    model.validation_step(batch, 0)
    sync_all()
    model.valid_metrics.reset()
    n_eval = 8
    te = time.perf_counter()
    for i in range(n_eval):
      model.validation_step(batch, i)
    res = model.valid_metrics.compute(sync_dist=True)
    torch.cuda.synchronize()
    te = mdist.max_over_ranks(time.perf_counter() - te, dev)
    out['eval'] = {'metric': 'NELBO eval tokens/sec', 'value': round(world * n_eval * eb * L / te, 1),
                   'unit': 'tokens/s', 'batch_per_gpu': eb, 'batches': n_eval,
                   'val_nll': round(res['val/nll'], 4)}

  if not args.no_semi_ar and args.cache_policy == 'sample' and not args.set:
    # BASELINE configs[3] on the same model and sampler state (already warm): one batch
    out['semi_ar'] = semi_ar_block(model, cfg, args.T, 1, 0, rank, world, dev,
                                   profile_stride=args.profile_stride)

  if rank == 0 and world == 1 and not args.no_policy_compare and args.cache_policy == 'sample':
    # the reference's literal all-or-nothing cache policy, for the record
    cfg.sampling.cache_policy = 'batch'
    torch.cuda.synchronize(); t1 = time.perf_counter()
    model.restore_model_and_sample(num_steps=args.T)
    torch.cuda.synchronize(); tb = time.perf_counter() - t1
    out['reference_cache_policy'] = {
      'policy': 'batch (all-or-nothing, diffusion.py:682-685)',
      'value': round(B / tb, 4), 'unit': 'samples/s',
      'nfe_per_sample': round(int(model.last_nfe.item()) / B, 1)}
    cfg.sampling.cache_policy = args.cache_policy

  if rank == 0 and world == 1 and not args.no_policy_compare and args.cache_policy == 'sample':
    # opt-in two-stage step: same distribution as the reference step, not
    # bit-identical to it on a shared uniform tensor -> never the headline value
    cfg.sampling.lazy_unmask = True
    model.restore_model_and_sample(num_steps=args.T)   # warm-up (state rebuild)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    model.restore_model_and_sample(num_steps=args.T)
    torch.cuda.synchronize(); tb = time.perf_counter() - t1
    out['lazy_unmask_mode'] = {
      'note': ('sampling.lazy_unmask=true: Bernoulli unmask decision, then a categorical draw and the '
               'vocabulary projection for the unmasking rows only; distribution-identical to '
               'diffusion.py:592-610, NOT bit-identical to the reference race on a shared uniform tensor'),
      'value': round(B / tb, 4), 'unit': 'samples/s',
      'nfe_per_sample': round(int(model.last_nfe.item()) / B, 1)}
    cfg.sampling.lazy_unmask = False

  if rank == 0 and world == 1 and not args.no_policy_compare and B * L * V < (1 << 33):
    out['update_kernel'] = update_kernel_block(model, B, L, V)
    # the same sampling workload on a PEAKED output layer (output weights x 3.6: logit std ~ 8, what a
    # trained checkpoint's logits look like), where the fast race skips most of the vocabulary's random
    # numbers.  A labelled secondary figure: the headline `value` is measured on the BASELINE weights.
    w = model.backbone.output_layer.linear.weight
    with torch.no_grad():
      w.mul_(3.6)
    model.backbone.refresh()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    model.restore_model_and_sample(num_steps=args.T)
    torch.cuda.synchronize(); tb = time.perf_counter() - t1
    out['peaked_output_layer'] = {
      'note': 'same workload, output_layer.linear.weight x 3.6 (logit std ~ 8); one batch, state already warm',
      'value': round(B / tb, 4), 'unit': 'samples/s',
      'nfe_per_sample': round(int(model.last_nfe.item()) / B, 1)}
    with torch.no_grad():
      w.div_(3.6)
    model.backbone.refresh()
  del model
  torch.cuda.empty_cache()
  if not args.no_train:
    # BASELINE configs[4]: 20 LM1B training steps (its own model: V = 30,522, L = 128)
    out['train'] = train_block(20, 3, args.global_batch, rank, world, dev,
                               profile=args.profile_stride > 0)
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    out['cpu_baseline'] = cpu_baseline(cfg, args.T)
  if rank == 0:
    print(json.dumps(out), flush=True)
  mdist.barrier()


if __name__ == '__main__':
  main()
