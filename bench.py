"""Benchmark of the MDLM ddpm_cache sampling hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full `restore_model_and_sample(num_steps=1000)` over one batch
(BASELINE.json configs[1]: DiT-small 169M, L=1024, T=1000, batch 64/GPU, bf16
operands, GPT-2-shaped vocab 50,258, seeded synthetic weights, all-mask prior).
Ranks are independent replicas (seed + rank); no data-path collective.  Rank 0
prints ONE JSON line: metric = ddpm_cache samples/sec (whole job), plus
`roofline` for the dominant kernel (the gate/residual bf16 MFMA GEMM
instantiation, timed live with HIP events on the launch stream; the vocab
projection is reported beside it) and `cpu_baseline` (the CPU oracle timed on
this box's host cores on a bounded sample).

Other workloads of BASELINE.json's config list, each printing its own JSON line
(never the driver's default): `--workload train` (configs[4], LM1B training
step tokens/s) and `--workload semi_ar` (configs[3], 2048-token semi-AR
sequences/s).  `MDLM_DIST_BACKEND=gloo` rehearses N ranks on a one-GPU box.
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


def train_bench(args):
  """BASELINE.json configs[4]: LM1B MDLM training step, DiT-small, L=128, global
  batch 512 split over the ranks (strong scaling), bert-base-uncased-shaped
  vocabulary (30,522, [MASK]=103 in-vocabulary), dropout 0.1, clip 1.0, AdamW,
  EMA, DDP gradient all-reduce on RCCL overlapped with the backward.  A step =
  forward + backward + all-reduce + optimizer/EMA over one global batch."""
  from mdlm_amd.train import Trainer
  rank, local_rank, world = mdist.init()
  torch.cuda.set_device(mdist.local_device(local_rank))
  dev = torch.device('cuda', mdist.local_device(local_rank))
  gb = args.global_batch
  cfg = load_config(['model=small', 'data=lm1b', 'model.length=128',
                     f'loader.global_batch_size={gb}', f'trainer.devices={world}'])
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
  for _ in range(max(1, args.warmup)):
    loss = tr.training_step(batch)
  torch.cuda.synchronize(); mdist.barrier(); torch.cuda.synchronize()
  steps = args.steps if args.steps > 2 else 20
  t0 = time.perf_counter()
  for _ in range(steps):
    loss = tr.training_step(batch)
  torch.cuda.synchronize(); mdist.barrier(); torch.cuda.synchronize()
  el = mdist.max_over_ranks(time.perf_counter() - t0, dev)
  n_par = sum(p.numel() for p in model.backbone.parameters())
  # forward flops per token: blocks 2*(3+1+8)*D^2 + attention 4*L*D, vocab 2*D*V; x3 for fwd+bwd
  D, nb = cfg.model.hidden_size, cfg.model.n_blocks
  f_tok = nb * (24 * D * D + 4 * L * D) + 2 * D * V
  out = {'metric': 'LM1B MDLM training step tokens/sec (DiT-small, L=128, global batch 512)',
         'value': round(world * B * L * steps / el, 1), 'unit': 'tokens/s', 'n_gpus': world,
         'steps': steps, 'warmup': max(1, args.warmup), 'ms_per_step': round(el / steps * 1e3, 2),
         'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'bf16',
         'data': 'synthetic',
         'config': {'workload': f'LM1B MDLM training step, DiT-small ({n_par/1e6:.1f}M), L={L}, '
                                f'global batch {world * B}, V={V}, dropout 0.1, clip 1.0, AdamW+EMA',
                    'batch_per_gpu': B, 'parallelism': f'ddp x{world}',
                    'grad_allreduce_mbytes': round(tr.n_flat * 4 / 1e6, 1)},
         'loss': round(float(loss), 4),
         'model_tflops': round(3 * f_tok * world * B * L * steps / el / 1e12, 1)}
  if rank == 0:
    print(json.dumps(out), flush=True)
  mdist.barrier()


def semi_ar_bench(args):
  """BASELINE.json configs[3]: semi-autoregressive generation (README recipe:
  L=1024 windows, stride 512, 2 extra strides -> 2048 tokens per sequence,
  dt = 1/1000), batch 64 per GPU, replicas sharded by sample (no collective).
  A step = one restore_model_and_semi_ar_sample over one batch."""
  rank, local_rank, world = mdist.init()
  torch.cuda.set_device(mdist.local_device(local_rank))
  dev = torch.device('cuda', mdist.local_device(local_rank))
  cfg = load_config(['model=small', f'loader.eval_batch_size={args.batch}',
                     f'sampling.steps={args.T}', 'sampling.semi_ar=True',
                     'sampling.stride_length=512', 'sampling.num_strides=2',
                     'sampling.fast_math=true', f'trainer.devices={world}'])
  model = build_model(cfg, seed=1)
  model.seed_rng(mdist.rank_seed(cfg.seed, rank))
  B, L = args.batch, cfg.model.length
  run = lambda: model.restore_model_and_semi_ar_sample(
    stride_length=cfg.sampling.stride_length, num_strides=cfg.sampling.num_strides,
    dt=1 / args.T)
  for _ in range(args.warmup):
    run()
  torch.cuda.synchronize(); mdist.barrier(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    nfe_steps, _, _ = run()
  torch.cuda.synchronize(); mdist.barrier(); torch.cuda.synchronize()
  el = mdist.max_over_ranks(time.perf_counter() - t0, dev)
  n_tok = L + cfg.sampling.num_strides * cfg.sampling.stride_length
  out = {'metric': 'semi-AR generation sequences/sec (2048 tokens, stride 512, T=1000 per window)',
         'value': round(world * B * args.steps / el, 4), 'unit': 'sequences/s', 'n_gpus': world,
         'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(el / args.steps * 1e3, 1),
         'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16',
         'data': 'synthetic',
         'config': {'workload': f'semi-AR sample_subs_guidance, DiT-small, window L={L}, stride 512, '
                                f'num_strides 2 -> {n_tok} tokens, dt=1/{args.T}, batch={B}/GPU',
                    'parallelism': f'replicas x{world}'},
         'tokens_per_sec': round(world * B * n_tok * args.steps / el, 1),
         'sec_per_sequence_at_this_batch': round(el / args.steps / B, 4),
         'steps_with_change': int(nfe_steps)}
  if rank == 0:
    print(json.dumps(out), flush=True)
  mdist.barrier()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--workload', default='sample', choices=['sample', 'train', 'semi_ar'])
  ap.add_argument('--global-batch', type=int, default=512)
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=2)
  ap.add_argument('--warmup', type=int, default=1)
  ap.add_argument('--batch', type=int, default=64)
  ap.add_argument('--T', type=int, default=1000)
  ap.add_argument('--model', default='small')
  ap.add_argument('--cache-policy', default='sample', choices=['sample', 'batch', 'none'])
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-policy-compare', action='store_true')
  ap.add_argument('--no-eval', action='store_true')
  args = ap.parse_args()
  if os.environ.get('MDLM_FAULT_DUMP_S'):   # debugging aid: Python stacks of a stuck rank
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ['MDLM_FAULT_DUMP_S']), exit=True)
  if args.workload == 'train':
    return train_bench(args)
  if args.workload == 'semi_ar':
    return semi_ar_bench(args)

  rank, local_rank, world = mdist.init()
  if world != args.gpus:
    if args.gpus > 1:
      raise SystemExit(f'--gpus {args.gpus} needs torchrun with {args.gpus} ranks '
                       f'(WORLD_SIZE={world})')
  torch.cuda.set_device(mdist.local_device(local_rank))
  dev = torch.device('cuda', mdist.local_device(local_rank))

  cfg = load_config([f'model={args.model}', f'loader.eval_batch_size={args.batch}',
                     f'sampling.steps={args.T}', f'sampling.cache_policy={args.cache_policy}',
                     'sampling.fast_math=true', f'trainer.devices={world}'])
  model = build_model(cfg, seed=1)
  model.seed_rng(mdist.rank_seed(cfg.seed, rank))
  B, L, V = args.batch, cfg.model.length, model.vocab_size
  bb = model.backbone
  bb.profile_events = None

  def sync_all():
    torch.cuda.synchronize()
    mdist.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    model.restore_model_and_sample(num_steps=args.T)
  sync_all()
  # live HIP-event timing of the two heaviest GEMM instantiations
  bb.profile_events = {'gate_residual': [], 'vocab': []}
  t0 = time.perf_counter()
  for _ in range(args.steps):
    model.restore_model_and_sample(num_steps=args.T)
  torch.cuda.synchronize()
  mdist.barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  elapsed = mdist.max_over_ranks(elapsed, dev)
  events = bb.profile_events
  bb.profile_events = None
  nfe_last = int(model.last_nfe.item())

  # ---- roofline: algorithmic flops / launch time, from the live events ------
  traffic = {}
  tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
  if os.path.exists(tpath):   # HBM bytes per launch from separate rocprofv3 --pmc passes
    traffic = json.load(open(tpath))

  def _traffic(tag, rows):
    t = traffic.get(tag)
    if not t:
      return None
    return round(t['hbm_bytes_per_launch'] / traffic.get('_rows_per_launch_profiled', 65536) * rows)

  def roof(tag, kernel):
    evs = events[tag]
    ms = [s.elapsed_time(e) for s, e, _, _ in evs]
    cache = {}
    flops = 0.0
    rows_sum = 0
    for _, _, r, fpr in evs:
      if torch.is_tensor(r):
        k = r.data_ptr()
        if k not in cache:
          cache[k] = int(r)
        r = cache[k]
      flops += fpr * r
      rows_sum += r
    n = len(ms)
    t = sum(ms) * 1e-3
    ach = flops / t / 1e12 if t > 0 else 0.0
    return {'kernel': kernel, 'bound': 'mfma', 'achieved': round(ach, 1),
            'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(ach / PEAK_BF16_TFLOPS, 4),
            'traffic': _traffic(tag, rows_sum / max(1, n)),
            'traffic_unit': 'HBM bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/pmc_traffic.json, scaled by rows)',
            'launches': n, 'avg_launch_ms': round(sum(ms) / max(1, n), 4),
            'avg_rows_per_launch': round(rows_sum / max(1, n), 1),
            'algorithmic_gflop_per_launch': round(flops / max(1, n) / 1e9, 2)}
  # dominant kernel by total time (profiles/r01_*_kernel_stats.csv): the
  # gate/residual GEMM instantiation (out-projection K=768 and MLP-down K=3072)
  roofline = roof('gate_residual',
                  'gemm_bf16_kernel<EPI_GATE_RESIDUAL,fp32,256x256> (attn_out 768->768 and mlp.2 3072->768)')
  roofline_vocab = roof('vocab', 'gemm_bf16_kernel<EPI_STORE,bf16,256x256> (vocab projection 768->50258, masked rows)')

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
               'cache_policy': args.cache_policy, 'rng': 'in-kernel Philox4x32-10',
               'parallelism': f'replicas x{world}'},
    'nfe_per_sample': round(nfe_last / B, 1),
    'ms_per_denoise_step': round(elapsed / args.steps / (args.T + 1) * 1e3, 3),
    'roofline': roofline,
    'roofline_vocab_gemm': roofline_vocab,
  }
  out['backbone_tflops'] = round(
    (nfe_last * 291.65e9 * args.steps) / elapsed / 1e12, 1)

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

  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    del model
    torch.cuda.empty_cache()
    out['cpu_baseline'] = cpu_baseline(cfg, args.T)
  if rank == 0:
    print(json.dumps(out), flush=True)
  mdist.barrier()


if __name__ == '__main__':
  main()
