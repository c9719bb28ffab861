"""CPU-only tests: host logic, configuration, the C-ABI surface (load + symbol
export, no compute), loud failure without a GPU, and the N>1 sharding /
all-reduce path over gloo (world_size 2)."""
import json
import math
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


# ------------------------------------------------------------------ C-ABI
def _declared_functions():
  txt = open(os.path.join(ROOT, 'include', 'mdlm_hip.h')).read()
  txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
  return sorted(set(re.findall(r'\b(mdlm_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
  from mdlm_amd import _lib
  declared = _declared_functions()
  assert len(declared) >= 19
  lib = _lib.load()
  for name in declared:
    assert hasattr(lib, name), f'{name} declared in include/mdlm_hip.h but not exported'
  bound = (set(_lib.SIGNATURES) | set(_lib.SIGNATURES_I64)
           | {'mdlm_version', 'mdlm_last_error_string'})
  assert set(declared) == bound, set(declared) ^ bound
  assert _lib.check_symbols()
  assert lib.mdlm_version() >= 100


def test_header_cites_reference_for_each_entry_point():
  txt = open(os.path.join(ROOT, 'include', 'mdlm_hip.h')).read()
  # every hot-path declaration is preceded by a comment citing file:line
  for name in ('mdlm_subs_logprobs', 'mdlm_ddpm_cache_update', 'mdlm_subs_ddpm_update',
               'mdlm_q_xt', 'mdlm_subs_nll', 'mdlm_embed_rows', 'mdlm_ln_modulate',
               'mdlm_gemm_bf16', 'mdlm_attn_fwd', 'mdlm_compact_active'):
    head = txt[:txt.index(f'int {name}(')]
    comment = head[head.rindex('/*'):]
    assert re.search(r'(diffusion|dit)\.py:\d+', comment), name


def test_argument_errors_do_not_need_a_gpu():
  """Argument validation happens before any launch: rc < 0 + message."""
  from mdlm_amd import _lib
  with pytest.raises(RuntimeError, match='null pointer'):
    _lib.call('mdlm_subs_logprobs', 0, 8, 0, 0, 0, 8, 0, 1, 8, 7, 0, 0)
  with pytest.raises(RuntimeError, match='multiple of 128'):
    _lib.call('mdlm_gemm_bf16', 16, 64, 16, 0, 16, 64, 1, 100, 64, 64, 0, 0, 0, 1, 1, 0,
              0, 0, 0, 0, 0, 0, 0, -1, 0, 0)
  with pytest.raises(RuntimeError, match='multiple of'):
    _lib.call('mdlm_attn_fwd', 16, 16, 16, 16, 1, 1, 100, 0, 0)


def test_workspace_queries_match_what_the_host_side_allocates():
  """SURVEY 8(b) "Ownership": workspaces are sized by `*_workspace_bytes`
  (host arithmetic: runs without a GPU).  The sizes must be the ones the kernels
  index (header prose) and the ones mdlm_amd allocates."""
  from mdlm_amd import ops
  B, H, L, D, M = 4, 12, 128, 768, 512
  assert ops.workspace_bytes('attn_bwd', B, H, L) == B * H * L * 4
  assert ops.workspace_bytes('grad_sumsq') == 4096
  assert ops.workspace_bytes('gemm_bf16_splitk', 768, 768, 8) == 8 * 768 * 768 * 4
  assert ops.workspace_bytes('gemm_tn_splitk', 768, 3072, 28) == 28 * 768 * 3072 * 4
  assert ops.workspace_bytes('embed_grad', M + 1, D) == (M // 128 + 1) * D * 4
  assert ops.workspace_bytes('ln_modulate_bwd', M, D) == (M // 32) * 3 * D * 4
  assert ops.workspace_bytes('dropout_gate_residual_bwd', M, D) == (M // 32) * D * 4
  assert ops.workspace_bytes('colsum_bf16_partial', M, 3072) == (M // 128) * 3072 * 4
  assert ops.workspace_bytes('transpose_bf16', 100, 64) == 2 * 64 * 4
  assert ops.workspace_bytes('masked_row_map', M, 256) == (M + 256) * 4
  with pytest.raises(RuntimeError, match='bad shape'):
    ops.workspace_bytes('attn_bwd', 1, 0, 128)
  w = ops.workspace('ln_modulate_bwd', M, D, device='cpu')
  assert w.dtype == torch.float32 and w.numel() == (M // 32) * 3 * D
  # the trainer and the sampler size their workspaces with the queries
  src = open(os.path.join(ROOT, 'mdlm_amd', 'train.py')).read()
  for op in ('attn_bwd', 'grad_sumsq', 'embed_grad', 'ln_modulate_bwd',
             'dropout_gate_residual_bwd'):
    assert f"workspace('{op}'" in src, op


def test_no_cpu_fallback():
  from mdlm_amd import ops
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    ops.subs_logprobs(torch.zeros(1, 2, 8), torch.zeros(1, 2, dtype=torch.int64), 7)
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    ops.q_xt(torch.zeros(1, 2, dtype=torch.int64), torch.zeros(1, 1), 7)


def test_missing_library_fails_loudly(monkeypatch):
  from mdlm_amd import _lib
  monkeypatch.setattr(_lib, '_lib', None)
  monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libmdlm_hip.so')
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    _lib.load()


def test_product_never_imports_the_oracle():
  for dirpath, _, files in os.walk(os.path.join(ROOT, 'mdlm_amd')):
    for f in files:
      if f.endswith('.py'):
        src = open(os.path.join(dirpath, f)).read()
        assert 'mdlm_oracle' not in src and 'oracle' not in src.replace('# oracle', ''), f
  assert 'oracle' not in open(os.path.join(ROOT, 'main.py')).read()


# ------------------------------------------------------------------ config
def test_config_defaults_and_overrides():
  from mdlm_amd.config import load_config
  cfg = load_config([])
  assert cfg.parameterization == 'subs' and cfg.backbone == 'dit' and cfg.T == 0
  assert cfg.time_conditioning is False and cfg.seed == 1
  assert cfg.sampling.predictor == 'ddpm_cache' and cfg.sampling.noise_removal is True
  assert cfg.model.hidden_size == 768 and cfg.model.n_heads == 12 and cfg.model.length == 1024
  assert cfg.training.sampling_eps == 1e-3 and cfg.training.antithetic_sampling is True
  assert cfg.noise.type == 'loglinear'
  assert cfg.loader.eval_batch_size == 512          # div_up(512, 1 device)
  cfg = load_config(['mode=sample_eval', 'sampling.steps=1000', 'loader.eval_batch_size=64',
                     'model.length=1024', 'sampling.semi_ar=True', 'sampling.stride_length=512',
                     'sampling.num_strides=2', 'eval.checkpoint_path=/x/y.ckpt',
                     'trainer.devices=8', '+wandb.offline=true', 'model=small'])
  assert cfg.sampling.steps == 1000 and cfg.loader.eval_batch_size == 64
  assert cfg.sampling.semi_ar is True and cfg.sampling.stride_length == 512
  assert cfg.eval.checkpoint_path == '/x/y.ckpt' and cfg.wandb.offline is True
  assert cfg.loader.batch_size == 64                # div_up(512, 8)
  with pytest.raises(KeyError):
    load_config(['sampling.nope=1'])
  assert load_config(['loader.eval_global_batch_size=100', 'trainer.devices=8']
                     ).loader.eval_batch_size == 13  # div_up


def test_tokenizer_contract():
  from mdlm_amd.config import load_config
  from mdlm_amd.tokenizer import get_tokenizer
  tok = get_tokenizer(load_config([]))
  assert tok.vocab_size == 50257 and tok.bos_token_id == tok.eos_token_id == 50256
  assert tok.mask_token is None
  assert tok.batch_decode([[1, 2], [3]]) == ['<1> <2>', '<3>']


# -------------------------------------------------------------- host math
def test_noise_schedule_matches_reference_golden():
  from mdlm_amd.noise_schedule import LogLinearNoise, get_noise
  from mdlm_amd.config import load_config
  g = np.load(os.path.join(GOLDEN, 'g5_noise.npz'))
  n = get_noise(load_config([]))
  assert isinstance(n, LogLinearNoise)
  s, ds = n(torch.from_numpy(g['t']))
  assert torch.equal(s, torch.from_numpy(g['sigma']))
  assert torch.equal(ds, torch.from_numpy(g['dsigma']))
  assert torch.equal(n.sigma_max, torch.from_numpy(g['sigma_max']))
  with pytest.raises(ValueError):
    get_noise(load_config(['noise.type=cosine']))


def _cpu_model(overrides=()):
  from mdlm_amd.config import load_config
  from mdlm_amd.diffusion import Diffusion
  from mdlm_amd.tokenizer import SyntheticTokenizer
  cfg = load_config(['model=tiny', *overrides])
  return Diffusion(cfg, SyntheticTokenizer(vocab_size=299)), cfg


def test_diffusion_host_logic_on_cpu():
  model, cfg = _cpu_model()
  assert model.mask_index == 299 and model.vocab_size == 300
  # _sample_t against the reference golden (same torch.rand draw)
  g = np.load(os.path.join(GOLDEN, 'g4_sample_t.npz'))
  torch.manual_seed(42)
  t = model._sample_t(8, 'cpu')
  assert torch.equal(t, torch.from_numpy(g['t']))
  # sigma is zeroed without time conditioning (diffusion.py:307-308)
  assert torch.equal(model._process_sigma(torch.rand(3, 1)), torch.zeros(3))
  m2, _ = _cpu_model(['time_conditioning=true'])
  s = torch.rand(3, 1)
  assert torch.equal(m2._process_sigma(s), s.squeeze(-1))
  # a mask-token tokenizer keeps the vocabulary size (diffusion.py:85-90)
  from mdlm_amd.diffusion import Diffusion
  from mdlm_amd.tokenizer import SyntheticTokenizer
  m3 = Diffusion(cfg, SyntheticTokenizer(vocab_size=30522, mask_token_id=103))
  assert m3.mask_index == 103 and m3.vocab_size == 30522
  # the hot path refuses to run on the CPU
  with pytest.raises(RuntimeError):
    model.forward(torch.zeros(1, cfg.model.length, dtype=torch.int64), torch.zeros(1))
  # Philox bookkeeping: offsets advance by the counters a draw consumes
  model.seed_rng(9)
  assert model._draw(10) == (9, 0) and model._draw(5) == (9, 10) and model._rng_offset == 15


def test_ema_swap_roundtrip():
  model, _ = _cpu_model()
  w = model.backbone.blocks[0].attn_qkv.weight
  orig = w.detach().clone()
  model.ema.shadow_params = [p.detach().clone() + 1 for p in model.ema.shadow_params]
  model._ema_swap_in()
  assert torch.allclose(w, orig + 1)
  model._ema_swap_out()
  assert torch.equal(w.detach(), orig)


def test_shadow_weights_layout_checkpoint_and_guards():
  """`ShadowWeights` (stands where models/ema.py's object stands): the checkpoint
  layout of models/ema.py:89-97 in `parameters()` order, 256-byte aligned views of
  one flat buffer, an exchange that needs no third copy, and the guards."""
  from mdlm_amd.shadow import ALIGN, ShadowWeights, flat_layout
  table, n = flat_layout([(3, 5), (7,), (64,), ()])
  assert [t[0] for t in table] == [0, 64, 128, 192] and n == 256 and ALIGN == 64
  assert [t[1] for t in table] == [15, 7, 64, 1]
  ps = [torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7)),
        torch.nn.Parameter(torch.randn(2), requires_grad=False)]
  sh = ShadowWeights(ps, decay=0.9999)
  assert len(sh.table) == 2 and sh.flat.numel() == 128      # frozen parameters have no shadow
  assert all(torch.equal(v, p) for v, p in zip(sh.shadow_params, ps))
  assert sh.shadow_params[1].data_ptr() - sh.flat.data_ptr() == 256
  sd = sh.state_dict()
  assert list(sd) == ['decay', 'num_updates', 'shadow_params'] and sd['num_updates'] == 0
  assert [tuple(t.shape) for t in sd['shadow_params']] == [(3, 5), (7,)]
  # round trip through the reference layout
  sd2 = {'decay': 0.5, 'num_updates': 41, 'shadow_params': [t + 2 for t in sd['shadow_params']]}
  sh.load_state_dict(sd2)
  assert sh.decay == 0.5 and sh.num_updates == 41
  assert torch.equal(sh.shadow_params[0], ps[0].detach() + 2)
  # exchange: live <-> averaged, and back
  live = [p.detach().clone() for p in ps]
  sh.swap_in(ps)
  assert torch.equal(ps[0].detach(), live[0] + 2) and torch.equal(ps[2], live[2])
  assert torch.equal(sh.views()[0], live[0])                # the live values wait in the buffer
  for call in (sh.state_dict, lambda: sh.shadow_params, lambda: sh.load_state_dict(sd2),
               lambda: sh.swap_in(ps)):
    with pytest.raises(RuntimeError, match='swapped in'):
      call()
  sh.swap_out(ps)
  assert all(torch.equal(p.detach(), l) for p, l in zip(ps, live))
  assert torch.equal(sh.shadow_params[0], live[0] + 2)
  with pytest.raises(RuntimeError, match='without swap_in'):
    sh.swap_out(ps)
  with pytest.raises(ValueError, match='outside'):
    ShadowWeights(ps, decay=1.5)
  with pytest.raises(ValueError, match='shadow tensors for'):
    sh.assign([torch.zeros(3, 5)])
  with pytest.raises(ValueError, match='of shape'):
    sh.assign([torch.zeros(3, 5), torch.zeros(8)])
  assert bool(ShadowWeights([], decay=0.5)) is True          # truthiness is not "has parameters"


def test_bench_replays_pmc_traffic_only_for_the_kernels_it_was_measured_on(tmp_path, monkeypatch):
  """bench.py's `roofline.traffic` comes from committed PMC passes; the file is
  stamped with a hash of the kernel sources and dropped when they changed."""
  sys.path.insert(0, ROOT)
  import bench
  assert len(bench.csrc_sha16()) == 16 and bench.csrc_sha16() == bench.csrc_sha16()
  prof = tmp_path / 'profiles'
  prof.mkdir()
  monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
  t, src = bench.load_traffic()
  assert t == {} and src['used'] is False and src['file'] is None
  monkeypatch.setattr(bench, 'csrc_sha16', lambda: 'a' * 16)
  body = {'_csrc_sha16': 'a' * 16, '_git_head': 'deadbeef', '_date': 'd', '_T': 100,
          'gemm_fc1': {'hbm_bytes_per_launch': 5}}
  (prof / 'pmc_traffic.json').write_text(json.dumps(body))
  t, src = bench.load_traffic()
  assert t['gemm_fc1']['hbm_bytes_per_launch'] == 5 and src['used'] is True and src['git_head'] == 'deadbeef'
  monkeypatch.setattr(bench, 'csrc_sha16', lambda: 'b' * 16)
  t, src = bench.load_traffic()
  assert t == {} and src['used'] is False and 'changed' in src['why']
  # the committed file carries a stamp (whether it matches this tree is the bench's business)
  committed = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
  assert len(committed['_csrc_sha16']) == 16 and committed['_T'] == 100


def test_bench_block_guard_keeps_the_headline():
  """bench.py's secondary blocks (eval, semi_ar, train, ...) cannot cost the headline:
  an exception becomes the block's error entry; a block that never returns makes the
  process print the line as it stands and exit 0."""
  sys.path.insert(0, ROOT)
  import bench
  out = {'value': 1.0}
  g = bench.BlockGuard(out, 0)
  g.run('ok', 5, lambda: out.__setitem__('ok', 1))
  g.run('bad', 5, lambda: 1 / 0)
  assert out['ok'] == 1 and 'ZeroDivisionError' in out['bad']['error'] and out['value'] == 1.0
  code = ('import sys, time; sys.path.insert(0, %r); import bench; out = {"value": 2.5}; '
          'bench.BlockGuard(out, 0).run("stuck", 1, lambda: time.sleep(120)); print("not reached")' % ROOT)
  r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=200)
  assert r.returncode == 0, r.stderr[-500:]
  lines = r.stdout.strip().splitlines()
  d = json.loads(lines[-1])
  assert d['value'] == 2.5 and 'no result after 1 s' in d['stuck']['error'] and 'not reached' not in r.stdout
  # the other ranks leave too, silently
  r = subprocess.run([sys.executable, '-c', code.replace('BlockGuard(out, 0)', 'BlockGuard(out, 3)')],
                     capture_output=True, text=True, timeout=200)
  assert r.returncode == 0 and r.stdout.strip() == ''


def test_state_dict_layout_matches_reference():
  """131 entries for DiT-small; same names as models/dit.py."""
  from mdlm_amd.dit import DIT
  from mdlm_amd.config import load_config
  import mdlm_oracle as orc
  cfg = load_config(['model=tiny'])
  m = DIT(cfg, vocab_size=300)
  ref_keys = set(orc.synthetic_dit_params(128, 2, 2, 64, 300)) | {'rotary_emb.inv_freq'}
  assert set(m.state_dict()) == ref_keys
  assert len(ref_keys) == 5 + 1 + 2 * 10 + 5
  with pytest.raises(ValueError, match='head_dim'):
    DIT({'model': dict(hidden_size=64, cond_dim=16, n_heads=4, n_blocks=1, dropout=0.1,
                       scale_by_sigma=True)}, vocab_size=10)


def test_expected_nfe_model():
  sys.path.insert(0, ROOT)
  import bench
  nfe = bench.expected_nfe_b1(1000, 1024, trials=8)
  assert 600 < nfe < 680      # ~ 1000 (1 - e^-1.02) + 1, SURVEY section 3.1


# ------------------------------------------------------------ multi-process
def test_shard_indices():
  from mdlm_amd.dist import rank_seed, shard_indices
  assert shard_indices(10, 0, 4) == [0, 4, 8]
  assert shard_indices(10, 3, 4) == [3, 7, 1]        # padded by wrap-around
  allidx = sum((shard_indices(10, r, 4) for r in range(4)), [])
  assert set(allidx) == set(range(10)) and len(allidx) == 12
  assert rank_seed(1, 3) == 4


_WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, sys.argv[1])
from mdlm_amd import dist as mdist
rank, local_rank, world = mdist.init(backend='gloo')
assert world == 2 and mdist.world_size() == 2 and mdist.rank() == rank
# eval sharded by sequence: each rank sums its shard's (nll * mask, mask)
torch.manual_seed(0)
nll = torch.rand(11, 16, dtype=torch.float64)
mask = (torch.rand(11, 16) > 0.2).double()
idx = mdist.shard_indices(11, rank, world)
sums = torch.stack([(nll[idx] * mask[idx]).sum(), mask[idx].sum()])
mdist.all_reduce_sums(sums)
tok = torch.arange(4).view(1, 4) + 10 * rank
allt = mdist.gather_tokens(tok)
mx = mdist.max_over_ranks(1.0 + rank, 'cpu')
# DDP gradient buckets of the trainer (configs/strategy/ddp.yaml): contiguous
# slices of one flat gradient buffer, summed across ranks as they are launched
from mdlm_amd.train import GradBuckets
torch.manual_seed(7)
base = torch.randn(2, 1000)
flat = base[rank].clone()
slices = [(640, 1000), (128, 640), (64, 128), (64, 64), (0, 64)]
gb = GradBuckets(flat, slices)
assert gb.enabled
for i in range(len(slices)):
  gb.launch(i)
gb.wait_all()
ddp_ok = bool(torch.equal(flat, base[0] + base[1])) and gb.work == []
mdist.barrier()
if rank == 0:
  full = mdist.shard_indices(11, 0, 2) + mdist.shard_indices(11, 1, 2)
  ref = torch.stack([(nll[full] * mask[full]).sum(), mask[full].sum()])
  print(json.dumps({'ok': bool(torch.allclose(sums, ref)), 'gathered': allt.tolist(),
                    'max': mx, 'ddp_ok': ddp_ok, 'seeds': [mdist.rank_seed(1, r) for r in range(world)]}))
'''


def test_gloo_world_size_2(tmp_path):
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  script = tmp_path / 'worker.py'
  script.write_text(_WORKER)
  procs = []
  for r in range(2):
    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2',
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
  outs = [p.communicate(timeout=120) for p in procs]
  for p, (o, e) in zip(procs, outs):
    assert p.returncode == 0, e[-2000:]
  import json
  res = json.loads(outs[0][0].strip().splitlines()[-1])
  assert res['ok'] and res['ddp_ok'] and res['max'] == 2.0 and res['seeds'] == [1, 2]
  assert res['gathered'] == [[0, 1, 2, 3], [10, 11, 12, 13]]


def test_bench_self_launch_two_gloo_ranks():
  """`python bench.py --gpus 2` without torchrun starts its own ranks (one child
  process each, RANK / WORLD_SIZE / MASTER_* set by the parent) and relays rank
  0's single JSON line; here over gloo on the CPU with the collective-only
  workload (reference launcher: main.py:137-145; metric sync diffusion.py:380-383)."""
  import json
  import subprocess
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
  env['MDLM_DIST_BACKEND'] = 'gloo'
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
                      '--workload', 'dist_check'], env=env, capture_output=True,
                     text=True, timeout=300)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1
  d = json.loads(lines[0])['dist']
  assert d == {'backend': 'gloo', 'world_size': 2, 'allreduce_check': True}


def _local_gpt2_style_tokenizer(path):
  """A byte-level BPE tokenizer in the GPT-2 file format, trained here on a few
  sentences (the hub is not reachable) and saved as a local directory."""
  import tokenizers
  import transformers
  tk = tokenizers.Tokenizer(tokenizers.models.BPE())
  tk.pre_tokenizer = tokenizers.pre_tokenizers.ByteLevel(add_prefix_space=False)
  tk.decoder = tokenizers.decoders.ByteLevel()
  trainer = tokenizers.trainers.BpeTrainer(
    vocab_size=400, special_tokens=['<|endoftext|>'],
    initial_alphabet=tokenizers.pre_tokenizers.ByteLevel.alphabet())
  corpus = ['the quick brown fox jumps over the lazy dog',
            'masked diffusion language models sample tokens in any order',
            'a journey of a thousand miles begins with a single step'] * 4
  tk.train_from_iterator(corpus, trainer)
  fast = transformers.PreTrainedTokenizerFast(
    tokenizer_object=tk, bos_token='<|endoftext|>', eos_token='<|endoftext|>',
    unk_token='<|endoftext|>')
  fast.save_pretrained(path)
  return corpus


def test_local_tokenizer_directory_roundtrip(tmp_path):
  """dataloader.py:488-522 with a LOCAL tokenizer directory (hub names cannot be
  fetched here): loading, the bos / eos / pad post-processing, the
  mask_index = vocab_size rule of diffusion.py:85-90, `batch_decode` of sampled
  ids (main.py:103-104) and the EOS-based `sequence_lengths` of
  diffusion.py:989-992."""
  from mdlm_amd.config import load_config
  from mdlm_amd.tokenizer import get_tokenizer
  d = str(tmp_path / 'tok')
  corpus = _local_gpt2_style_tokenizer(d)
  cfg = load_config([f'data.tokenizer_name_or_path={d}'])
  tok = get_tokenizer(cfg)
  assert tok.bos_token == tok.eos_token == '<|endoftext|>' and tok.pad_token == '[PAD]'
  assert getattr(tok, 'mask_token', None) is None        # -> mask_index = vocab_size
  ids = tok(corpus[:2])['input_ids']
  assert tok.batch_decode(ids) == corpus[:2]             # exact round trip
  # sequence_lengths: tokens before the first EOS after position 0 (989-992)
  eos = tok.eos_token_id
  toks = np.array([[eos] + ids[0][:5] + [eos] + ids[1][:3],
                   [eos] + ids[1][:9]])
  seq_len = ((toks[:, 1:] == eos).cumsum(-1) == 0).sum(-1)
  assert seq_len.tolist() == [5, 9]
