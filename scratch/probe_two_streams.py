"""Do two half-size launch chains on two streams fill each other's tail rounds?  One launch chain of n samples
against two concurrent chains of ceil(n/2) and floor(n/2) samples (LN -> MLP-up -> MLP-down, 6 times)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdlm_amd import ops
from mdlm_amd._lib import EPI_BIAS_GELU, EPI_GATE_RESIDUAL
torch.manual_seed(0)
D, L, CAP = 768, 1024, 64
dev = 'cuda'
x = torch.randn(CAP * L, D, device=dev)
h = torch.empty(CAP * L, D, device=dev, dtype=torch.bfloat16)
a = torch.empty(CAP * L, 4 * D, device=dev, dtype=torch.bfloat16)
w1 = (torch.randn(4 * D, D, device=dev) * 0.03).bfloat16(); b1 = torch.zeros(4 * D, device=dev)
w2 = (torch.randn(D, 4 * D, device=dev) * 0.02).bfloat16(); b2 = torch.zeros(D, device=dev)
lnw = torch.ones(D, device=dev); mod = torch.randn(CAP, 6 * D, device=dev) * 0.1
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def chain(s0, n, reps=6):
  r0, M = s0 * L, n * L
  xs, hs, as_ = x[r0:r0 + M], h[r0:r0 + M], a[r0:r0 + M]
  md = mod[s0:s0 + n]
  for _ in range(reps):
    ops.ln_modulate(xs, lnw, md[:, 0:D], md[:, D:2 * D], hs, L, n, 6 * D)
    ops.gemm_bf16(hs, w1, as_, EPI_BIAS_GELU, bias=b1)
    ops.gemm_bf16(as_, w2, xs, EPI_GATE_RESIDUAL, bias=b2, gate=md[:, 2 * D:3 * D], residual=xs,
                  rows_per_mod=L, n_mod=n, mod_ld=6 * D)

def timed(fn, iters=5):
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  fn(); torch.cuda.synchronize()
  ts = []
  for _ in range(iters):
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
  return sorted(ts)[len(ts) // 2]

def one(n):
  chain(0, n)

def two(n):
  nA = (n + 1) // 2
  ev = torch.cuda.Event(); ev.record()
  s1.wait_event(ev); s2.wait_event(ev)
  with torch.cuda.stream(s1): chain(0, nA)
  with torch.cuda.stream(s2): chain(nA, n - nA)
  e1 = torch.cuda.Event(); e2 = torch.cuda.Event()
  e1.record(s1); e2.record(s2)
  torch.cuda.current_stream().wait_event(e1); torch.cuda.current_stream().wait_event(e2)

def two_serial(n):
  nA = (n + 1) // 2
  chain(0, nA); chain(nA, n - nA)

for n in (34, 38, 40, 42, 43, 44, 46, 50, 56, 64):
  t1, t2, t3 = timed(lambda: one(n)), timed(lambda: two(n)), timed(lambda: two_serial(n))
  print(f'n={n}: one chain {t1:8.1f} us   two streams {t2:8.1f} us ({t2 / t1:.3f})   two halves, one stream {t3:8.1f} us', flush=True)
