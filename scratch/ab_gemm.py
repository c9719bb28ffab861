"""A/B two builds of libmdlm_hip.so in ONE process (same device, interleaved
launches): python3 scratch/ab_gemm.py libA.so libB.so [M]"""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from mdlm_amd._lib import SIGNATURES, EPI_STORE, EPI_BIAS_GELU, EPI_GATE_RESIDUAL, EPI_QKV_ROTARY
libs = []
for path in sys.argv[1:3]:
  l = C.CDLL(path)
  l.mdlm_gemm_bf16.restype = C.c_int
  l.mdlm_gemm_bf16.argtypes = SIGNATURES['mdlm_gemm_bf16']
  libs.append(l)
M = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
only = sys.argv[5].split(',') if len(sys.argv) > 5 else None
torch.manual_seed(0)
L, D, V = 1024, 768, 50258
dev = 'cuda'
h = torch.randn(M, D, device=dev).bfloat16(); mh = torch.randn(M, 4 * D, device=dev).bfloat16()
x = torch.randn(M, D, device=dev)
Wqkv = (torch.randn(3 * D, D, device=dev) * 0.04).bfloat16(); Wo = (torch.randn(D, D, device=dev) * 0.02).bfloat16()
W1 = (torch.randn(4 * D, D, device=dev) * 0.02).bfloat16(); b1 = torch.randn(4 * D, device=dev) * 0.02
W2 = (torch.randn(D, 4 * D, device=dev) * 0.02).bfloat16(); b2 = torch.randn(D, device=dev) * 0.02
Wv = (torch.randn(V, D, device=dev) * 0.08).bfloat16(); bv = torch.randn(V, device=dev) * 0.02
qkv = torch.empty(3 * M * D, device=dev, dtype=torch.bfloat16)
Vp = (V + 127) // 128 * 128
logits = torch.empty(M, Vp, device=dev, dtype=torch.bfloat16)
mod = torch.randn(1, 6 * D, device=dev); cos = torch.randn(L, 32, device=dev); sin = torch.randn(L, 32, device=dev)
P = lambda t: t.data_ptr() if t is not None else 0
st = torch.cuda.current_stream().cuda_stream
def call(l, A, lda, W, bias, Cc, ldc, dt, N, K, epi, gate=None, res=None, rpm=L, nmod=1, modld=6 * D, rc=None, rs=None, nh=0):
  rc_ = l.mdlm_gemm_bf16(P(A), lda, P(W), P(bias), P(Cc), ldc, dt, M, N, K, epi, P(gate), P(res), rpm, nmod, modld, P(rc), P(rs), nh, 0, 0, 0, 0, -1, mode, st)
  assert rc_ == 0
shapes = {
  'qkv': lambda l: call(l, h, D, Wqkv, None, qkv, 0, 1, 3 * D, D, EPI_QKV_ROTARY, rc=cos, rs=sin, nh=12),
  'out': lambda l: call(l, h, D, Wo, None, x, D, 0, D, D, EPI_GATE_RESIDUAL, gate=mod[:, :D], res=x),
  'mlp1': lambda l: call(l, h, D, W1, b1, mh, 4 * D, 1, 4 * D, D, EPI_BIAS_GELU),
  'mlp2': lambda l: call(l, mh, 4 * D, W2, b2, x, D, 0, D, 4 * D, EPI_GATE_RESIDUAL, gate=mod[:, :D], res=x),
  'vocab': lambda l: call(l, h, D, Wv, bv, logits, Vp, 1, V, D, EPI_STORE),
}
flops = {'qkv': 2 * M * 3 * D * D, 'out': 2 * M * D * D, 'mlp1': 2 * M * 4 * D * D, 'mlp2': 2 * M * 4 * D * D, 'vocab': 2 * M * V * D}
def tm(fn, l, iters):
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn(l)
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3
for name, fn in shapes.items():
  if only and name not in only: continue
  it = 4 if name == 'vocab' else 20
  for l in libs: tm(fn, l, 3)
  res = [[], []]
  for rep in range(5):
    for i, l in enumerate(libs):
      res[i].append(tm(fn, l, it))
  a, b = (sorted(r)[len(r) // 2] for r in res)
  print(f'{name:6s} A {a:8.1f}us {flops[name]/a/1e6:6.0f}TF   B {b:8.1f}us {flops[name]/b/1e6:6.0f}TF   B/A {b/a:.3f}')
