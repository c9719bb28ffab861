"""Two builds, per active-sample count: gate/residual, MLP-up and QKV GEMMs with a device-side row count."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from mdlm_amd._lib import SIGNATURES, EPI_STORE, EPI_BIAS_GELU, EPI_GATE_RESIDUAL, EPI_QKV_ROTARY
libs = []
for path in sys.argv[1:3]:
  l = C.CDLL(path); l.mdlm_gemm_bf16.restype = C.c_int; l.mdlm_gemm_bf16.argtypes = SIGNATURES['mdlm_gemm_bf16']; libs.append(l)
torch.manual_seed(0)
L, D = 1024, 768
M = 64 * L
dev = 'cuda'
h = torch.randn(M, D, device=dev).bfloat16(); mh = torch.randn(M, 4 * D, device=dev).bfloat16()
x0 = torch.randn(M, D, device=dev)
Wqkv = (torch.randn(3 * D, D, device=dev) * 0.04).bfloat16(); Wo = (torch.randn(D, D, device=dev) * 0.02).bfloat16()
W1 = (torch.randn(4 * D, D, device=dev) * 0.02).bfloat16(); b1 = torch.randn(4 * D, device=dev) * 0.02
W2 = (torch.randn(D, 4 * D, device=dev) * 0.02).bfloat16(); b2 = torch.randn(D, device=dev) * 0.02
mod = torch.randn(1, 6 * D, device=dev); cos = torch.randn(L, 32, device=dev); sin = torch.randn(L, 32, device=dev)
P = lambda t: t.data_ptr() if t is not None else 0
st = torch.cuda.current_stream().cuda_stream
def call(l, A, lda, W, bias, Cc, ldc, dt, N, K, epi, live, gate=None, res=None, rc=None, rs=None, nh=0):
  rc_ = l.mdlm_gemm_bf16(P(A), lda, P(W), P(bias), P(Cc), ldc, dt, M, N, K, epi, P(gate), P(res), L, 1, 6 * D, P(rc), P(rs), nh, P(live), 0, 0, 0, -1, 0, st)
  assert rc_ == 0
def tm(fn, iters=20):
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3
ns = [int(a) for a in sys.argv[3].split(',')] if len(sys.argv) > 3 else list(range(30, 56, 2))
tot = [[0.0] * 4 for _ in libs]
for n in ns:
  live = torch.tensor([n * L], dtype=torch.int32, device=dev)
  outs = []
  row = f'n={n}:'
  for i, l in enumerate(libs):
    x = x0.clone(); mh2 = torch.zeros_like(mh); qkv = torch.zeros(3 * M * D, device=dev, dtype=torch.bfloat16)
    fns = [lambda: call(l, h, D, Wo, None, x, D, 0, D, D, EPI_GATE_RESIDUAL, live, gate=mod[:, :D], res=x),
           lambda: call(l, mh, 4 * D, W2, b2, x, D, 0, D, 4 * D, EPI_GATE_RESIDUAL, live, gate=mod[:, :D], res=x),
           lambda: call(l, h, D, W1, b1, mh2, 4 * D, 1, 4 * D, D, EPI_BIAS_GELU, live),
           lambda: call(l, h, D, Wqkv, None, qkv, 0, 1, 3 * D, D, EPI_QKV_ROTARY, live, rc=cos, rs=sin, nh=12)]
    # correctness of one application each (fresh x)
    x.copy_(x0); fns[0](); o_out = x.clone(); x.copy_(x0); fns[1](); o_fc2 = x.clone(); fns[2](); fns[3]()
    outs.append((o_out, o_fc2, mh2.clone(), qkv.clone()))
    ts = []
    for k, fn in enumerate(fns):
      tm(fn, 3); t = sorted(tm(fn) for _ in range(3))[1]; ts.append(t); tot[i][k] += t
    row += '   ' + ' '.join(f'{t:6.1f}' for t in ts)
  same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[-1]))
  print(row + f'   (out fc2 fc1 qkv us; A then B)  bit-identical={same}', flush=True)
print('sum:', [[round(v, 1) for v in t] for t in tot])
