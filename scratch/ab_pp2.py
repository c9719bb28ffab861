"""MLP-up (bias + GELU, bf16) : two-accumulator-set kernel (tile | 32) vs the 256x256 kernel."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdlm_amd._lib import SIGNATURES, EPI_BIAS_GELU
path = sys.argv[1] if len(sys.argv) > 1 else 'mdlm_amd/libmdlm_hip.so'
l = C.CDLL(os.path.abspath(path))
l.mdlm_gemm_bf16.restype = C.c_int
l.mdlm_gemm_bf16.argtypes = SIGNATURES['mdlm_gemm_bf16']
l.mdlm_last_error_string.restype = C.c_char_p
ns = [int(a) for a in sys.argv[2:]] or [64, 40]
torch.manual_seed(0)
L, D = 1024, 768
dev = 'cuda'
P = lambda t: t.data_ptr() if t is not None else 0
st = torch.cuda.current_stream().cuda_stream
def call(mode, A, W, bias, out, M, N, K):
  rc = l.mdlm_gemm_bf16(P(A), K, P(W), P(bias), P(out), N, 1, M, N, K, EPI_BIAS_GELU, 0, 0,
                        L, 1, 0, 0, 0, 0, 0, 0, 0, 0, -1, mode, st)
  assert rc == 0, l.mdlm_last_error_string()
def tm(fn, iters):
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3
for n in ns:
  M = n * L
  h = torch.randn(M, D, device=dev).bfloat16()
  W1 = (torch.randn(4 * D, D, device=dev) * 0.02).bfloat16(); b1 = torch.randn(4 * D, device=dev) * 0.02
  o0 = torch.zeros(M, 4 * D, device=dev, dtype=torch.bfloat16); o1 = torch.zeros_like(o0)
  call(0, h, W1, b1, o0, M, 4 * D, D); call(32, h, W1, b1, o1, M, 4 * D, D)
  torch.cuda.synchronize()
  ref = torch.nn.functional.gelu(h.float() @ W1.float().T + b1, approximate='tanh')
  print(f'n={n} mlp1: identical to 256x256 kernel: {torch.equal(o0, o1)}  max err vs fp32 {float((o1.float() - ref).abs().max()):.3e}', flush=True)
  if not torch.equal(o0, o1):
    d = (o0.float() - o1.float()).abs()
    print('  mismatches', int((d > 0).sum()), 'max', float(d.max()), 'first idx', (d > 0).nonzero()[:4].tolist())
    continue
  out = torch.empty_like(o0)
  fa = lambda: call(0, h, W1, b1, out, M, 4 * D, D)
  fb = lambda: call(32, h, W1, b1, out, M, 4 * D, D)
  for f in (fa, fb): tm(f, 3)
  ra, rb = [], []
  for rep in range(5):
    ra.append(tm(fa, 20)); rb.append(tm(fb, 20))
  a, b = sorted(ra)[2], sorted(rb)[2]
  fl = 2 * M * 4 * D * D
  print(f'n={n} mlp1  base {a:7.1f}us {fl/a/1e6:6.0f}TF   pp {b:7.1f}us {fl/b/1e6:6.0f}TF   pp/base {b/a:.3f}', flush=True)
