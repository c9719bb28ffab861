"""gate/residual GEMMs: the two-accumulator-set kernel (tile | 32) against the 256x256
kernel, one library, one process, interleaved launches; bit-identity checked first.
  python3 scratch/ab_pp.py [lib.so] [n_samples ...]     (dbg bits via MODE env: 2 = no epilogue)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdlm_amd._lib import SIGNATURES, EPI_GATE_RESIDUAL
path = sys.argv[1] if len(sys.argv) > 1 else 'mdlm_amd/libmdlm_hip.so'
l = C.CDLL(os.path.abspath(path))
l.mdlm_gemm_bf16.restype = C.c_int
l.mdlm_gemm_bf16.argtypes = SIGNATURES['mdlm_gemm_bf16']
l.mdlm_last_error_string.restype = C.c_char_p
ns = [int(a) for a in sys.argv[2:]] or [64, 40]
dbg = int(os.environ.get('MODE', 0))
torch.manual_seed(0)
L, D = 1024, 768
dev = 'cuda'
P = lambda t: t.data_ptr() if t is not None else 0
st = torch.cuda.current_stream().cuda_stream
def call(mode, A, W, bias, res, out, gate, M, N, K):
  rc = l.mdlm_gemm_bf16(P(A), K, P(W), P(bias), P(out), N, 0, M, N, K, EPI_GATE_RESIDUAL, P(gate), P(res),
                        L, 1, 6 * D, 0, 0, 0, 0, 0, 0, 0, -1, mode | dbg, st)
  assert rc == 0, l.mdlm_last_error_string()
def tm(fn, iters):
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3
for n in ns:
  M = n * L
  h = torch.randn(M, D, device=dev).bfloat16(); mh = torch.randn(M, 4 * D, device=dev).bfloat16()
  x = torch.randn(M, D, device=dev)
  Wo = (torch.randn(D, D, device=dev) * 0.02).bfloat16()
  W2 = (torch.randn(D, 4 * D, device=dev) * 0.02).bfloat16(); b2 = torch.randn(D, device=dev) * 0.02
  mod = torch.randn(1, 6 * D, device=dev)
  for name, A, W, bias, K in (('out', h, Wo, None, D), ('mlp2', mh, W2, b2, 4 * D)):
    if not dbg:
      o0 = torch.empty_like(x); o1 = torch.empty_like(x)
      call(0, A, W, bias, x, o0, mod[:, :D], M, D, K)
      call(32, A, W, bias, x, o1, mod[:, :D], M, D, K)
      torch.cuda.synchronize()
      ref = x + mod[0, :D] * (A.float() @ W.float().T + (bias if bias is not None else 0))
      print(f'n={n} {name}: identical to 256x256 kernel: {torch.equal(o0, o1)}  max err vs fp32 {float((o1 - ref).abs().max()):.3e}')
    out = torch.empty_like(x)
    fa = lambda: call(0, A, W, bias, x, out, mod[:, :D], M, D, K)
    fb = lambda: call(32, A, W, bias, x, out, mod[:, :D], M, D, K)
    for f in (fa, fb): tm(f, 3)
    ra, rb = [], []
    for rep in range(5):
      ra.append(tm(fa, 20)); rb.append(tm(fb, 20))
    a, b = sorted(ra)[2], sorted(rb)[2]
    fl = 2 * M * D * K
    print(f'n={n} {name:5s} base {a:7.1f}us {fl/a/1e6:6.0f}TF   pp {b:7.1f}us {fl/b/1e6:6.0f}TF   pp/base {b/a:.3f}')
