"""Vocabulary GEMM with lse partials: two builds, with a scatter row map like the sampling loop."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from mdlm_amd._lib import SIGNATURES, EPI_STORE
libs = []
for path in sys.argv[1:]:
  l = C.CDLL(path); l.mdlm_gemm_bf16.restype = C.c_int; l.mdlm_gemm_bf16.argtypes = SIGNATURES['mdlm_gemm_bf16']; libs.append(l)
torch.manual_seed(0)
D, V, M = 768, 50258, 22016
Vp = (V + 127) // 128 * 128
h = torch.randn(M, D, device='cuda').bfloat16()
Wv = (torch.randn(V, D, device='cuda') * 0.08).bfloat16(); bv = torch.randn(V, device='cuda') * 0.02
logits = torch.empty(65536, Vp, dtype=torch.bfloat16, device='cuda')
part = torch.empty(M * ((V + 63) // 64) * 2, device='cuda')
scatter = torch.sort(torch.randperm(65536, device='cuda')[:M]).values.to(torch.int32)
live = torch.tensor([M], dtype=torch.int32, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def run(l, with_part):
  rc = l.mdlm_gemm_bf16(h.data_ptr(), D, Wv.data_ptr(), bv.data_ptr(), logits.data_ptr(), Vp, 1, M, V, D, EPI_STORE, 0, 0, 1, 1, 0, 0, 0, 0,
                        live.data_ptr(), 0, scatter.data_ptr(), part.data_ptr() if with_part else 0, V - 1, 0, st)
  assert rc == 0
def tm(l, wp, iters=6):
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): run(l, wp)
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3
for i, l in enumerate(libs):
  for wp in (False, True):
    tm(l, wp, 2)
    r = sorted(tm(l, wp) for _ in range(5))[2]
    print(f'{sys.argv[1 + i]:28s} partials={wp}: {r:8.1f} us', flush=True)
