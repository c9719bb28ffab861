"""A/B builds of libmdlm_hip.so on ln_modulate in one process: python3 scratch/ab_ln.py libA.so libB.so ..."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdlm_amd._lib import SIGNATURES
libs = []
for path in sys.argv[1:]:
  l = C.CDLL(path)
  l.mdlm_ln_modulate.restype = C.c_int
  l.mdlm_ln_modulate.argtypes = SIGNATURES['mdlm_ln_modulate']
  libs.append(l)
torch.manual_seed(0)
D, L, CAP = 768, 1024, 64
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(CAP * L, D, device='cuda'); w = torch.randn(D, device='cuda')
mod = torch.randn(CAP, 6 * D, device='cuda')
outs = [torch.empty(CAP * L, D, device='cuda', dtype=torch.bfloat16) for _ in libs]
for n in (30, 40, 48, 64):
  nd = torch.tensor([n * L], device='cuda', dtype=torch.int32)
  def run(i):
    assert libs[i].mdlm_ln_modulate(x.data_ptr(), w.data_ptr(), mod.data_ptr(), mod[:, D:].data_ptr(), outs[i].data_ptr(),
                                    CAP * L, D, L, CAP, 6 * D, 0, nd.data_ptr(), st) == 0
  def tm(i, iters=50):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
  for i in range(len(libs)): tm(i, 5)
  res = [[] for _ in libs]
  for rep in range(5):
    for i in range(len(libs)): res[i].append(tm(i))
  by = n * L * D * 6
  for i, r in enumerate(res):
    a = sorted(r)[2]
    same = torch.equal(outs[0][:n * L], outs[i][:n * L])
    print(f'ln n={n}: {sys.argv[1 + i]:24s} {a:6.1f}us {by/a/1e6:5.2f}TB/s same={same}', flush=True)
