"""A/B two builds on the fused update kernel (all rows masked): python3 scratch/ab_update.py libA.so libB.so"""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from mdlm_amd._lib import SIGNATURES
libs = []
for path in sys.argv[1:3]:
  l = C.CDLL(path)
  l.mdlm_subs_ddpm_update.restype = C.c_int
  l.mdlm_subs_ddpm_update.argtypes = SIGNATURES['mdlm_subs_ddpm_update']
  libs.append(l)
torch.manual_seed(0)
B, L, V = 64, 1024, 50258
Vp = (V + 127) // 128 * 128
logits = (torch.randn(B * L, Vp, device='cuda') * 2).bfloat16()
x = torch.full((B, L), V - 1, dtype=torch.int64, device='cuda')
t = torch.full((B,), 0.5, device='cuda')
lse = torch.zeros(B * L, device='cuda'); fresh = torch.ones(B, dtype=torch.int32, device='cuda')
outs = [torch.empty_like(x) for _ in libs]
changed = torch.zeros(B, dtype=torch.int32, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def run(i):
  rc = libs[i].mdlm_subs_ddpm_update(logits.data_ptr(), Vp, 1, lse.data_ptr(), fresh.data_ptr(), x.data_ptr(), t.data_ptr(), 1e-3, 0, 7, 11, outs[i].data_ptr(), changed.data_ptr(), B, L, V, V - 1, 1, st)
  assert rc == 0
def tm(i, iters=5):
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): run(i)
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters
for i in range(2): tm(i, 2)
res = [[], []]
for rep in range(5):
  for i in range(2): res[i].append(tm(i))
a, b = (sorted(r)[2] for r in res)
print(f'update all-masked: A {a:.3f} ms  B {b:.3f} ms  B/A {b/a:.3f}  equal {torch.equal(outs[0], outs[1])}  unmasked {(outs[1] != V - 1).sum().item()}')
