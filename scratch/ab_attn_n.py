"""A/B several builds of libmdlm_hip.so on the attention kernel in ONE process:
python3 scratch/ab_attn_n.py libA.so libB.so ..."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from mdlm_amd._lib import SIGNATURES
libs = []
for path in sys.argv[1:]:
  l = C.CDLL(path)
  l.mdlm_attn_fwd.restype = C.c_int
  l.mdlm_attn_fwd.argtypes = SIGNATURES['mdlm_attn_fwd']
  libs.append(l)
torch.manual_seed(0)
H, L = 12, 1024
st = torch.cuda.current_stream().cuda_stream
for B in (40, 64):
  q = torch.randn(B, H, L, 64, device='cuda').bfloat16(); k = torch.randn_like(q)
  vt = torch.randn(B, H, 64, L, device='cuda').bfloat16()
  outs = [torch.empty(B * L, H * 64, device='cuda', dtype=torch.bfloat16) for _ in libs]
  def run(i):
    assert libs[i].mdlm_attn_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), outs[i].data_ptr(), B, H, L, 0, st) == 0
  def tm(i, iters=20):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
  for i in range(len(libs)): tm(i, 3)
  res = [[] for _ in libs]
  for rep in range(5):
    for i in range(len(libs)): res[i].append(tm(i))
  fl = 4 * L * L * 64 * H * B
  for i, r in enumerate(res):
    a = sorted(r)[2]
    d = (outs[0].float() - outs[i].float()).abs().max().item()
    print(f'attn B={B}: {sys.argv[1 + i]:32s} {a:7.1f}us {fl/a/1e6:5.0f}TF  maxdiff {d:.3g}', flush=True)
