import torch
def bench(fn, iters=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters
M = 65536
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072), (50304, 768)]:
  A = torch.randn(M, K, device='cuda').bfloat16(); W = (torch.randn(N, K, device='cuda') * 0.02).bfloat16()
  out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
  t = bench(lambda: torch.matmul(A, W.t(), out=out), iters=10)
  print(f'hipblaslt M={M} N={N} K={K}: {t*1e3:8.1f}us {2*M*N*K/t/1e9:7.0f} TF')
