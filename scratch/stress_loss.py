"""NELBO path: the masked-rows projection and the row lse from the GEMM epilogue against the plain path
(every row projected, lse inside subs_nll) on small shapes, supplied t and u."""
import itertools, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'oracle')]
import test_gpu_diffusion as t
worst = 0.0
for B, L in ((4, 128), (3, 128), (2, 256), (5, 256), (8, 128)):
  torch.manual_seed(B * 1000 + L)
  x0 = torch.randint(0, t.V - 1, (B, L)).cuda()
  tt = (torch.rand(B) * 0.98 + 0.01).cuda()
  u = torch.rand(B, L).cuda()
  am = torch.ones(B, L).cuda(); am[:, L - 7:] = 0
  base = None
  for mr, lg in itertools.product(('false', 'true'), ('false', 'true')):
    model, cfg, _ = t.make_model([f'sampling.masked_rows_only={mr}', f'sampling.lse_from_gemm={lg}',
                                  f'loader.eval_batch_size={B}', f'model.length={L}'])
    for rep in range(2):   # second call reuses the eval state
      out = model._loss(x0, am, t=tt, u=u)
    n = out.nlls.float().cpu()
    if base is None: base = n
    d = (n - base).abs().max().item()
    worst = max(worst, d)
    print(B, L, mr, lg, 'loss', float(out.loss), 'max |dnll|', d, 'nan', bool(torch.isnan(n).any()), flush=True)
print('worst', worst)
