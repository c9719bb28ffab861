"""All result-identical knobs against each other on small shapes (tiny DiT): every combination must emit the
tokens of the plainest configuration (no cache, one stream, every row projected, lse in the update kernel)."""
import itertools, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'oracle')]
import test_gpu_diffusion as t
bad = 0
for B, L, steps in ((4, 128, 24), (8, 128, 48), (6, 256, 40), (5, 128, 30)):
  base = None
  combos = [dict(cache_policy='none', stream_groups=1, masked_rows_only='false', lse_from_gemm='false', fast_math='false')]
  for cp, g, mr, lg, fm in itertools.product(('sample', 'batch'), (1, 2, 3), ('true', 'false'), ('true', 'false'), ('true', 'false')):
    combos.append(dict(cache_policy=cp, stream_groups=g, masked_rows_only=mr, lse_from_gemm=lg, fast_math=fm))
  for c in combos:
    ov = [f'sampling.{k}={v}' for k, v in c.items()] + [f'loader.eval_batch_size={B}', f'model.length={L}']
    model, cfg, _ = t.make_model(ov)
    model.seed_rng(21)
    x = model._sample(num_steps=steps).cpu()
    if base is None:
      base = x
      continue
    if not torch.equal(base, x):
      bad += 1
      print('MISMATCH', B, L, c, int((base != x).sum()), 'tokens; groups used', len(model._state.get('groups') or [0]), flush=True)
  print(f'B={B} L={L}: {len(combos)} configurations done', flush=True)
print('mismatches:', bad)
