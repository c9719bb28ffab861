"""Semi-AR driver against the oracle loop (GPU logits, materialised Philox uniforms) over stride lengths that do
and do not divide L, stride counts, dt and batch sizes (stream groups on where the batch allows)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'oracle')]
import test_gpu_diffusion as t
from mdlm_amd import ops
bad = 0
for n, stride, ns, inv_dt, L in ((2, 64, 2, 8, 128), (3, 48, 1, 6, 128), (4, 100, 3, 5, 128), (8, 32, 2, 4, 128), (2, 1, 2, 4, 128), (4, 128, 1, 7, 256), (2, 127, 1, 3, 128)):
  model, cfg, p = t.make_model([f'loader.eval_batch_size={n}', f'model.length={L}'])
  model.seed_rng(31)
  steps, texts, seqlens = model.sample_subs_guidance(n_samples=n, stride_length=stride, num_strides=ns, dt=1 / inv_dt)
  toks = torch.from_numpy(model.last_tokens)
  od = t.GpuBackedOracle(model, ops, p, cfg.model.n_heads, cfg.model.n_blocks, t.V, t.MASK, L)
  od.seed = 31
  od.rand = lambda *shape: od.draw(*shape)
  ref_steps, ref_toks = od.sample_subs_guidance(n, stride, ns, dt=1 / inv_dt)
  ok = torch.equal(toks, ref_toks) and steps == ref_steps
  print(dict(n=n, stride=stride, strides=ns, dt=f'1/{inv_dt}', L=L), 'groups', len(model._state['groups']), 'steps', steps, ref_steps,
        'OK' if ok else f'DIFFERENT ({int((toks != ref_toks).sum())} tokens)', flush=True)
  bad += not ok
print('failures:', bad)
