"""Trainer.forward_backward against the oracle's autograd at batch shapes other than the tests' 4 x 128."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'oracle')]
import test_gpu_train as T
import mdlm_oracle as orc
for B, L in ((2, 128), (3, 128), (6, 128), (8, 128), (2, 256), (5, 256), (16, 128)):
  try:
    model, cfg, p, tr = T._make_trainer(0.0, extra=[f'model.length={L}', f'loader.eval_batch_size={B}'])
    m = cfg.model
    x0, attn, t, u = T._batch(B, L, seed=B * 10 + L)
    loss, nlls = tr.forward_backward(x0.cuda(), attn.cuda(), t=t.cuda(), u_xt=u.cuda())
    ref_loss, ref_nlls, ref_g = orc.loss_and_grads(p, x0, attn, t, u, m.n_heads, m.n_blocks, T.MASK_T, None, 0.0)
    g = tr.grads()
    worst = max((T._rel(g[k], ref_g[k]), k) for k in ref_g)
    flat = torch.cat([g[k].reshape(-1).cpu() for k in ref_g]).double(); fr = torch.cat([ref_g[k].reshape(-1) for k in ref_g]).double()
    cos = float((flat @ fr) / (flat.norm() * fr.norm()))
    print(f'B={B} L={L}: loss {loss.item():.4f} ref {ref_loss.item():.4f}  nll rel {T._rel(nlls, ref_nlls):.2e}  cos {cos:.6f}  worst {worst[0]:.3f} {worst[1]}', flush=True)
  except Exception as e:
    print(f'B={B} L={L}: {type(e).__name__}: {str(e)[:200]}', flush=True)
