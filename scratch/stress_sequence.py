"""State reuse: one model runs sample / semi-AR / loss / API-level update calls in sequence; every result must equal
that of a fresh model given the same Philox seed."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'oracle')]
import test_gpu_diffusion as t
B, L = 8, 128
ov = [f'loader.eval_batch_size={B}', f'model.length={L}']
torch.manual_seed(0)
x0 = torch.randint(0, t.V - 1, (B, L)).cuda(); tt = (torch.rand(B) * 0.98 + 0.01).cuda(); u = torch.rand(B, L).cuda()
xt = x0.clone(); xt[torch.rand(B, L).cuda() < 0.5] = t.MASK
def ops_list(model):
  def sample40(): model.seed_rng(1); return model._sample(num_steps=40).cpu()
  def semi(): model.seed_rng(2); return model.sample_subs_guidance(n_samples=B, stride_length=64, num_strides=1, dt=1 / 16)[1]
  def sample24(): model.seed_rng(3); return model._sample(num_steps=24).cpu()
  def loss(): return model._loss(x0, None, t=tt, u=u).nlls.float().cpu()
  def api():
    model.seed_rng(4)
    p, xn = model._ddpm_caching_update(xt, torch.full((B, 1), 0.5).cuda(), 0.01)
    p2, xn2 = model._ddpm_caching_update(xn, torch.full((B, 1), 0.49).cuda(), 0.01, p_x0=p)
    return xn2.cpu()
  def fwd(): return model.forward(xt, torch.zeros(B).cuda()).float().cpu()
  return dict(sample40=sample40, semi=semi, sample24=sample24, loss=loss, api=api, fwd=fwd)
ref = {}
for name in ('sample40', 'semi', 'sample24', 'loss', 'api', 'fwd'):
  model, _, _ = t.make_model(ov)
  ref[name] = ops_list(model)[name]()
model, _, _ = t.make_model(ov)
fns = ops_list(model)
bad = 0
for name in ('sample40', 'semi', 'loss', 'sample24', 'api', 'fwd', 'semi', 'sample40', 'api', 'loss', 'sample24', 'fwd', 'sample40'):
  out = fns[name]()
  same = (out == ref[name]) if not torch.is_tensor(out) else (torch.equal(out, ref[name]) if out.dtype != torch.float32 else bool((out - ref[name]).abs().max() < 1e-5))
  print(name, 'same' if same else 'DIFFERENT', flush=True)
  bad += not same
print('different:', bad)
