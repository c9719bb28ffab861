# Does splitting the batch over two HIP streams fill kernel tails?  Times the DiT
# forward for 2 x B/2 on one stream, 2 x B/2 on two streams, and 1 x B.
import sys, time, torch
sys.path.insert(0, '.')
from mdlm_amd.config import load_config
from bench import build_model
cfg = load_config(['model=small', 'loader.eval_batch_size=64', 'sampling.steps=1000', 'sampling.fast_math=true'])
mA = build_model(cfg, seed=1).backbone
mB = build_model(cfg, seed=2).backbone
V = mA.vocab_size if hasattr(mA, 'vocab_size') else 50258
def bench(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for half in (16, 20, 21, 22, 32):
  xa = torch.randint(0, 50257, (half, 1024), device='cuda')
  xb = torch.randint(0, 50257, (half, 1024), device='cuda')
  xab = torch.cat([xa, xb])
  sg = torch.zeros(half, device='cuda'); sg2 = torch.zeros(2 * half, device='cuda')
  oa = mA.forward(xa, sg); ob = mB.forward(xb, sg); oab = mA.forward(xab, sg2)
  def seq():
    mA.forward(xa, sg, out=oa); mB.forward(xb, sg, out=ob)
  def par():
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1): mA.forward(xa, sg, out=oa)
    with torch.cuda.stream(s2): mB.forward(xb, sg, out=ob)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
  def one():
    mA.forward(xab, sg2, out=oab)
  print(f'half={half}: seq {bench(seq):.3f} ms  two-stream {bench(par):.3f} ms  one-batch {bench(one):.3f} ms', flush=True)
