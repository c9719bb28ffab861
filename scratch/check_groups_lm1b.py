"""stream_groups=2 against 1 at the LM1B shapes (L = 128, bert vocabulary 30,522, mask id 103 in-vocab)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mdlm_amd.config import load_config
outs = []
for g in (1, 2):
  cfg = load_config(['model=small', 'data=lm1b', 'model.length=128', 'loader.eval_batch_size=64', 'sampling.steps=128',
                     f'sampling.stream_groups={g}'])
  model = bench.build_model(cfg, seed=1)
  model.seed_rng(5)
  x = model.restore_model_and_sample(num_steps=128)
  torch.cuda.synchronize()
  outs.append((x.cpu(), int(model.last_nfe.item()), len(model._state['groups'])))
  print(g, 'groups used', outs[-1][2], 'nfe', outs[-1][1], 'mask id', model.mask_index, 'V', model.vocab_size, 'masks left', int((x == model.mask_index).sum()), flush=True)
print('identical', torch.equal(outs[0][0], outs[1][0]), outs[0][1] == outs[1][1])
