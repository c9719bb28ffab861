"""Shadow (moving-average) weights of the backbone, kept as ONE flat fp32 buffer.

What the hot path needs from the reference's EMA object (models/ema.py) is:
the checkpoint layout `{'decay', 'num_updates', 'shadow_params': [tensor, ...]}`
in `backbone.parameters()` order (models/ema.py:89-97, diffusion.py:168-181),
the decay warm-up `min(decay, (1 + n) / (10 + n))` (models/ema.py:40-44), and
"sample / validate on the averaged weights, then go back" (diffusion.py:399-406,
703-716).  This module provides those three on the memory layout the training
step already uses (`mdlm_amd.train.Trainer`): every parameter's shadow is a
256-byte aligned view of one buffer, so the Trainer adopts `flat` as the EMA
operand of its fused optimizer kernel without a copy, and the inference-side
swap is an in-place EXCHANGE of live and shadow values -- no third copy of the
model is held while the averaged weights are live.
"""
import torch

ALIGN = 64   # elements: every view starts 256-byte aligned (fp32)


def flat_layout(shapes):
  """[(offset, numel, shape)] of tensors packed at ALIGN-element boundaries,
  and the buffer length."""
  table, off = [], 0
  for shape in shapes:
    n = 1
    for d in shape:
      n *= int(d)
    table.append((off, n, tuple(shape)))
    off += (n + ALIGN - 1) // ALIGN * ALIGN
  return table, off


class ShadowWeights:
  def __init__(self, params, decay):
    decay = float(decay)
    if not 0.0 <= decay <= 1.0:
      raise ValueError(f'EMA decay {decay} is outside [0, 1]')
    ps = [p for p in params if p.requires_grad]
    self.decay = decay
    self.num_updates = 0
    self.table, self.n_flat = flat_layout([p.shape for p in ps])
    dev = ps[0].device if ps else torch.device('cpu')
    self.flat = torch.zeros(self.n_flat, dtype=torch.float32, device=dev)
    self.live_in_shadow = False   # True while the averaged weights are swapped in
    self.assign(ps)

  # ------------------------------------------------------------------ views
  def views(self):
    return [self.flat[o:o + n].view(s) for o, n, s in self.table]

  @property
  def shadow_params(self):
    """The per-parameter list of the checkpoint layout (views of `flat`)."""
    self._check_not_swapped('shadow_params')
    return self.views()

  @shadow_params.setter
  def shadow_params(self, tensors):
    self.assign(tensors)

  def assign(self, tensors):
    tensors = list(tensors)
    if len(tensors) != len(self.table):
      raise ValueError(f'{len(tensors)} shadow tensors for {len(self.table)} parameters')
    with torch.no_grad():
      for v, t in zip(self.views(), tensors):
        t = torch.as_tensor(t)
        if tuple(t.shape) != tuple(v.shape):
          raise ValueError(f'shadow tensor of shape {tuple(t.shape)} for a parameter of shape {tuple(v.shape)}')
        v.copy_(t.detach())

  def to(self, device):
    self.flat = self.flat.to(device)
    return self

  def _check_not_swapped(self, what):
    if self.live_in_shadow:
      raise RuntimeError(f'{what}: the averaged weights are swapped in (call swap_out first)')

  # ------------------------------------------------------------------- swaps
  @torch.no_grad()
  def _exchange(self, params):
    ps = [p for p in params if p.requires_grad]
    if len(ps) != len(self.table):
      raise ValueError(f'{len(ps)} parameters for {len(self.table)} shadow tensors')
    if self.flat.device != ps[0].device:
      self.to(ps[0].device)
    for v, p in zip(self.views(), ps):
      keep = p.data.clone()
      p.data.copy_(v)
      v.copy_(keep)

  def swap_in(self, params):
    self._check_not_swapped('swap_in')
    self._exchange(params)
    self.live_in_shadow = True

  def swap_out(self, params):
    if not self.live_in_shadow:
      raise RuntimeError('swap_out without swap_in')
    self._exchange(params)
    self.live_in_shadow = False

  # -------------------------------------------------------------- checkpoints
  def state_dict(self):
    self._check_not_swapped('state_dict')
    return {'decay': self.decay, 'num_updates': self.num_updates,
            'shadow_params': [v.detach().clone() for v in self.views()]}

  def load_state_dict(self, sd):
    self._check_not_swapped('load_state_dict')
    self.assign(sd['shadow_params'])
    self.decay = float(sd['decay'])
    self.num_updates = int(sd['num_updates'] or 0)
