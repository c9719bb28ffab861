"""Inference-side subset of models/ema.py: holds shadow parameters restored
from a checkpoint (ema.py:89-97) and swaps them in/out around sampling and
validation (ema.py:52-87; diffusion.py:399-406, 703-716)."""
import torch


class ExponentialMovingAverage:
  def __init__(self, parameters, decay):
    if decay < 0.0 or decay > 1.0:
      raise ValueError('Decay must be between 0 and 1')
    self.decay = decay
    self.num_updates = 0
    self.shadow_params = [p.clone().detach() for p in parameters
                          if p.requires_grad]
    self.collected_params = []

  def move_shadow_params_to_device(self, device):
    self.shadow_params = [p.to(device) for p in self.shadow_params]

  def copy_to(self, parameters):
    params = [p for p in parameters if p.requires_grad]
    for s, p in zip(self.shadow_params, params):
      p.data.copy_(s.data)

  def store(self, parameters):
    self.collected_params = [p.clone() for p in parameters]

  def restore(self, parameters):
    for c, p in zip(self.collected_params, parameters):
      p.data.copy_(c.data)

  def state_dict(self):
    return dict(decay=self.decay, num_updates=self.num_updates,
                shadow_params=self.shadow_params)

  def load_state_dict(self, state_dict):
    self.decay = state_dict['decay']
    self.num_updates = state_dict['num_updates']
    self.shadow_params = [torch.as_tensor(p) for p in state_dict['shadow_params']]
