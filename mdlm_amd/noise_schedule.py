"""Noise schedules on the hot path.  Only LogLinear is in scope:
`_ddpm_caching_update` asserts it (reference diffusion.py:593) and every MDLM
recipe uses it (configs/noise/loglinear.yaml).  Mirrors noise_schedule.py:37-43
(Noise.forward) and 126-151 (LogLinearNoise).  Tiny (B,)-sized torch math:
host-side plumbing, not a kernel."""
import torch
import torch.nn as nn


class LogLinearNoise(nn.Module):
  """sigma(t) = -log1p(-(1-eps) t);  sigma'(t) = (1-eps) / (1-(1-eps) t)."""

  def __init__(self, eps=1e-3):
    super().__init__()
    self.eps = eps
    self.sigma_max = self.total_noise(torch.tensor(1.0))
    self.sigma_min = self.eps + self.total_noise(torch.tensor(0.0))

  def rate_noise(self, t):
    return (1 - self.eps) / (1 - (1 - self.eps) * t)

  def total_noise(self, t):
    return -torch.log1p(-(1 - self.eps) * t)

  def forward(self, t):
    return self.total_noise(t), self.rate_noise(t)


def get_noise(config, dtype=torch.float32):
  """noise_schedule.py:13-28; schedules other than loglinear are out of scope
  for this hot path."""
  if config.noise.type == 'loglinear':
    return LogLinearNoise()
  raise ValueError(
    f'{config.noise.type}: only the loglinear schedule is implemented in '
    'mdlm_amd (the ddpm_cache sampler requires it)')
