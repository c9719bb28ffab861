"""The entry point that allocates nothing and takes no workspace
(`mdlm_grad_sumsq_static`, include/mdlm_hip.h): called through the C-ABI and
checked bit for bit against `mdlm_grad_sumsq` (same kernels; the partial sums
live in a file-scope device array instead of a caller buffer), on two calls in a
row -- the second with a smaller input, so anything the first call left in the
shared array would show."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(g):
  from mdlm_amd._lib import call
  st = torch.cuda.current_stream().cuda_stream
  part = torch.full((1024,), float('nan'), device='cuda')
  ref = torch.zeros(1, device='cuda')
  out = torch.full((1,), float('nan'), device='cuda')
  call('mdlm_grad_sumsq', g.data_ptr(), g.numel(), part.data_ptr(), ref.data_ptr(), st)
  free0, _ = torch.cuda.mem_get_info()
  call('mdlm_grad_sumsq_static', g.data_ptr(), g.numel(), out.data_ptr(), st)
  torch.cuda.synchronize()
  free1, _ = torch.cuda.mem_get_info()
  assert free1 == free0          # no device allocation behind the call
  return out.cpu(), ref.cpu()


def test_grad_sumsq_static_matches_workspace_variant():
  assert torch.cuda.is_available()
  torch.manual_seed(0)
  big = torch.randn(3_000_017, device='cuda')
  out, ref = _both(big)
  assert torch.equal(out, ref)
  assert abs(out.item() / big.double().pow(2).sum().item() - 1) < 1e-5
  small = torch.randn(777, device='cuda') * 3     # fewer elements than partials
  out, ref = _both(small)
  assert torch.equal(out, ref)
  assert abs(out.item() / small.double().pow(2).sum().item() - 1) < 1e-5
  out, ref = _both(small[:1])                     # a single element
  assert torch.equal(out, ref)


def test_grad_sumsq_static_rejects_oversize():
  from mdlm_amd._lib import call
  g = torch.zeros(8, device='cuda')
  out = torch.zeros(1, device='cuda')
  with pytest.raises(RuntimeError, match='exceeds the supported maximum'):
    call('mdlm_grad_sumsq_static', g.data_ptr(), (1 << 34) + 1, out.data_ptr(),
         torch.cuda.current_stream().cuda_stream)
