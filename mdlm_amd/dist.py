"""One process per GPU; `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" for CPU tests).  The hot path shards by sample / eval sequence
(SURVEY.md section 8(e)): nothing inside a sample crosses GPUs.  The only
collective on the data path is ONE all-reduce of the fp64 pair
[sum(nll * mask), sum(mask)] at the end of an eval epoch (the reference gets it
from torchmetrics `sync_dist=True`, diffusion.py:380-383, 117-125)."""
import os

import torch
import torch.distributed as dist


def env_rank():
  return (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)),
          int(os.environ.get('WORLD_SIZE', 1)))


def local_device(local_rank):
  """GPU index of this rank: LOCAL_RANK on a node with one GPU per rank.  With
  fewer GPUs than ranks (rehearsing N ranks on a one-GPU box over gloo) ranks
  wrap around the visible devices."""
  n = torch.cuda.device_count()
  return local_rank % n if n else 0


def init(backend=None):
  """Initialise the default process group from the torchrun environment.
  Returns (rank, local_rank, world_size).  No-op for world_size == 1.
  MDLM_DIST_BACKEND overrides the backend (gloo: rehearsals and CPU tests)."""
  rank, local_rank, world = env_rank()
  if world > 1 and not dist.is_initialized():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
      backend = os.environ.get('MDLM_DIST_BACKEND') or (
        'nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
      if world > torch.cuda.device_count():
        raise RuntimeError(f'{world} RCCL ranks need {world} GPUs '
                           f'({torch.cuda.device_count()} visible)')
      torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
  return rank, local_rank, world


def world_size():
  return dist.get_world_size() if dist.is_initialized() else 1


def rank():
  return dist.get_rank() if dist.is_initialized() else 0


def barrier():
  if dist.is_initialized():
    if dist.get_backend() == 'nccl':
      dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
      dist.barrier()


def shard_indices(n_items, rank_, world):
  """DistributedSampler-style sharding without shuffle (reference
  dataloader.py:679-694): item i goes to rank i % world; every rank gets the
  same count, the tail being padded by wrapping around."""
  per = (n_items + world - 1) // world
  total = per * world
  idx = list(range(n_items)) + list(range(total - n_items))
  return idx[rank_:total:world]


def rank_seed(seed, rank_):
  """The reference seeds every rank identically (main.py:190), so DDP ranks
  would emit identical samples; replicas here offset the seed per rank."""
  return int(seed) + int(rank_)


def _all_reduce(t, op):
  """In-place all-reduce.  Device tensors go straight to RCCL; under any other
  backend (gloo rehearsals of N ranks on one card) they are staged through
  host memory: gloo's own device path is unreliable when ranks share a GPU."""
  if t.is_cuda and dist.get_backend() != 'nccl':
    host = t.cpu()
    dist.all_reduce(host, op=op)
    t.copy_(host)
  else:
    dist.all_reduce(t, op=op)
  return t


def all_reduce_sums(sums):
  """SUM all-reduce of the (2,) fp64 metric vector, in place."""
  if dist.is_initialized() and dist.get_world_size() > 1:
    _all_reduce(sums, dist.ReduceOp.SUM)
  return sums


def max_over_ranks(value, device):
  t = torch.tensor([float(value)], dtype=torch.float64, device=device)
  if dist.is_initialized() and dist.get_world_size() > 1:
    _all_reduce(t, dist.ReduceOp.MAX)
  return t.item()


def gather_tokens(x):
  """Optional all_gather of (B, L) int64 samples (8 KiB/sample)."""
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return x
  if x.is_cuda and dist.get_backend() != 'nccl':   # see _all_reduce
    host = x.cpu()
    out = [torch.empty_like(host) for _ in range(dist.get_world_size())]
    dist.all_gather(out, host)
    return torch.cat(out, 0).to(x.device)
  out = [torch.empty_like(x) for _ in range(dist.get_world_size())]
  dist.all_gather(out, x)
  return torch.cat(out, 0)
