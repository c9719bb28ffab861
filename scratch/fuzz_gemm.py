"""GEMM epilogues against torch over odd shapes: M any multiple of 128, N with an edge, K multiples of 64, a
device-side live-row count, row / sample scatter lists, both tile configurations, lse partials."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdlm_amd import ops
from mdlm_amd._lib import EPI_STORE, EPI_BIAS_GELU, EPI_GATE_RESIDUAL
random.seed(0); torch.manual_seed(0)
dev = 'cuda'
def rel(a, b): return ((a - b).norm() / b.norm().clamp(min=1e-12)).item()
bad = 0
for case in range(60):
  M = 128 * random.randint(1, 12); K = 64 * random.randint(1, 20)
  tile = random.choice((0, 1, 16))
  kind = random.choice(('store_bf16', 'store_f32', 'gelu', 'gate', 'rowmap', 'samplemap'))
  N = random.choice((64, 128, 192, 256, 320, 768)) if kind == 'gate' else random.choice((8, 72, 130, 256, 300, 515, 1000, 1031))
  A = (torch.randn(M, K, device=dev)).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
  bias = torch.randn((N + 3) // 4 * 4, device=dev)[:N] if random.random() < 0.7 else None
  live_n = random.randint(0, M // 64) * 64
  live = torch.tensor([live_n], dtype=torch.int32, device=dev)
  ref = A.float() @ W.float().T + (bias if bias is not None else 0)
  name = f'{kind} M={M} N={N} K={K} tile={tile} live={live_n}'
  try:
    if kind in ('store_bf16', 'store_f32'):
      dt = torch.bfloat16 if kind == 'store_bf16' else torch.float32
      ld = (N + 7) // 8 * 8
      C = torch.full((M, ld), 7.0, dtype=dt, device=dev)
      ops.gemm_bf16(A, W, C[:, :N], EPI_STORE, bias=bias, m_rows_dev=live, tile=tile)
      e = rel(C[:live_n, :N].float(), ref[:live_n]) if live_n else 0.0
      TB = 256 if (tile & 1) == 0 and M % 256 == 0 and N >= 256 else 128
      dead0 = -(-live_n // TB) * TB
      ok = e < 1e-2 and bool((C[dead0:, :N] == 7.0).all()) and bool((C[:, N:] == 7.0).all())
    elif kind == 'gelu':
      ld = (N + 7) // 8 * 8
      C = torch.full((M, ld), 7.0, dtype=torch.bfloat16, device=dev)
      ops.gemm_bf16(A, W, C[:, :N], EPI_BIAS_GELU, bias=bias, m_rows_dev=live, tile=tile)
      e = rel(C[:live_n, :N].float(), torch.nn.functional.gelu(ref[:live_n], approximate='tanh')) if live_n else 0.0
      ok = e < 1e-2 and bool((C[:, N:] == 7.0).all())
    elif kind == 'gate':
      L = 64 * random.randint(1, 4)
      if M % L: L = 64
      n_mod = random.choice((1, M // L))
      mod = torch.randn(n_mod, 3 * N, device=dev)
      x0 = torch.randn(M, N, device=dev); x = x0.clone()
      ops.gemm_bf16(A, W, x, EPI_GATE_RESIDUAL, bias=bias, gate=mod[:, N:2 * N], residual=x, rows_per_mod=L,
                    n_mod=n_mod, mod_ld=3 * N, m_rows_dev=live, tile=tile)
      g = mod[:, N:2 * N][(torch.arange(M, device=dev) // L) % n_mod]
      want = x0 + g * ref
      e = rel(x[:live_n], want[:live_n]) if live_n else 0.0
      ok = e < 1e-2
    elif kind == 'rowmap':
      if M % 256: M2 = M
      rows_out = M + 64
      perm = torch.randperm(rows_out, device=dev)[:M].to(torch.int32)
      perm[torch.rand(M, device=dev) < 0.2] = -1
      ld = (N + 7) // 8 * 8
      C = torch.full((rows_out, ld), 7.0, dtype=torch.bfloat16, device=dev)
      lsep = None
      use_lse = M % 256 == 0 and random.random() < 0.7
      if use_lse:
        lsep = torch.zeros(ops.workspace_bytes('gemm_lse_partials', M, N) // 4, device=dev)
      skip = random.choice((-1, N - 1, 0))
      ops.gemm_bf16(A, W, C[:, :N], EPI_STORE, bias=bias, m_rows_dev=live, out_row_map=perm, tile=tile,
                    lse_partials=lsep, lse_skip_col=skip if use_lse else -1)
      want = torch.full_like(C, 7.0)
      sel = (perm[:live_n] >= 0)
      want[perm[:live_n][sel].long(), :N] = ref[:live_n][sel].bfloat16()
      TB = 256 if (tile & 1) == 0 and M % 256 == 0 and N >= 256 else 128
      dead0 = -(-live_n // TB) * TB
      # rows of a partly live tile past the live count may be written too (their list entries are the caller's padding)
      maybe = perm[live_n:dead0]; maybe = maybe[maybe >= 0].long()
      mask_rows = torch.ones(rows_out, dtype=torch.bool, device=dev); mask_rows[maybe] = False
      e = rel(C[mask_rows].float(), want[mask_rows].float())
      ok = e < 1e-2
      if use_lse and live_n:
        lse = torch.zeros(M, device=dev)
        ops.lse_combine(lsep, M, N, lse, n_rows_dev=live)
        y = ref[:live_n].bfloat16().float()
        if skip >= 0: y[:, skip] = -1e30
        want_lse = torch.logsumexp(y, -1)
        el = (lse[:live_n] - want_lse).abs().max().item()
        ok = ok and el < 2e-2
        e = max(e, el)
    else:  # samplemap
      L = 64 * random.randint(1, 4)
      if M % L: L = 64
      n_slots = M // L
      n_samples = n_slots + 2
      amap = torch.randperm(n_samples, device=dev)[:n_slots].to(torch.int32)
      live_s = random.randint(0, n_slots); live_n = live_s * L
      live = torch.tensor([live_n], dtype=torch.int32, device=dev)
      ld = (N + 7) // 8 * 8
      C = torch.full((n_samples * L, ld), 7.0, dtype=torch.bfloat16, device=dev)
      ops.gemm_bf16(A, W, C[:, :N], EPI_STORE, bias=bias, rows_per_mod=L, m_rows_dev=live, out_sample_map=amap, tile=tile)
      want = torch.full_like(C, 7.0)
      for s in range(live_s):
        want[amap[s] * L:(amap[s] + 1) * L, :N] = ref[s * L:(s + 1) * L].bfloat16()
      e = rel(C.float(), want.float())
      ok = e < 1e-2
      name += f' L={L}'
    if not ok:
      bad += 1; print('FAIL', name, e, flush=True)
  except Exception as ex:
    msg = str(ex)[:160]
    print('ERR ', name, type(ex).__name__, msg, flush=True)
print('failures:', bad)
