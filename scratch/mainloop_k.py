"""Main-loop-only rate (ablation build, epilogue skipped) against K and the epilogue instantiation, N = 768."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdlm_amd._lib import SIGNATURES, EPI_STORE, EPI_GATE_RESIDUAL
l = C.CDLL(sys.argv[1]); l.mdlm_gemm_bf16.restype = C.c_int; l.mdlm_gemm_bf16.argtypes = SIGNATURES['mdlm_gemm_bf16']
torch.manual_seed(0)
M, D, L = 65536, 768, 1024
dev = 'cuda'; st = torch.cuda.current_stream().cuda_stream
x = torch.randn(M, D, device=dev); xb = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
mod = torch.randn(1, 6 * D, device=dev)
for K in (768, 1536, 3072, 6144):
  a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(D, K, device=dev) * 0.02).bfloat16()
  for name, epi, Cc, dt in (('gate_residual fp32', EPI_GATE_RESIDUAL, x, 0), ('store bf16', EPI_STORE, xb, 1)):
    for mode in (0, 2):
      def run():
        rc = l.mdlm_gemm_bf16(a.data_ptr(), K, w.data_ptr(), 0, Cc.data_ptr(), D, dt, M, D, K, epi,
                              mod.data_ptr() if epi == EPI_GATE_RESIDUAL else 0, x.data_ptr() if epi == EPI_GATE_RESIDUAL else 0,
                              L, 1, 6 * D, 0, 0, 0, 0, 0, 0, 0, -1, mode, st)
        assert rc == 0
      for _ in range(3): run()
      ts = []
      for rep in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10 * 1e3)
      t = sorted(ts)[2]
      print(f'K={K:5d} {name:18s} {"main loop only" if mode else "whole kernel  "} {t:7.1f} us {2 * M * D * K / t / 1e6:6.0f} TF', flush=True)
