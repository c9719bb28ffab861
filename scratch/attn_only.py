import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdlm_amd import ops
torch.manual_seed(0)
H, L, B = 12, 1024, 64
q = torch.randn(B, H, L, 64, device='cuda').bfloat16(); k = torch.randn_like(q); vt = torch.randn(B, H, 64, L, device='cuda').bfloat16()
out = torch.empty(B * L, H * 64, device='cuda', dtype=torch.bfloat16)
for _ in range(5): ops.attn_fwd(q, k, vt, out, B, H, L)
torch.cuda.synchronize()
