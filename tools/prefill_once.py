"""A few launches of the prompt-attention kernel at one shape (for counter collection)."""
import sys
sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd", "/root/repo/tools"]
import torch
from splitwise_amd import _C

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
h, d, dtype = 32, 128, torch.bfloat16
s = d ** -0.5
qkv = torch.empty(L * n, 3 * h, d, dtype=dtype, device="cuda").uniform_(-s, s)
q, k, v = qkv.split([h, h, h], dim=1)
out = torch.empty(L * n, h, d, dtype=dtype, device="cuda")
st = torch.arange(n, dtype=torch.int32, device="cuda") * L
ln = torch.full((n,), L, dtype=torch.int32, device="cuda")
for _ in range(5):
    _C.ops.prefill_attention(out, q, k, v, st, ln, L, s)
torch.cuda.synchronize()
