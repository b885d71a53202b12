"""A few launches of the split form of prompt attention over a cached prefix (for kernel traces)."""
import sys
sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd", "/root/repo/tools"]
import torch
from splitwise_amd import _C

new, ctx = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 3584)
parts = int(sys.argv[3]) if len(sys.argv) > 3 else None
h, d, bs, dtype = 32, 128, 16, torch.bfloat16
s = d ** -0.5
q = torch.empty(new, h, d, dtype=dtype, device="cuda").uniform_(-s, s)
k = torch.empty(new, h, d, dtype=dtype, device="cuda").uniform_(-s, s)
v = torch.empty(new, h, d, dtype=dtype, device="cuda").uniform_(-s, s)
out = torch.empty_like(q)
blocks = -(-ctx // bs)
k_cache = torch.empty(blocks, h, d // 8, bs, 8, dtype=dtype, device="cuda").uniform_(-s, s)
v_cache = torch.empty(blocks, h, d, bs, dtype=dtype, device="cuda").uniform_(-s, s)
table = torch.randperm(blocks, device="cuda").to(torch.int32).view(1, -1)
start = torch.tensor([0], device="cuda")
total = torch.tensor([new + ctx], device="cuda")
cl = torch.tensor([ctx], device="cuda")
for _ in range(10):
    _C.ops.context_attention_fwd(q, k, v, out, k_cache, v_cache, table, start, total, cl, new, None, None, parts)
torch.cuda.synchronize()
