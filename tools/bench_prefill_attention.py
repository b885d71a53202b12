import sys
sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd", "/root/repo/tools"]
import torch
from splitwise_amd import _C
from bench_kernels import timeit
def bench(lens, h=32, kvh=32, d=128, dtype=torch.bfloat16):
    n = sum(lens)
    s = d ** -0.5
    qkv = torch.empty(n, h + 2 * kvh, d, dtype=dtype, device="cuda").uniform_(-s, s)
    q, k, v = qkv.split([h, kvh, kvh], dim=1)
    out = torch.empty(n, h, d, dtype=dtype, device="cuda")
    starts, acc = [], 0
    for L in lens:
        starts.append(acc); acc += L
    st = torch.tensor(starts, dtype=torch.int32, device="cuda"); ln = torch.tensor(lens, dtype=torch.int32, device="cuda")
    t = timeit(lambda: _C.ops.prefill_attention(out, q, k, v, st, ln, max(lens), s), iters=50)
    flops = sum(4.0 * L * L * h * d / 2 for L in lens)
    print(f"prefill_attention lens={lens[0]}x{len(lens)} H={h} KVH={kvh} D={d}: {t*1e6:8.1f} us  {flops/t/1e12:7.1f} TFLOP/s (causal flops)")
if __name__ == "__main__":
    bench([772] * 3); bench([772]); bench([2048]); bench([512] * 8); bench([4096]); bench([2048], 64, 8); bench([1024] * 2, 40, 40)


def bench_context(new, ctx, h=32, kvh=32, d=128, bs=16, dtype=torch.bfloat16):
    """One sequence: `new` tokens attending to `ctx` cached tokens plus themselves."""
    s = d ** -0.5
    q = torch.empty(new, h, d, dtype=dtype, device="cuda").uniform_(-s, s)
    k = torch.empty(new, kvh, d, dtype=dtype, device="cuda").uniform_(-s, s)
    v = torch.empty(new, kvh, d, dtype=dtype, device="cuda").uniform_(-s, s)
    out = torch.empty_like(q)
    blocks = -(-ctx // bs)
    k_cache = torch.empty(blocks, kvh, d // 8, bs, 8, dtype=dtype, device="cuda").uniform_(-s, s)
    v_cache = torch.empty(blocks, kvh, d, bs, dtype=dtype, device="cuda").uniform_(-s, s)
    table = torch.randperm(blocks, device="cuda").to(torch.int32).view(1, -1)
    start = torch.tensor([0], device="cuda")
    total = torch.tensor([new + ctx], device="cuda")
    cl = torch.tensor([ctx], device="cuda")
    t = timeit(lambda: _C.ops.context_attention_fwd(q, k, v, out, k_cache, v_cache, table, start, total, cl, new),
               iters=30)
    flops = 4.0 * h * d * (new * ctx + new * new / 2)
    print(f"context_attention new={new} ctx={ctx} H={h} KVH={kvh} D={d}: {t*1e6:8.1f} us  {flops/t/1e12:7.1f} TFLOP/s")


if __name__ == "__main__":
    bench_context(772, 772); bench_context(512, 3584); bench_context(2048, 2048); bench_context(1024, 1024, 64, 8)
