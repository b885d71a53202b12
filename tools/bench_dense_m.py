"""hipBLASLt time of the four 7B projections over the number of rows around a prompt step's 3 x 772 = 2316."""
import sys
sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd", "/root/repo/tools"]
import torch
from bench_kernels import timeit

MS = [int(a) for a in sys.argv[1:]] or [2048, 2304, 2316, 2432, 2560, 3088, 3860]


def main():
    shapes = [("qkv", 4096, 12288), ("o", 4096, 4096), ("gate_up", 4096, 22016), ("down", 11008, 4096)]
    for name, k, n in shapes:
        w = torch.randn(n, k, dtype=torch.bfloat16, device="cuda") * 0.02
        line = f"{name:8s}"
        for m in MS:
            x = torch.randn(m, k, dtype=torch.bfloat16, device="cuda")
            t = timeit(lambda: torch.nn.functional.linear(x, w), iters=30)
            line += f"  M={m}: {t * 1e6:6.1f} us ({2.0 * m * n * k / t / 1e15:4.2f} PF)"
        print(line, flush=True)

if __name__ == "__main__":
    main()
