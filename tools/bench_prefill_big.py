import sys
sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd", "/root/repo/tools"]
from bench_prefill_attention import bench
bench([2048] * 16, 64, 8); bench([4096] * 4, 32, 32); bench([1024] * 16, 32, 32); bench([772] * 24, 32, 32)
