"""Bandwidth of the KV push kernel on one device (local HBM -> local HBM), by workgroup cap.

The cap is what a prompt worker uses towards a token worker on another GPU (32 by default): what one capped launch
sustains locally bounds what it can put on a link."""
import sys
sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd", "/root/repo/tools"]
import torch
from splitwise_amd import _C

def main():
    if len(sys.argv) > 1:                 # another build of the library, for A/B runs
        _C._LIB_PATH = sys.argv[1]
    dev = torch.device("cuda:0")
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    block = 1 << 20
    runs = [(i * 2 * block, block) for i in range(nbytes // (2 * block))]      # every other MiB: ragged like block runs
    stream = torch.cuda.current_stream().cuda_stream
    for cap in (0, 256, 64, 32, 16, 8):
        def one():
            for i in range(0, len(runs), 32):
                part = runs[i:i + 32]
                _C.kv_push(dst.data_ptr(), src.data_ptr(), [o for o, _ in part], [n for _, n in part], stream, cap)
        one()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            one()
        e1.record()
        e1.synchronize()
        secs = e0.elapsed_time(e1) * 1e-3 / 5
        moved = sum(n for _, n in runs)
        print(f"cap {cap:4d}: {moved / secs / 1e9:8.1f} GB/s  ({moved >> 20} MiB in {len(runs) // 32} launches)")
    ok = all(torch.equal(dst[o:o + n], src[o:o + n]) for o, n in runs[:8]) and int(dst[block:2 * block].sum()) == 0
    print("bytes", "ok" if ok else "WRONG")

if __name__ == "__main__":
    main()
