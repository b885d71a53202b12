#!/usr/bin/env python3
"""Can the decode step's two halves overlap?  Decode attention is HBM-bound, the step's GEMMs are not: with the
batch cut into two micro-batches on two streams, one's attention could run beside the other's GEMMs.

Measures, per layer of Llama-2-7B at 256 sequences of `ctx` tokens (each weight matrix is a different one per
layer, so that weights come from HBM as in the model):
  serial    : attention(256) + GEMMs(256) on one stream                          (what a decode step does today)
  halves    : attention(128) + GEMMs(128), twice, on one stream                  (the price of splitting)
  ping-pong : two streams; attention of A, then of B, each ordered behind the other's by events; a micro-batch's
              GEMMs run on its own stream right after its attention
All three inside one replayed HIP graph of `layers` layers."""
import sys

sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd", "/root/repo/tools"]
import torch  # noqa: E402
from splitwise_amd import _C  # noqa: E402

DEV = "cuda"
H, D, BS, HIDDEN, FFN = 32, 128, 16, 4096, 11008


class Batch:
    def __init__(self, n, ctx, kc, vc, first_block):
        nblk = (ctx + BS - 1) // BS
        self.n, self.ctx = n, ctx
        self.bt = (first_block + torch.randperm(n * nblk, device=DEV)).to(torch.int32).view(n, nblk).contiguous()
        self.cl = torch.full((n,), ctx, dtype=torch.int32, device=DEV)
        self.q = torch.empty(n, H, D, dtype=torch.bfloat16, device=DEV).uniform_(-0.1, 0.1)
        self.out = torch.empty_like(self.q)
        self.x = torch.randn(n, HIDDEN, dtype=torch.bfloat16, device=DEV) * 0.1
        self.kc, self.vc = kc, vc

    def attention(self):
        _C.ops.paged_attention_v1(self.out, self.q, self.kc, self.vc, H, D ** -0.5, self.bt, self.cl, BS, self.ctx,
                                  None, "auto")

    def gemms(self, w):
        a = torch.nn.functional.linear(self.out.view(self.n, HIDDEN), w[1])      # o_proj
        g = torch.nn.functional.linear(a, w[2])                                   # gate_up
        d = torch.nn.functional.linear(g[:, :FFN], w[3])                          # down
        torch.nn.functional.linear(d, w[0])                                       # next layer's qkv


def main():
    ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 860
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    n = 256
    nblk = (ctx + BS - 1) // BS
    kc = torch.empty(n * nblk + 1, H, D // 8, BS, 8, dtype=torch.bfloat16, device=DEV).uniform_(-0.1, 0.1)
    vc = torch.empty(n * nblk + 1, H, D, BS, dtype=torch.bfloat16, device=DEV).uniform_(-0.1, 0.1)
    full = Batch(n, ctx, kc, vc, 0)
    half_a = Batch(n // 2, ctx, kc, vc, 0)
    half_b = Batch(n // 2, ctx, kc, vc, (n // 2) * nblk)
    weights = [[torch.randn(o, i, dtype=torch.bfloat16, device=DEV) * 0.02
                for (i, o) in ((HIDDEN, 3 * HIDDEN), (HIDDEN, HIDDEN), (HIDDEN, 2 * FFN), (FFN, HIDDEN))]
               for _ in range(layers)]

    def serial():
        for w in weights:
            full.attention()
            full.gemms(w)

    def halves():
        for w in weights:
            half_a.attention(); half_a.gemms(w)
            half_b.attention(); half_b.gemms(w)

    side = torch.cuda.Stream()

    def ping_pong():
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        ev_a = ev_b = None
        for w in weights:
            if ev_b is not None:
                main_s.wait_event(ev_b)                     # A's attention waits for B's previous attention
            half_a.attention()
            ev_a = torch.cuda.Event(); ev_a.record(main_s)
            half_a.gemms(w)
            with torch.cuda.stream(side):
                side.wait_event(ev_a)                       # B's attention waits for A's
                half_b.attention()
                ev_b = torch.cuda.Event(); ev_b.record(side)
                half_b.gemms(w)
        main_s.wait_stream(side)

    def free_run():
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        for w in weights:
            half_a.attention(); half_a.gemms(w)
        with torch.cuda.stream(side):
            half_b.gemms(weights[0])                         # half a layer of offset
            for w in weights:
                half_b.attention(); half_b.gemms(w)
        main_s.wait_stream(side)

    for name, fn in (("serial", serial), ("halves", halves), ("ping-pong", ping_pong), ("free-run", free_run)):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5):
            g.replay()
        t1.record()
        torch.cuda.synchronize()
        print(f"{name:10s} ctx={ctx}: {t0.elapsed_time(t1) * 1e3 / 5 / layers:8.1f} us per layer", flush=True)


if __name__ == "__main__":
    main()
