#!/usr/bin/env python3
"""Can a prompt step and decode steps share the GPU at the same time?  A prompt step is compute-bound (GEMMs at
2316 rows), a decode step HBM-bound (paged attention): on two streams one's matrix work could run beside the
other's memory traffic.  Llama-2-7B shapes, every layer with its own weights, each scenario one replayed HIP graph:
  prompt     : 32 layers of a prompt step (3 x 772 tokens)                       on one stream
  decode x2  : two decode steps (256 sequences of ~860 tokens), 32 layers each   on one stream
  serial     : prompt, then decode x2                                            on one stream (what the engine does)
  together   : prompt on one stream, decode x2 on another, joined at the end"""
import sys

sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd", "/root/repo/tools"]
import torch  # noqa: E402
from splitwise_amd import _C  # noqa: E402

DEV = "cuda"
H, D, BS, HIDDEN, FFN = 32, 128, 16, 4096, 11008
F = torch.nn.functional


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n, ctx, plen, nprompt = 256, 860, 772, 3
    nblk = (ctx + BS - 1) // BS
    kc = torch.empty(n * nblk + 1, H, D // 8, BS, 8, dtype=torch.bfloat16, device=DEV).uniform_(-0.1, 0.1)
    vc = torch.empty(n * nblk + 1, H, D, BS, dtype=torch.bfloat16, device=DEV).uniform_(-0.1, 0.1)
    bt = torch.randperm(n * nblk, device=DEV).to(torch.int32).view(n, nblk).contiguous()
    cl = torch.full((n,), ctx, dtype=torch.int32, device=DEV)
    weights = [[torch.randn(o, i, dtype=torch.bfloat16, device=DEV) * 0.02
                for (i, o) in ((HIDDEN, 3 * HIDDEN), (HIDDEN, HIDDEN), (HIDDEN, 2 * FFN), (FFN, HIDDEN))]
               for _ in range(layers)]
    norm_w = torch.ones(HIDDEN, dtype=torch.bfloat16, device=DEV)

    # decode state
    dq = torch.empty(n, H, D, dtype=torch.bfloat16, device=DEV).uniform_(-0.1, 0.1)
    dout = torch.empty_like(dq)
    dx = torch.randn(n, HIDDEN, dtype=torch.bfloat16, device=DEV) * 0.1
    dres = torch.randn_like(dx)
    dact = torch.empty(n, FFN, dtype=torch.bfloat16, device=DEV)

    def decode_step():
        for w in weights:
            _C.ops.fused_add_rms_norm(dx, dres, norm_w, 1e-5)
            F.linear(dx, w[0])
            _C.ops.paged_attention_v1(dout, dq, kc, vc, H, D ** -0.5, bt, cl, BS, ctx, None, "auto")
            a = F.linear(dout.view(n, HIDDEN), w[1])
            _C.ops.fused_add_rms_norm(a, dres, norm_w, 1e-5)
            g = F.linear(a, w[2])
            _C.ops.silu_and_mul(dact, g)
            F.linear(dact, w[3])

    # prompt state
    t = plen * nprompt
    px = torch.randn(t, HIDDEN, dtype=torch.bfloat16, device=DEV) * 0.1
    pres = torch.randn_like(px)
    pact = torch.empty(t, FFN, dtype=torch.bfloat16, device=DEV)
    pattn = torch.empty(t, H, D, dtype=torch.bfloat16, device=DEV)
    starts = torch.arange(nprompt, dtype=torch.int32, device=DEV) * plen
    lens = torch.full((nprompt,), plen, dtype=torch.int32, device=DEV)

    def prompt_step():
        for w in weights:
            _C.ops.fused_add_rms_norm(px, pres, norm_w, 1e-5)
            qkv = F.linear(px, w[0]).view(t, 3 * H, D)
            q, k, v = qkv.split([H, H, H], dim=1)
            _C.ops.prefill_attention(pattn, q, k, v, starts, lens, plen, D ** -0.5)
            a = F.linear(pattn.view(t, HIDDEN), w[1])
            _C.ops.fused_add_rms_norm(a, pres, norm_w, 1e-5)
            g = F.linear(a, w[2])
            _C.ops.silu_and_mul(pact, g)
            F.linear(pact, w[3])

    side = torch.cuda.Stream()

    def serial():
        prompt_step(); decode_step(); decode_step()

    def together():
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        prompt_step()
        with torch.cuda.stream(side):
            decode_step(); decode_step()
        main_s.wait_stream(side)

    def decode2():
        decode_step(); decode_step()

    for name, fn in (("prompt", prompt_step), ("decode x2", decode2), ("serial", serial), ("together", together)):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(3):
            g.replay()
        t1.record()
        torch.cuda.synchronize()
        print(f"{name:10s}: {t0.elapsed_time(t1) / 3:8.2f} ms", flush=True)


if __name__ == "__main__":
    main()
