"""CPU baseline for the serving benchmark: the Llama forward pass restated on the CPU with the
oracle's own operators (reshape_and_cache, prefill attention, paged decode attention) and
torch CPU GEMMs, timed on a bounded sample and scaled to the benchmark's unit.

The reference has no CPU execution path (vllm/config.py:523-541 raises without a GPU), so the
"CPU path" is this restatement; it is a reported baseline, not an optimisation target.
"""
import time

import torch
import torch.nn.functional as F

from . import attention as oattn
from . import cache_ops as ocache

_GEOMETRY = {
    "llama-2-7b": dict(hidden=4096, heads=32, kv_heads=32, inter=11008, layers=32, vocab=32000),
    "llama-2-13b": dict(hidden=5120, heads=40, kv_heads=40, inter=13824, layers=40, vocab=32000),
    "llama-2-70b": dict(hidden=8192, heads=64, kv_heads=8, inter=28672, layers=80, vocab=32000),
    "llama-tiny": dict(hidden=256, heads=4, kv_heads=2, inter=512, layers=2, vocab=512),
}


def _rms(x, eps=1e-5):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)


def _rope(x, heads, pos):
    n, d = x.shape[0], x.shape[1] // heads
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float) / d))
    f = pos.float()[:, None] * inv[None, :]
    cos, sin = f.cos()[:, None, :], f.sin()[:, None, :]
    x = x.view(n, heads, d)
    x1, x2 = x[..., :d // 2], x[..., d // 2:]
    return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), -1).reshape(n, heads * d)


class CpuLlama:
    def __init__(self, geometry: dict, layers: int, num_blocks: int, block_size: int = 16, seed: int = 0):
        g = torch.Generator().manual_seed(seed)
        self.g = geometry
        h, kvh, hid, inter = geometry["heads"], geometry["kv_heads"], geometry["hidden"], geometry["inter"]
        self.d = hid // h
        self.bs = block_size
        def w(*shape):
            return (torch.randn(*shape, generator=g) * 0.02).to(torch.bfloat16).float()
        self.layers = [dict(qkv=w((h + 2 * kvh) * self.d, hid), o=w(hid, hid), gate_up=w(2 * inter, hid),
                            down=w(hid, inter)) for _ in range(layers)]
        self.embed = w(geometry["vocab"], hid)
        self.lm_head = w(geometry["vocab"], hid)
        self.cache = [(torch.zeros(num_blocks, kvh, self.d // 8, block_size, 8, dtype=torch.bfloat16),
                       torch.zeros(num_blocks, kvh, self.d, block_size, dtype=torch.bfloat16))
                      for _ in range(layers)]

    def _layer(self, li, x, pos, slots, attend):
        h, kvh, d = self.g["heads"], self.g["kv_heads"], self.d
        L = self.layers[li]
        y = _rms(x)
        qkv = y @ L["qkv"].t()
        q, k, v = qkv.split([h * d, kvh * d, kvh * d], -1)
        q, k = _rope(q, h, pos), _rope(k, kvh, pos)
        kc, vc = self.cache[li]
        ocache.reshape_and_cache(k.view(-1, kvh, d).bfloat16(), v.view(-1, kvh, d).bfloat16(), kc, vc, slots)
        a = attend(q.view(-1, h, d).bfloat16(), k.view(-1, kvh, d).bfloat16(), v.view(-1, kvh, d).bfloat16(), kc, vc)
        x = x + a.float().view(x.shape[0], h * d) @ L["o"].t()
        y = _rms(x)
        gate, up = (y @ L["gate_up"].t()).chunk(2, -1)
        return x + (F.silu(gate) * up) @ L["down"].t()

    def prefill(self, tokens, table):
        n = len(tokens)
        pos = torch.arange(n)
        slots = torch.tensor([table[i // self.bs] * self.bs + i % self.bs for i in range(n)])
        x = self.embed[torch.tensor(tokens)]
        scale = self.d ** -0.5
        for li in range(len(self.layers)):
            x = self._layer(li, x, pos, slots,
                            lambda q, k, v, kc, vc: oattn.prefill_attention(q, k, v, [n], scale))
        return int((_rms(x[-1:]) @ self.lm_head.t()).argmax())

    def decode(self, last_tokens, lens, tables):
        """One decode step for a batch: lens = sequence lengths including the token being fed."""
        b = len(last_tokens)
        pos = torch.tensor([n - 1 for n in lens])
        slots = torch.tensor([t[(n - 1) // self.bs] * self.bs + (n - 1) % self.bs for t, n in zip(tables, lens)])
        width = max(len(t) for t in tables)
        bt = torch.tensor([t + [0] * (width - len(t)) for t in tables], dtype=torch.int32)
        cl = torch.tensor(lens, dtype=torch.int32)
        x = self.embed[torch.tensor(last_tokens)]
        scale = self.d ** -0.5
        for li in range(len(self.layers)):
            x = self._layer(li, x, pos, slots,
                            lambda q, k, v, kc, vc: oattn.paged_attention(q, kc, vc, bt, cl, scale))
        return (_rms(x) @ self.lm_head.t()).argmax(-1).tolist()


def timed_sample(model_name: str, prompt_len: int, decode_len: int, layers: int = 2, num_requests: int = 2,
                 decode_steps: int = 4, budget_s: float = 25.0, seed: int = 0) -> dict:
    geo = _GEOMETRY[model_name]
    bs = 16
    blocks_per_seq = -(-(prompt_len + decode_steps + 1) // bs)
    model = CpuLlama(geo, layers, num_requests * blocks_per_seq, bs, seed)
    g = torch.Generator().manual_seed(seed)
    tables = [list(range(i * blocks_per_seq, (i + 1) * blocks_per_seq)) for i in range(num_requests)]
    prompts = [torch.randint(0, geo["vocab"], (prompt_len,), generator=g).tolist() for _ in range(num_requests)]
    t0 = time.perf_counter()
    last = [model.prefill(p, t) for p, t in zip(prompts, tables)]
    t_prefill = (time.perf_counter() - t0) / num_requests
    lens = [prompt_len + 1] * num_requests
    t0 = time.perf_counter()
    done = 0
    for _ in range(decode_steps):
        last = model.decode(last, lens, tables)
        lens = [n + 1 for n in lens]
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    t_decode = (time.perf_counter() - t0) / done
    # the sampled layers stand for all of them (identical shapes); embedding / lm_head cost is
    # already included once per call and is not scaled
    ratio = geo["layers"] / layers
    per_request = t_prefill * ratio + decode_len * (t_decode * ratio) / num_requests
    return {
        "tokens_per_s": decode_len / per_request,
        "sample": (f"{layers} of {geo['layers']} layers timed and scaled x{ratio:g}; {num_requests} requests: "
                   f"{prompt_len}-token prefill each ({t_prefill:.2f} s/request/sample) + {done} batched decode "
                   f"steps ({t_decode:.3f} s/step/sample); torch CPU fp32 math on bf16-rounded weights"),
    }
