import sys, os, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vllm-splitwise_amd"))
from splitwise_amd import LLM, SamplingParams
from splitwise_amd.sequence import SequenceData, SequenceGroupMetadata
from splitwise_amd.model_executor.layers.basic import RMSNorm
from oracle import model as omodel
llm = LLM("llama-tiny", enforce_eager=True, num_gpu_blocks_override=64, max_num_seqs=4, max_model_len=256, swap_space=0.01)
w = llm.llm_engine.model_executor.driver_worker
m = w.model_runner.model
sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
cfg = w.model_runner.model_config.hf_config
random.seed(0)
toks = [random.randrange(512) for _ in range(5)]
meta = [SequenceGroupMetadata("0", True, {0: SequenceData(toks)}, SamplingParams(temperature=0, max_tokens=1), {0: [0]})]
t, p, md, pl, st = w.model_runner._prepare_prompt_packed(meta)
with torch.inference_mode():
    hid = m(t.view(-1), p.view(-1), w.gpu_cache, md)
    logits = (hid[:5].float() @ m.lm_head.weight.float().t()).cpu()
ref = omodel.llama_logits(sd, cfg, toks)
print("max diff logits", (logits - ref).abs().max().item(), "ref absmax", ref.abs().max().item())
# op-level checks
x = torch.randn(7, 256, dtype=torch.float16, device="cuda")
n = RMSNorm(256).cuda().half()
print("rms diff", (n(x).float() - n._forward(x).float()).abs().max().item())
r = torch.randn(7, 256, dtype=torch.float16, device="cuda")
a, b = n._forward(x.clone(), r.clone())
x2, r2 = x.clone(), r.clone()
n(x2, r2)
print("fused rms diff", (x2.float() - a.float()).abs().max().item(), (r2.float() - b.float()).abs().max().item())
rot = m.layers[0].self_attn.rotary_emb
qkv = torch.randn(7, 512, dtype=torch.float16, device="cuda")
q, k, v = qkv.split([256, 128, 128], dim=-1)
pos = torch.arange(7, device="cuda")
qr, kr = rot._forward(pos, q, k)
rot(pos, q, k)
print("rope diff", (q.float() - qr.float()).abs().max().item(), (k.float() - kr.float()).abs().max().item())
act = m.layers[0].mlp.act_fn
y = torch.randn(7, 1024, dtype=torch.float16, device="cuda")
print("silu diff", (act(y).float() - act._forward(y).float()).abs().max().item())
# ---- first-layer breakdown
import torch.nn.functional as F
caps = {}
def hook(name):
    def f(mod, inp, out):
        caps[name] = (out[0] if isinstance(out, tuple) else out).detach().float().cpu().clone()
    return f
L0 = m.layers[0]
hs = [L0.input_layernorm.register_forward_hook(hook("ln1")), L0.self_attn.qkv_proj.register_forward_hook(hook("qkv")),
      L0.self_attn.attn.register_forward_hook(hook("attn")), L0.self_attn.register_forward_hook(hook("attn_block")),
      m.embed_tokens.register_forward_hook(hook("emb"))]
with torch.inference_mode():
    m(t.view(-1), p.view(-1), w.gpu_cache, md)
sdf = {k: v.float() for k, v in sd.items()}
x = sdf["embed_tokens.weight"][torch.tensor(toks)]
print("emb", (caps["emb"][:5] - x).abs().max().item())
y = omodel._rms(x, sdf["layers.0.input_layernorm.weight"], 1e-5)
print("ln1", (caps["ln1"][:5] - y).abs().max().item())
qkv = y @ sdf["layers.0.self_attn.qkv_proj.weight"].t()
print("qkv (captured after in-place rope) vs pre-rope", (caps["qkv"][:5] - qkv).abs().max().item())
q, k, v = qkv.split([256, 128, 128], -1)
pos = torch.arange(5)
q = omodel._rope_neox(q, 4, pos, 10000.0); k = omodel._rope_neox(k, 2, pos, 10000.0)
print("q rope", (caps["qkv"][:5, :256] - q).abs().max().item(), "k rope", (caps["qkv"][:5, 256:384] - k).abs().max().item())
a = omodel._causal_attention(q, k, v, 4, 2)
print("attn", (caps["attn"][:5] - a).abs().max().item())
