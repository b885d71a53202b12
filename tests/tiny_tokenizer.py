"""A small tokenizer built in memory for the front-end tests (there is no hub access and the
repository carries no tokenizer files): byte-level BPE trained on a few sentences, wrapped as a
``PreTrainedTokenizerFast`` with bos / eos / pad tokens and a chat template."""
from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
from transformers import PreTrainedTokenizerFast

_CORPUS = [
    "the quick brown fox jumps over the lazy dog", "hello world, this is a test of the serving front end",
    "prompt and token machines hand the key value cache over", "naïve café déjà vu — ünïcödé 漢字 かな 🙂",
    "user: what is the answer? assistant: the answer is forty two.", "STOP here. And go on!",
    "scheduler block manager sequence group sampling parameters temperature nucleus logprobs",
    "paged attention reshape and cache copy blocks swap blocks prefill decode pipeline",
    "tensor parallel pools collective all reduce gather broadcast rendezvous handshake",
    "quantised weights zeros scales groups shuffle reconstruct matrix cores wavefront",
    "throughput latency histogram counter gauge prometheus metrics endpoint health",
    "0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 32 64 128 256 512 1024 2048 4096",
] * 4

CHAT_TEMPLATE = ("{% for m in messages %}<|{{ m['role'] }}|>{{ m['content'] }}</s>{% endfor %}"
                 "{% if add_generation_prompt %}<|assistant|>{% endif %}")


def build_tokenizer(vocab_size: int = 640) -> PreTrainedTokenizerFast:
    tok = Tokenizer(models.BPE(unk_token=None))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=["<s>", "</s>", "<pad>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(_CORPUS, trainer)
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", pad_token="<pad>")
    fast.chat_template = CHAT_TEMPLATE
    assert len(fast) >= 512, len(fast)            # the stand-in executors of the tests emit ids below 512
    return fast
