"""Tokenizer loading and incremental detokenisation (vllm/transformers_utils/tokenizer.py:14-245).

There is no hub access: a tokenizer is a local directory (``tokenizer.json`` / sentencepiece model
plus config) or an object the caller built.  Detokenisation is incremental: each step turns only a
short window of tokens into text instead of the whole output, and holds text back while the window
ends inside a multi-byte character.
"""
import os
from typing import List, Optional, Tuple

_WINDOW = 5          # tokens of context kept before the unread ones: enough for byte-fallback and space rules


def get_tokenizer(tokenizer_name: str, *args, tokenizer_mode: str = "auto", trust_remote_code: bool = False, **kwargs):
    """Loads a tokenizer from a local directory (vllm/transformers_utils/tokenizer.py:14-79)."""
    if not os.path.isdir(tokenizer_name):
        raise ValueError(f"tokenizer {tokenizer_name!r} is not a local directory; this build never downloads "
                         "(pass a directory, or start the engine without a tokenizer and send token ids)")
    if tokenizer_mode == "slow":
        if kwargs.get("use_fast", False):
            raise ValueError("Cannot use the fast tokenizer in slow tokenizer mode.")
        kwargs["use_fast"] = False
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(tokenizer_name, *args, trust_remote_code=trust_remote_code,
                                         local_files_only=True, **kwargs)


def _tokens_to_text(tokenizer, tokens: List[str], skip_special_tokens: bool, spaces_between_special_tokens: bool) -> str:
    """``convert_tokens_to_string`` that keeps added tokens whole: fast tokenizers do that themselves; slow
    ones would run their byte / sentencepiece decoding over the added token's text
    (vllm/transformers_utils/tokenizer.py:130-165)."""
    if tokenizer.is_fast or not tokenizer.get_added_vocab():
        return tokenizer.convert_tokens_to_string(tokens)
    added = tokenizer.get_added_vocab()
    special = set(tokenizer.all_special_tokens) if skip_special_tokens else set()
    pieces: List[str] = []
    run: List[str] = []
    for tok in tokens:
        if tok in special:
            continue
        if tok in added:
            if run:
                pieces.append(tokenizer.convert_tokens_to_string(run))
                run = []
            pieces.append(tok)
        else:
            run.append(tok)
    if run:
        pieces.append(tokenizer.convert_tokens_to_string(run))
    return (" " if spaces_between_special_tokens else "").join(pieces)


def detokenize_incrementally(tokenizer, all_input_ids: List[int], prev_tokens: Optional[List[str]],
                             prefix_offset: int = 0, read_offset: int = 0, skip_special_tokens: bool = False,
                             spaces_between_special_tokens: bool = True
                             ) -> Tuple[List[str], str, int, int]:
    """Text contributed by the last id of ``all_input_ids``
    (vllm/transformers_utils/tokenizer.py:176-245).

    Returns (new token strings, new text, prefix_offset, read_offset).  ``tokens[prefix_offset:read_offset]``
    is context that has been emitted already, ``tokens[read_offset:]`` has not: the new text is what the
    longer window decodes to beyond the context's own decoding.  While the window ends in U+FFFD (an
    unfinished byte sequence) nothing is emitted and the offsets stay."""
    new_id = all_input_ids[-1]
    if prev_tokens is None:
        # first call: every token so far (the prompt too) becomes context
        new_tokens = tokenizer.convert_ids_to_tokens(all_input_ids, skip_special_tokens=skip_special_tokens)
        tokens = new_tokens
        prefix_offset = max(len(tokens) - _WINDOW - 1, 0)
        # a special last token that is skipped left no string behind: nothing unread
        skipped = skip_special_tokens and new_id in tokenizer.all_special_ids
        read_offset = len(tokens) if skipped else max(len(tokens) - 1, 0)
    else:
        new_tokens = tokenizer.convert_ids_to_tokens([new_id], skip_special_tokens=skip_special_tokens)
        tokens = prev_tokens + new_tokens
    context = _tokens_to_text(tokenizer, tokens[prefix_offset:read_offset], skip_special_tokens,
                              spaces_between_special_tokens)
    window = _tokens_to_text(tokenizer, tokens[prefix_offset:], skip_special_tokens, spaces_between_special_tokens)
    if len(window) > len(context) and not window.endswith("�"):
        return new_tokens, window[len(context):], read_offset, len(tokens)
    return new_tokens, "", prefix_offset, read_offset
