"""The request / output data model, after the reference's tests/test_sequence.py:1-50 (a sampler output behaves like
the list of its groups), tests/test_sampling_params.py:6-8 and tests/test_cache_block_hashing.py:45-93 (content
addresses of cache blocks: equal exactly as long as everything up to the block's end is equal).  The reference
hashes tokenised sentences; here the token ids are drawn directly -- what is hashed is the ids either way."""
import random

import pytest

import conftest  # noqa: F401
from splitwise_amd import SamplingParams
from splitwise_amd.sequence import SamplerOutput, Sequence, SequenceGroupOutput, SequenceOutput


@pytest.fixture
def sample_outputs():
    return [SequenceGroupOutput(samples=[SequenceOutput(parent_seq_id=0, output_token=i, logprobs={})],
                                prompt_logprobs=None) for i in range(5)]


def test_sampler_output_is_a_list_of_its_groups(sample_outputs):
    out = SamplerOutput(outputs=sample_outputs)
    assert len(out) == len(sample_outputs) and out[2] == sample_outputs[2]
    assert out.sampled_token_probs is None and out.sampled_token_ids is None
    assert out.spec_decode_worker_metrics is None
    new = SequenceGroupOutput(samples=[SequenceOutput(parent_seq_id=0, output_token=99, logprobs={})],
                              prompt_logprobs=None)
    out[2] = new
    assert out[2] == new and list(out)[2] == new


def test_sampler_outputs_compare_by_content(sample_outputs):
    a, b = SamplerOutput(outputs=sample_outputs), SamplerOutput(outputs=sample_outputs.copy())
    c = SamplerOutput(outputs=sample_outputs[:-1])
    assert a == b and a != c and a != sample_outputs


def test_max_tokens_none_is_allowed():
    assert SamplingParams(temperature=0.01, top_p=0.1, max_tokens=None).max_tokens is None


@pytest.mark.parametrize("block_size", [8, 16])
def test_block_hashes_follow_the_prefix(block_size):
    rng = random.Random(0)
    common = [rng.randrange(32000) for _ in range(150)]
    starts = [[rng.randrange(32000) for _ in range(5)], [rng.randrange(32000) for _ in range(3)]]
    tails = [[rng.randrange(32000) for _ in range(n)] for n in (block_size, block_size + 7, 2 * block_size, 40)]
    hashes = []
    for start in starts:
        hashes.append([])
        for i, tail in enumerate(tails):
            ids = start + common + tail
            seq = Sequence(i, None, ids, block_size)
            hashes[-1].append([seq.hash_of_block(b) for b in range(len(ids) // block_size)])
    # prefixes with different first blocks: different everywhere
    for h0, h1 in zip(sum(hashes[0], []), sum(hashes[1], [])):
        assert h0 != h1
    # one prefix, different endings: equal up to the last block wholly inside the prefix, different once a block
    # holds tokens of the ending
    for start, per_prefix in zip(starts, hashes):
        shared = (len(start) + len(common)) // block_size
        assert len({tuple(h[:shared]) for h in per_prefix}) == 1
        for b in range(shared, min(len(h) for h in per_prefix)):
            assert len({h[b] for h in per_prefix}) == len(per_prefix)
    # the address of a block does not depend on what follows it, nor on the sequence object
    seq = Sequence(9, None, starts[0] + common, block_size)
    assert [seq.hash_of_block(b) for b in range(4)] == hashes[0][0][:4]
    assert seq.num_hashed_tokens_of_block(2) == 3 * block_size
