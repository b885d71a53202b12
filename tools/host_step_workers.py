"""Stand-in workers whose steps cost nothing: what remains of an engine step is the host's control plane."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-splitwise_amd"), os.path.join(ROOT, "tests")]

from split_fakes import FakeWorker  # noqa: E402
from splitwise_amd.sequence import Logprob, SamplerOutput, SequenceGroupOutput, SequenceOutput  # noqa: E402


class NullWorker(FakeWorker):
    """FakeWorker's life cycle and hand-off flags, no cache contents: every sequence gets token 7.  An optional
    sleep stands for the device's step (HOST_STEP_DEVICE_MS_PROMPT / HOST_STEP_DEVICE_MS_DECODE)."""

    def _execute(self, seq_group_metadata_list, blocks_to_nw, nw_epochs, nw_peer=0):
        blocks_to_nw = blocks_to_nw or {}
        prompt = bool(seq_group_metadata_list) and seq_group_metadata_list[0].is_prompt
        ms = float(os.environ.get("HOST_STEP_DEVICE_MS_PROMPT" if prompt else "HOST_STEP_DEVICE_MS_DECODE", "0"))
        if ms > 0:
            time.sleep(ms * 1e-3)
        outputs = []
        for meta in seq_group_metadata_list:
            (seq_id, _), = meta.seq_data.items()
            outputs.append(SequenceGroupOutput([SequenceOutput(seq_id, 7, {7: Logprob(0.0)})]))
        if self.is_prompt:
            flags = self.peer_flag_arrays[nw_peer]
            for slot in blocks_to_nw:
                flags[slot] = nw_epochs[slot]
        return SamplerOutput(outputs)
