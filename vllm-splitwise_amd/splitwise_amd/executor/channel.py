"""Control-plane channels between the engine process and a remote worker process: small pickled
messages only (step commands, sampled tokens, IPC handles).  The reference uses Ray RPC for this
(vllm/executor/ray_gpu_executor.py:343-452); the KV bytes never travel here."""
from typing import Any


def pickled(obj: Any) -> bytes:
    """The bytes ``PipeChannel.send`` would put on the wire for ``obj``: a command that goes to several ranks is
    serialised once (``send_pickled``)."""
    from multiprocessing.reduction import ForkingPickler
    return bytes(ForkingPickler.dumps(obj))


class PipeChannel:
    """A multiprocessing connection (full duplex)."""

    def __init__(self, conn) -> None:
        self.conn = conn

    def send(self, obj: Any) -> None:
        self.conn.send(obj)

    def send_pickled(self, payload: bytes) -> None:
        """``send`` of an object serialised beforehand with ``pickled``; the receiver sees no difference."""
        self.conn.send_bytes(payload)

    def recv(self, timeout: float = None) -> Any:
        """Blocks at most ``timeout`` seconds (None: for ever).  A peer that has gone away ends the wait at
        once with EOFError (its end of the pipe closes with it); TimeoutError when nothing came in time."""
        if timeout is not None and not self.conn.poll(timeout):
            raise TimeoutError(f"no message on the control channel within {timeout:g} s")
        return self.conn.recv()

    def poll(self, timeout: float = 0.0) -> bool:
        return self.conn.poll(timeout)

    def close(self) -> None:
        self.conn.close()


class DistChannel:
    """torch.distributed point-to-point objects over a gloo group (ranks launched by torchrun)."""

    def __init__(self, peer_rank: int, group=None) -> None:
        self.peer = peer_rank
        self.group = group

    def send(self, obj: Any) -> None:
        import torch.distributed as dist
        dist.send_object_list([obj], dst=self.peer, group=self.group)

    def recv(self, timeout: float = None) -> Any:
        """``timeout`` is the process group's own (set when it was created): a gloo receive cannot be
        given another one per call."""
        import torch.distributed as dist
        box = [None]
        dist.recv_object_list(box, src=self.peer, group=self.group)
        return box[0]

    def poll(self, timeout: float = 0.0) -> bool:
        return True                    # no way to look without receiving: the receive itself then waits

    def close(self) -> None:
        pass
