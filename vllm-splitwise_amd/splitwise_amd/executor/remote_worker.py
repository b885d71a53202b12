"""The command loop of a worker that lives in another process (the token worker of a split pair)."""
import gc
import os
import traceback
from typing import Any, Dict

import torch

from ..config import DeviceConfig
from ..sequence import SamplerOutput, pack_sampler_output
from ..worker.worker import Worker


def make_exchange(channel):
    """Symmetric swap with the paired worker over the control channel."""
    def exchange(obj, peer_index: int = 0):
        # (a prompt worker with several token pools swaps with them in turn; the other end of the channel
        # knows the order, so the index does not travel)
        channel.send(("kv_exchange", obj))
        tag, peer = channel.recv()
        assert tag == "kv_exchange", tag
        return peer
    return exchange


def worker_loop(channel, cfg: Dict[str, Any], worker_cls=Worker) -> None:
    """cfg: model_config, parallel_config, scheduler_config, cache_config, rank, local_rank, transport.
    ``worker_cls`` is the worker implementation (tests drive the control plane with a CPU stand-in)."""
    worker = None
    try:
        transport_init = None
        if cfg["parallel_config"].sep_prompt_token:
            transport_init = {"exchange": make_exchange(channel), "transport": cfg["transport"]}
        extra = {}
        if cfg.get("distributed_init_method"):
            extra["distributed_init_method"] = cfg["distributed_init_method"]
        worker = worker_cls(cfg["model_config"], cfg["parallel_config"], cfg["scheduler_config"], DeviceConfig("cuda"),
                            local_rank=cfg["local_rank"], rank=cfg["rank"],
                            kv_cache_dtype=cfg["cache_config"].cache_dtype, is_driver_worker=True,
                            kv_transport_init=transport_init, **extra)
        reports = cfg.get("reports_output", True)      # tensor-parallel followers compute but do not report
        pending = None                                  # samples of a deferred step that have not been reported yet
        frozen = os.environ.get("SPLITWISE_FREEZE_GC", "1") == "0"
        while True:
            msg = channel.recv()
            cmd, args = msg[0], msg[1:]
            if not frozen and cmd in ("execute", "execute_deferred"):
                # set-up is over (model, cache, captured graphs): keep the cyclic collector away from those
                # long-lived objects, as the engine does on its side
                gc.collect()
                gc.freeze()
                frozen = True
            if cmd == "shutdown":
                channel.send(("ok", None))
                break
            try:
                if cmd == "call":
                    name, kwargs = args
                    result = getattr(worker, name)(**kwargs)
                elif cmd == "execute":
                    (kwargs,) = args
                    result = worker.execute_model(**kwargs)
                    if not reports:
                        result = None
                    elif isinstance(result, SamplerOutput):
                        channel.send(("samples", pack_sampler_output(result)))
                        continue
                elif cmd in ("execute_deferred", "flush"):
                    # launch the new step first, then report the PREVIOUS one: the device always has
                    # the next step queued while the engine works on the last one's tokens
                    new = worker.execute_model(**args[0], defer=True) if cmd == "execute_deferred" else None
                    prev, pending = pending, (new if reports else None)
                    if prev is None:
                        channel.send(("ok", None))
                    else:
                        channel.send(("samples", pack_sampler_output(prev.resolve())))
                    continue
                elif cmd == "bytes_pushed":
                    comm = worker.kvcache_comm_manager.kvcache_comm if worker.kvcache_comm_manager else None
                    result = comm.bytes_pushed if comm else 0
                else:
                    raise ValueError(f"unknown command {cmd!r}")
                channel.send(("ok", result))
            except Exception as exc:            # report and keep serving: the engine decides
                channel.send(("error", f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"))
    finally:
        try:
            from ..model_executor.custom_all_reduce import destroy_custom_ar
            destroy_custom_ar()
        except Exception:
            pass
        if worker is not None and worker.kvcache_comm_manager is not None:
            try:
                worker.destroy_kvcache_comm()
            except Exception:
                pass


def spawn_entry(conn, cfg: Dict[str, Any]) -> None:
    from .channel import PipeChannel
    import os
    if os.environ.get("SPLITWISE_DEBUG_HANG"):          # dump this process's stacks if it is still alive then
        import faulthandler
        import sys
        faulthandler.dump_traceback_later(float(os.environ["SPLITWISE_DEBUG_HANG"]), repeat=False, file=sys.stderr)
    worker_cls = cfg.get("worker_cls") or Worker      # tests drive the control plane with CPU stand-ins
    if worker_cls is Worker:
        torch.cuda.set_device(cfg["local_rank"])
    worker_loop(PipeChannel(conn), cfg, worker_cls=worker_cls)
