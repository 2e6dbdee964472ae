"""Data-parallel gradient synchronisation: one process per GPU, RCCL over xGMI.

The reference's only multi-GPU mode is single-process nn.DataParallel
(train.py:509-510): scatter the batch, replicate the weights, reduce-add the
gradients to GPU 0 every step.  Here each rank owns a model replica and a shard
of the minibatch; the ONE exchange per step is a sum-all-reduce of the flat
fp32 gradient buffer (torch.distributed backend "nccl" = RCCL), issued bucket by
bucket on a side stream while the LAST backward level (pass 0 of the level
loop, the only one after which a gradient slot is final) is still running, and
averaged inside the fused AdamW (grad_scale = 1/world).  BatchNorm statistics
stay per-rank, exactly as the reference's DataParallel + SyncBatchNorm without
a process group behaves.

Bucket boundaries are the tape marks the model emits ("shared_head",
"transition3", ...): when the reverse pass crosses mark m, every parameter
registered at or after module m is final, so [offset(m), previous boundary) goes
out.  xGMI is point-to-point; with 263 MB (HRNet) in 5-6 buckets each
all-reduce stays large enough (>= 10 MB) to be link-bandwidth- not
latency-bound.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def bucket_offsets(flat, marks):
    """mark name -> flat offset of the first parameter whose name starts with it"""
    out = {}
    for m in marks:
        cands = [off for name, (off, _) in flat.slots.items() if name == m or name.startswith(m + ".")]
        if cands:
            out[m] = min(cands)
    return out


class RcclComm:
    """The library's own RCCL communicator (`hrseg_comm_*`, include/hrseg.h): one per process, the 128-byte
    id travels from rank 0 through a TCP store at MASTER_ADDR:(MASTER_PORT+1).  `GradSync(backend="rccl")`
    uses it instead of torch.distributed for the gradient all-reduce."""

    def __init__(self, rank, world, device, addr=None, port=None):
        import ctypes
        import os
        from . import _lib
        self._lib, self.rank, self.world, self.device = _lib, rank, world, torch.device(device)
        torch.cuda.set_device(self.device)
        idbuf = ctypes.create_string_buffer(128)
        if world > 1:
            addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
            port = port or int(os.environ.get("MASTER_PORT", "29500")) + 1
            store = dist.TCPStore(addr, port, world, is_master=(rank == 0))
            if rank == 0:
                _lib.call_raw("hrseg_comm_unique_id", idbuf)
                store.set("hrseg_rccl_id", idbuf.raw)
            else:
                idbuf.raw = store.get("hrseg_rccl_id")
            self._store = store
        else:
            _lib.call_raw("hrseg_comm_unique_id", idbuf)
        self._comm = ctypes.c_void_p()
        _lib.call_raw("hrseg_comm_init", ctypes.byref(self._comm), rank, world, idbuf)

    def all_reduce_(self, t, stream):
        """in-place sum of a contiguous fp32 tensor over the ranks, stream-ordered on `stream`"""
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        self._lib.call_raw("hrseg_comm_allreduce_async", self._comm, t.data_ptr(), t.numel(), stream.cuda_stream)

    def wait(self, comm_stream, consumer):
        self._lib.call_raw("hrseg_comm_wait", comm_stream.cuda_stream, consumer.cuda_stream)

    def close(self):
        if self._comm:
            self._lib.call_raw("hrseg_comm_destroy", self._comm)
            self._comm = None


class GradSync:
    """Attach to a model: ``sync = GradSync(model, group); ...; loss.backward()`` leaves the
    summed gradient in model._flat.grad (the optimizer divides by world size)."""

    def __init__(self, model, group=None, marks=("layer1", "transition1", "transition2", "transition3",
                                                 "shared_head", "down2", "down4", "up2", "up4"), min_bucket=1 << 20,
                 backend="torch", comm=None):
        """backend "torch": torch.distributed (nccl = RCCL on ROCm, gloo on CPU); "rccl": the library's own
        communicator `comm` (an RcclComm) through the C ABI."""
        self.model = model
        self.group = group
        self.backend, self.comm = backend, comm
        assert backend in ("torch", "rccl") and (backend == "torch" or comm is not None)
        self.marks = marks
        import os as _os
        if "HRSEG_SYNC_MIN_BUCKET" in _os.environ:        # elements; a huge value = one all-reduce at the end
            min_bucket = int(_os.environ["HRSEG_SYNC_MIN_BUCKET"])
        self.min_bucket = min_bucket
        self.world = comm.world if backend == "rccl" else (dist.get_world_size(group) if dist.is_initialized() else 1)
        # HRSEG_FORCE_SYNC=1: issue the collectives even on a single rank (rehearses the RCCL + stream
        # choreography on a one-GPU box)
        import os
        self.force = os.environ.get("HRSEG_FORCE_SYNC", "0") == "1" and (dist.is_initialized() or backend == "rccl")
        self._comm_stream = None
        self._handles = []
        self._boundary = None
        self._offsets = None
        self.launched = []          # [(lo, hi)] of the last step, for tests/inspection
        # timing = True: every bucket of a step is bracketed by events on the exchange stream and the end of the reverse pass
        # is marked on the compute stream; `bucket_report()` then says, per bucket, when it started, how long it took and how
        # much of it ran AFTER the reverse pass had finished (= not overlapped).  Off in production (events cost host time).
        self.timing = False
        self._events = []
        model._grad_hook = self
        if self.world > 1:
            self.broadcast_state()

    def broadcast_state(self, src=0):
        """Replicas must start from identical weights and buffers (what nn.DataParallel's per-step
        replicate() guarantees in the reference): rank `src`'s flat parameter buffer and every module
        buffer (BN running statistics) are broadcast once."""
        m = self.model
        flat = m.flatten_parameters() if hasattr(m, "flatten_parameters") else getattr(m, "_flat", None)
        bufs = list(m.buffers()) if hasattr(m, "buffers") else []
        if self.backend == "rccl":
            # broadcast = all-reduce with every other rank contributing zeros (the ABI has one collective)
            cur = torch.cuda.current_stream()
            packed = torch.cat([b.detach().reshape(-1).float() for b in bufs]) if bufs else None
            for t in ([flat.data] if flat is not None else []) + ([packed] if packed is not None else []):
                if self.comm.rank != src:
                    t.zero_()
                self.comm.all_reduce_(t, cur)
            o = 0
            with torch.no_grad():
                for b in bufs:
                    b.copy_(packed[o:o + b.numel()].view(b.shape).to(b.dtype))
                    o += b.numel()
            return
        if flat is not None and getattr(flat, "data", None) is not None:
            dist.broadcast(flat.data, src=src, group=self.group)
        for b in bufs:
            dist.broadcast(b, src=src, group=self.group)

    # the model calls this with a mark name while the last backward level runs, then with "end"
    def __call__(self, mark):
        flat = self.model._flat
        if self._offsets is None or self._offsets[0] is not flat:
            self._offsets = (flat, bucket_offsets(flat, self.marks))
        if self._boundary is None:
            self._boundary = flat.numel
            self.launched = []
            self._events = []
        if mark == "end":
            if self.timing and flat.grad.is_cuda:
                self._bwd_end = torch.cuda.Event(enable_timing=True)
                self._bwd_end.record()                      # compute stream: the reverse pass (and the weight-gradient join) is done
            self._launch(flat, 0, self._boundary)
            self._finish()
            self._boundary = None
            return
        off = self._offsets[1].get(mark)
        if off is None or self._boundary - off < self.min_bucket:
            return
        self._launch(flat, off, self._boundary)
        self._boundary = off

    def _launch(self, flat, lo, hi):
        if hi <= lo:
            return
        self.launched.append((lo, hi))
        if self.world == 1 and not self.force:
            return
        chunk = flat.grad[lo:hi]
        if chunk.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=chunk.device)
            self._comm_stream.wait_stream(torch.cuda.current_stream(chunk.device))
            from .engine import wgrad_stream
            side = wgrad_stream(chunk.device)
            if side is not None:                 # the bucket's weight gradients come from the side stream
                self._comm_stream.wait_stream(side)
            ev = None
            if self.timing:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), lo, hi)
                ev[0].record(self._comm_stream)
                self._events.append(ev)
            if self.backend == "rccl":
                self.comm.all_reduce_(chunk, self._comm_stream)
                if ev is not None:
                    ev[1].record(self._comm_stream)
                return
            with torch.cuda.stream(self._comm_stream):
                h = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if ev is not None:
                    h.wait()                                # exchange stream waits for the collective's own stream
                    ev[1].record(self._comm_stream)
                self._handles.append(h)
        else:
            self._handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _finish(self):
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._comm_stream is not None:
            if self.backend == "rccl":
                self.comm.wait(self._comm_stream, torch.cuda.current_stream())
            else:
                torch.cuda.current_stream().wait_stream(self._comm_stream)


def _bucket_report(self):
    """after a synchronised step with timing on: [{bytes, start_ms (after the first bucket's start), ms, exposed_ms}] --
    exposed = the part of the bucket's interval that lies behind the end of the reverse pass"""
    if not self._events:
        return []
    torch.cuda.synchronize()
    t0 = self._events[0][0]
    end_bwd = t0.elapsed_time(self._bwd_end) if getattr(self, "_bwd_end", None) is not None else None
    out = []
    for a, b, lo, hi in self._events:
        start, stop = t0.elapsed_time(a), t0.elapsed_time(b)
        exposed = None if end_bwd is None else max(0.0, stop - max(start, end_bwd))
        out.append({"bytes": 4 * (hi - lo), "start_ms": round(start, 3), "ms": round(stop - start, 3),
                    "exposed_ms": None if exposed is None else round(exposed, 3)})
    if end_bwd is not None:
        out.append({"reverse_pass_end_ms": round(end_bwd, 3)})
    return out


GradSync.bucket_report = _bucket_report


def all_reduce_confusion(cms, group=None):
    """Global per-step metrics under data parallelism: sum the per-level confusion counts (int64,
    (C+child)^2 entries each) over the ranks.  The reference computes its metrics on the batch gathered
    on GPU 0 (nn.DataParallel); summed counts give exactly those values.  One small collective."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or not cms:
        return cms
    flat = torch.cat([c.reshape(-1) for c in cms])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    out, o = [], 0
    for c in cms:
        out.append(flat[o:o + c.numel()].view_as(c))
        o += c.numel()
    return out


def init_distributed():
    """Process group from torchrun's environment; returns (rank, local_rank, world)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs for a one-GPU box: HRSEG_DIST_BACKEND=gloo with HRSEG_FORCE_DEVICE=0 runs
    # several ranks on one card (RCCL refuses two ranks per device); production is nccl = RCCL
    backend = os.environ.get("HRSEG_DIST_BACKEND", "nccl")
    if "HRSEG_FORCE_DEVICE" in os.environ:
        local = int(os.environ["HRSEG_FORCE_DEVICE"])
    force = os.environ.get("HRSEG_FORCE_SYNC", "0") == "1"
    if force and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if (world > 1 or force) and not dist.is_initialized():
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
