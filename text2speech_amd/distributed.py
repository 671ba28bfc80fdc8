"""Data-parallel WaveGlow training over RCCL / xGMI: the replacement for reference
``waveglow/distributed.py`` (same three entry points: ``init_distributed``, ``apply_gradient_allreduce``,
``reduce_tensor``; same "modifies the module, returns the same object" contract, distributed.py:90-94).

What differs, by design:

* the reference flattens ALL 268 M gradients into one buffer and all-reduces it from an autograd
  *final* callback, i.e. strictly after backward (distributed.py:105-133): 1.07 GB on the wire with
  nothing to overlap.  Here the hand-scheduled backward (glow_autograd.backward_train) walks the flows
  12 -> 1 and hands each finished flow's gradients (one flat ~89 MB bucket) to RCCL *while the next
  flow's backward kernels run*: 12 bucketed all-reduces on RCCL's own stream, only the last one exposed.
  xGMI is point-to-point (7 links x ~153 GB/s), so a ring all-reduce is per-link bound; ~89 MB buckets
  keep every link busy without serialising the whole gradient behind the last kernel.
* one process per GPU (torchrun / torch.distributed.run), backend "nccl" == RCCL on ROCm.
* the initial weight sync is one broadcast per flat bucket, not one per tensor (938 messages -> 13).

For modules other than this package's WaveGlow a generic bucketed post-backward all-reduce is used.
"""
import torch
import torch.distributed as dist


def init_distributed(rank, num_gpus, group_name=None, dist_backend="nccl", dist_url="tcp://127.0.0.1:54321"):
    """Reference distributed.py:43-53.  ``group_name`` is accepted for signature parity (unused by torch >= 1.x)."""
    if dist_backend == "nccl":
        assert torch.cuda.is_available(), "Distributed mode requires a GPU."
        torch.cuda.set_device(rank % torch.cuda.device_count())
    if not dist.is_initialized():
        dist.init_process_group(dist_backend, init_method=dist_url, world_size=num_gpus, rank=rank)


def reduce_tensor(tensor, num_gpus):
    """Reference distributed.py:37-41: mean over ranks of a (loss) tensor."""
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= num_gpus
    return rt


def _avg_op(backend=None):
    """AVG where the backend of the group that runs the collective has it (RCCL); SUM (then scaled by 1/world) elsewhere -
    gloo rejects ReduceOp.AVG.  ``backend``: that group's backend (default: the default group's)."""
    if backend is None:
        backend = dist.get_backend()
    return dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM


class GradSync:
    """Bucketed, overlapped gradient averaging used from inside the hand-written backward.

    Each bucket is shipped from a dedicated communication stream: that stream waits for the event that marks the bucket
    complete (recorded by the caller on whichever stream wrote the last gradient), hands the flat buffer to the collective and
    records an event behind it, so the compute streams are never blocked by a collective until ``finish()``, and every bucket has
    a (ready, done) event pair of its own - the per-bucket timing an 8-GPU run needs to be diagnosable:
    ``stats()`` returns, per bucket of the last step, bytes, ready -> complete ms and the gap to the previous completion.
    The overlap holds for RCCL ("nccl") groups only: there ``h.wait()`` is a stream-level dependency.  On a gloo group with
    device tensors (the CPU rehearsal path) ``h.wait()`` blocks the host until the bucket is averaged - correct, not overlapped.
    """

    def __init__(self, group=None, timing=True):
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.pending = []
        self.n_buckets = 0
        self.bytes = 0
        self.timing = timing
        self.comm_stream = None
        self._last = []            # (bytes, ev_ready, ev_done) of the step that finished last

    def reduce_async(self, flat, ready_events=()):
        """Start averaging one flat gradient bucket; returns immediately.  ``ready_events``: events recorded by the caller, one
        per stream that wrote gradients into the bucket, each after its last write (default: now, on the current stream)."""
        op = _avg_op(self.backend)
        cuda = flat.is_cuda
        if cuda and self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=flat.device)
        ev_ready = ev_done = None
        if cuda:
            if not ready_events:
                e = torch.cuda.Event()
                e.record()
                ready_events = (e,)
            for e in ready_events:
                self.comm_stream.wait_event(e)
            with torch.cuda.stream(self.comm_stream):
                ev_ready = torch.cuda.Event(enable_timing=self.timing)      # fires when the last of the bucket's writers is done
                ev_ready.record()
                h = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
                h.wait()                                    # stream-level for RCCL: the comm stream now trails the collective
                if op == dist.ReduceOp.SUM:
                    flat.mul_(1.0 / self.world)
                ev_done = torch.cuda.Event(enable_timing=self.timing)
                ev_done.record()
            flat.record_stream(self.comm_stream)
        else:
            h = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        self.pending.append((h, flat, op, ev_ready, ev_done))
        self.n_buckets += 1
        self.bytes += flat.numel() * flat.element_size()

    def finish(self):
        """Every shipped bucket has been averaged before the caller's stream goes on (and before ``optimizer.step()``)."""
        last = []
        for h, flat, op, ev_ready, ev_done in self.pending:
            if ev_done is not None:
                torch.cuda.current_stream(flat.device).wait_event(ev_done)
            else:
                h.wait()
                if op == dist.ReduceOp.SUM:
                    flat.mul_(1.0 / self.world)
            last.append((flat.numel() * flat.element_size(), ev_ready, ev_done))
        self._last = last
        self.pending = []

    def stats(self):
        """Per-bucket figures of the last finished step (call after a device synchronize): bytes, ms from 'bucket ready' to
        'all-reduce complete', ms between consecutive completions (back-to-back buckets: bytes / that = the achieved bus rate)."""
        out, prev_done = [], None
        for nbytes, ev_ready, ev_done in self._last:
            row = {"bytes": nbytes}
            if ev_ready is not None and ev_done is not None and self.timing:
                try:
                    row["ready_to_complete_ms"] = ev_ready.elapsed_time(ev_done)
                    if prev_done is not None:
                        row["since_previous_complete_ms"] = prev_done.elapsed_time(ev_done)
                except RuntimeError:            # an event that was never recorded / not yet complete
                    pass
            prev_done = ev_done
            out.append(row)
        return {"world": self.world, "backend": self.backend, "buckets": out}


def _broadcast_flat(tensors, src=0, bucket_elems=32 * 1024 * 1024):
    """One broadcast per flat bucket instead of one per tensor (reference distributed.py:100-103)."""
    bucket, n = [], 0

    def flush():
        nonlocal bucket, n
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.broadcast(flat, src)
        off = 0
        for t in bucket:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        bucket, n = [], 0

    by_type = {}
    for t in tensors:
        by_type.setdefault((t.dtype, t.device), []).append(t)
    for group in by_type.values():
        for t in group:
            bucket.append(t)
            n += t.numel()
            if n >= bucket_elems:
                flush()
        flush()


def apply_gradient_allreduce(module):
    """Make ``module`` average its gradients over all ranks on every backward and start from rank 0's
    weights.  Same object is returned (reference distributed.py:90-142)."""
    with torch.no_grad():
        _broadcast_flat([p for p in module.state_dict().values() if torch.is_tensor(p)])
    from .glow import WaveGlow
    if isinstance(module, WaveGlow):
        module._eng().grad_sync = GradSync()
        return module

    # generic path: bucket by size, all-reduce once every gradient of the step exists
    sync = GradSync()

    def allreduce_params():
        if not module.needs_reduction:
            return
        module.needs_reduction = False
        bucket, n = [], 0

        def flush():
            nonlocal bucket, n
            if not bucket:
                return
            flat = torch.cat([g.reshape(-1) for g in bucket])
            sync.reduce_async(flat)
            sync.finish()
            off = 0
            for g in bucket:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
            bucket, n = [], 0

        for p in module.parameters():
            if p.requires_grad and p.grad is not None:
                bucket.append(p.grad.data)
                n += p.grad.numel()
                if n >= 32 * 1024 * 1024:
                    flush()
        flush()

    def hook(*unused):
        torch.autograd.Variable._execution_engine.queue_callback(allreduce_params)

    for p in module.parameters():
        if p.requires_grad:
            p.register_hook(hook)

    def set_needs_reduction(self, inputs, output):
        self.needs_reduction = True

    module.needs_reduction = False
    module.register_forward_hook(set_needs_reduction)
    return module
