"""Data parallelism over RCCL/xGMI: one process per GPU, gradients only.

Replaces the reference's ``torch.nn.parallel.DistributedDataParallel`` wiring (src/models/model_utils.py:41-67,
src/train.py:67,212) and its per-iteration loss all-reduce (src/utils/train_utils.py:107-111).  Differences by design:

  * parameter gradients already live in ONE flat fp32 buffer (Darknet._grad_table); the backward plan finishes
    modules last-to-first, so the tail of that buffer becomes final first.  Whenever >= bucket_bytes of tail are
    final, that contiguous range is all-reduced on a side HIP stream while the backward kernels keep running on
    the compute stream (event-ordered, no host synchronisation);
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU): a 256 MB gradient is ~3 ms of ring all-reduce against a
    ~50 ms step, so a few large buckets (default 64 MB) beat DDP's 25 MB default -- fewer, larger collectives;
  * no per-forward broadcast of BatchNorm buffers (DDP's broadcast_buffers default): BN statistics are per-GPU in
    the reference too (no SyncBN) and rank 0's are the ones checkpointed (train_utils.py:82-85);
  * the 1/world of the mean is folded into the backward kernels' gradient reductions (no extra pass over 256 MB);
  * the loss scalar is reduced only when asked (``reduce_tensor``), not as a side effect.

Gradient accumulation over sub-divisions (reference train.py:212-221: ``backward()`` every batch, ``optimizer.step()``
every ``subdivisions`` batches).  The flat buffer is accumulated into across backwards, and a SUM all-reduce is only
idempotent on contributions that have not been reduced yet, so the buffer's form is tracked:

    EMPTY    zero_grad() happened (every ``.grad`` is None at the next backward)
    LOCAL    holds sum_i g_i(this rank) / world -- not reduced yet (after a backward under ``no_sync()``)
    REDUCED  holds sum_i mean_over_ranks(g_i)  -- identical on every rank (after a synchronised backward)

A backward that finds the buffer REDUCED first rescales it by 1/world (one pass over 256 MB, ~0.1 ms): the SUM over
ranks of R/world is R again, so the all-reduce of ``R/world + g_local/world`` yields ``R + mean(g)``.  ``no_sync()``
(same contract as DistributedDataParallel.no_sync) skips the collectives of the enclosed backwards; with it a
sub-divided step costs ONE all-reduce and no rescale -- ``accumulate(model, i, subdivisions)`` picks the right context
for micro-batch i.
"""
import contextlib

import torch
import torch.distributed as dist

EMPTY, LOCAL, REDUCED = 'empty', 'local', 'reduced'


class RcclDataParallel(torch.nn.Module):
    def __init__(self, module, bucket_bytes=64 << 20, process_group=None):
        super().__init__()
        self.module = module
        self.group = process_group
        self.bucket_bytes = int(bucket_bytes)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # CY_DDP_FORCE=1 runs the collectives even for a single rank (plumbing check on a 1-GPU box)
        import os
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get('CY_DDP_FORCE') == '1')
        self._side = None
        self._offsets = None
        self._tail = None
        self._pending = []
        self._sync = True
        self._form = EMPTY
        # bench.py: a list here collects one (before, after) event pair per backward around the compute stream's wait for
        # the all-reduce stream -- their distance is the part of the collective that backward did not hide
        self.exposed_events = None
        module._pre_backward_hooks.append(self._pre_backward)
        module._post_backward_hooks.append(self._finish)
        module._module_grad_hooks.append(self._module_done)
        if self.active:
            self._broadcast_state()
            module.grad_prescale = 1.0 / self.world   # the backward kernels emit gradient/world: SUM all-reduce = mean

    def _broadcast_state(self):
        """Rank 0's parameters and buffers everywhere (what the DDP constructor does once)."""
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            dist.broadcast(t.data, src=0, group=self.group)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    @contextlib.contextmanager
    def no_sync(self):
        """Backwards inside accumulate locally (no collective); the next backward outside reduces the lot."""
        prev, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = prev

    def _pre_backward(self, model, accumulating):
        """Runs before the backward plan adds this micro-batch's gradients (see the module docstring)."""
        if not self.active:
            return
        if not accumulating:
            self._form = EMPTY
        elif self._form == REDUCED:
            model.flat_grad.mul_(1.0 / self.world)
            self._form = LOCAL

    # ---- bucketed all-reduce -----------------------------------------------------------------------
    def _prepare(self):
        named = list(self.module.named_parameters())
        offs, off = {}, 0
        for name, p in named:
            idx = int(name.split('.')[1])
            offs.setdefault(idx, off)
            off += p.numel()
        self._offsets, self._total = offs, off
        self._tail = off

    def _reduce_range(self, lo, hi):
        flat = self.module.flat_grad
        if flat is None or hi <= lo or not self.active or not self._sync:
            return
        chunk = flat[lo:hi]
        if flat.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)

    def _module_done(self, model, idx):
        """Called by the engine after the backward of module ``idx``: every gradient at or after its offset is final."""
        if not self.active:
            return
        if self._offsets is None:
            self._prepare()
        lo = self._offsets.get(idx)
        if lo is None:
            return
        if (self._tail - lo) * 4 >= self.bucket_bytes:
            self._reduce_range(lo, self._tail)
            self._tail = lo

    def _finish(self, model):
        if not self.active:
            return
        if self._offsets is None:
            self._prepare()
        self._reduce_range(0, self._tail)
        self._tail = self._total
        self._form = REDUCED if self._sync else LOCAL
        flat = model.flat_grad
        if flat is not None and flat.is_cuda and self._side is not None:
            cur = torch.cuda.current_stream(flat.device)
            if self.exposed_events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self._side)
                e1.record(cur)
                self.exposed_events.append((e0, e1))
            else:
                cur.wait_stream(self._side)


def reduce_tensor(tensor, world_size):
    """reference train_utils.py:107-111 (mean over ranks of a logging scalar)."""
    rt = tensor.detach().clone()
    if dist.is_initialized() and world_size > 1:
        dist.all_reduce(rt, op=dist.ReduceOp.SUM)
        rt /= world_size
    return rt


def accumulate(model, micro_step, subdivisions):
    """Context for micro-batch ``micro_step`` (0-based) of a ``subdivisions``-fold accumulated step: ``no_sync()`` for all
    but the last one when ``model`` is an RcclDataParallel, a null context otherwise."""
    if isinstance(model, RcclDataParallel) and (micro_step + 1) % max(1, subdivisions) != 0:
        return model.no_sync()
    return contextlib.nullcontext()


def subdivisions_for(batch_size, ngpus_per_node):
    """reference train.py:69 computes int(64 / batch_size / ngpus) which is 0 for large batches and then divides by
    it (ZeroDivisionError at train.py:213); clamp to >= 1."""
    return max(1, int(64 / batch_size / max(1, ngpus_per_node)))
