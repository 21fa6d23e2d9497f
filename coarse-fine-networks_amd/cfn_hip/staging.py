"""Input staging: host batches -> HBM on a copy stream, double buffered, one slab per batch (">= 6x at 8 GPUs hinges on input
staging and launch overhead").

The reference moves every batch with `Variable(inputs.cuda())` on the default stream (train_fine.py:184-197, train_coarse_fineFEAT.py:205-224):
a SYNCHRONOUS pageable copy of 0.3-1.2 GB in front of every step.  Here

    for inputs, labels, masks, names in staging.stage(loader, device):      # same structure, tensors now live on `device`
        train_step(...)

a background thread takes batch i + 1 from the loader while step i runs, lays ALL its tensors out in ONE device slab (256-byte aligned views:
clip, labels, masks, the 5 feature maps, meta ...) and enqueues the host -> device transfer on a dedicated copy stream:

  * tensors the loader already pinned (DataLoader(pin_memory=True), collate into pinned memory) go straight from where they are, one async copy
    each -- no host-side copy at all;
  * pageable tensors are first gathered into the slot's PINNED host slab (one memcpy, in the background thread, the GIL released) and leave as
    ONE async copy of the slab.

Ordering is by events only (no host synchronisation in the consumer's thread): the compute stream waits for the slot's `ready` event before the
first kernel reads the batch; the copy stream waits for the slot's `free` event -- recorded on the compute stream when the consumer asks for
the NEXT batch, i.e. behind every kernel it enqueued for this one -- before it overwrites the slab.  A batch's tensors are therefore valid for
all work enqueued before the next batch is requested; clone what has to live longer.  `depth` slots (default 2) = `depth - 1` batches in flight
beside the one being consumed.  Non-tensor members (names, durations) and tensors already on the device pass through untouched.
"""
import queue
import threading
import time

import torch

ALIGN = 256


def _map_tensors(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    if isinstance(obj, tuple) and hasattr(obj, '_fields'):
        return type(obj)(*[_map_tensors(v, fn) for v in obj])
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(v, fn) for v in obj)
    return obj


class _Slot(object):
    def __init__(self):
        self.host = self.dev = None
        self.ready = torch.cuda.Event()
        self.free = torch.cuda.Event()
        self.used = False
        self.retired = []         # outgrown slabs: kernels enqueued earlier may still read them


class _End(object):
    pass


class _Raise(object):
    def __init__(self, exc):
        self.exc = exc


class HostStager(object):
    """stage(iterable) -> iterator of device-resident batches (see the module docstring).  bytes_staged / batches: running totals."""

    def __init__(self, device, depth=2, pin_pageable=True):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('HostStager moves batches into HBM: a cuda device is required (there is no CPU path)')
        self.depth = max(int(depth), 2)
        self.pin_pageable = pin_pageable
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = [_Slot() for _ in range(self.depth)]
        self.bytes_staged = 0
        self.batches = 0
        self._thread = self._stop = None

    # -- layout ----------------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _plan(batch):
        """[(tensor, offset, nbytes)] of the CPU tensors of `batch` in traversal order, total bytes"""
        plan, off = [], 0

        def visit(t):
            nonlocal off
            if t.device.type == 'cpu' and t.numel() > 0:
                nb = t.numel() * t.element_size()
                plan.append((t, off, nb))
                off += (nb + ALIGN - 1) // ALIGN * ALIGN
            return t
        _map_tensors(batch, visit)
        return plan, off

    def _ensure(self, slot, total, need_host):
        if slot.dev is None or slot.dev.numel() < total:
            if slot.dev is not None:
                slot.retired.append(slot.dev)
            cap = max(total, 2 * (slot.dev.numel() if slot.dev is not None else 0))
            slot.dev = torch.empty(cap, dtype=torch.uint8, device=self.device)
        if need_host and (slot.host is None or slot.host.numel() < total):
            cap = max(total, 2 * (slot.host.numel() if slot.host is not None else 0))
            slot.host = torch.empty(cap, dtype=torch.uint8, pin_memory=True)

    # -- producer side (background thread) -----------------------------------------------------------------------------------------------------
    def _put(self, slot, batch):
        plan, total = self._plan(batch)
        pageable = [p for p in plan if not p[0].is_pinned()]
        use_slab = self.pin_pageable and bool(pageable)
        if slot.used:
            # the pinned host slab is about to be rewritten: its previous transfer must have left the host (polled, not synchronised: the
            # consumer thread's "no host synchronisation" guarantee is checked with torch.cuda.set_sync_debug_mode)
            while not slot.ready.query():
                time.sleep(0.0002)
        self._ensure(slot, total, use_slab)
        views = {}
        with torch.cuda.device(self.device), torch.cuda.stream(self.copy_stream):
            if slot.used:
                self.copy_stream.wait_event(slot.free)          # every kernel that read the previous batch of this slot has finished
            if use_slab:
                lo, hi = min(p[1] for p in pageable), max(p[1] + p[2] for p in pageable)
                for t, off, nb in pageable:
                    slot.host[off:off + nb].view(t.dtype).view(t.shape).copy_(t)                       # host memcpy (GIL released)
                slot.dev[lo:hi].copy_(slot.host[lo:hi], non_blocking=True)                               # ONE transfer for all of them
            for t, off, nb in plan:
                dv = slot.dev[off:off + nb].view(t.dtype).view(t.shape)
                views[id(t)] = dv
                if t.is_pinned() or not use_slab:
                    dv.copy_(t, non_blocking=True)                # pinned source: asynchronous, straight from where the loader left it
            slot.ready.record(self.copy_stream)
        slot.used = True
        self.bytes_staged += total
        self.batches += 1
        keep = [t for t, _, _ in plan if t.is_pinned()]            # pinned sources must outlive their asynchronous copies
        return _map_tensors(batch, lambda t: views.get(id(t), t)), keep

    def _producer(self, it, free_q, ready_q, stop):
        try:
            for batch in it:
                slot = None
                while slot is None:
                    if stop.is_set():
                        return
                    try:
                        slot = free_q.get(timeout=0.05)
                    except queue.Empty:
                        pass
                staged, keep = self._put(slot, batch)
                ready_q.put((slot, staged, keep))
            ready_q.put(_End())
        except BaseException as exc:                              # surfaces in the consumer's thread
            ready_q.put(_Raise(exc))

    # -- consumer side -----------------------------------------------------------------------------------------------------------------------
    def stage(self, iterable):
        if self._thread is not None and self._thread.is_alive():      # an earlier pass over a loader was abandoned mid-way: its producer
            self._stop.set()                                          # must be gone before the slots are handed out again
            self._thread.join()
        free_q, ready_q, stop = queue.Queue(), queue.Queue(), threading.Event()
        for s in self.slots:
            free_q.put(s)
        th = threading.Thread(target=self._producer, args=(iter(iterable), free_q, ready_q, stop), name='cfn-staging', daemon=True)
        self._thread, self._stop = th, stop
        th.start()
        held = None
        try:
            while True:
                item = ready_q.get()
                if held is not None:
                    # the consumer came back for the next batch: everything it enqueued for the previous one is on its stream by now
                    held.free.record(torch.cuda.current_stream(self.device))
                    free_q.put(held)
                    held = None
                if isinstance(item, _End):
                    return
                if isinstance(item, _Raise):
                    raise item.exc
                slot, staged, keep = item
                torch.cuda.current_stream(self.device).wait_event(slot.ready)
                held = slot
                yield staged
                del keep
        finally:
            stop.set()
            if held is not None:
                held.free.record(torch.cuda.current_stream(self.device))


def stage(iterable, device, depth=2):
    """one-shot spelling: `for batch in stage(loader, device)` (a fresh HostStager per call; keep a HostStager to reuse its slabs).  A CPU
    `device` (the gloo tests of the distributed host logic) iterates the loader as it is."""
    if torch.device(device).type != 'cuda':
        return iter(iterable)
    return HostStager(device, depth=depth).stage(iterable)
