"""Clip-level data parallelism: one process per GPU, gradients summed with RCCL all-reduce over xGMI.

The reference uses single-process ``nn.DataParallel`` (train_fine.py:123): replicate / scatter /
gather / reduce-add through GPU 0 every iteration.  Here every rank owns a full replica and a shard
of the mini-batch; the only exchange step is the gradient all-reduce (13.2 MB fp32 for x3d_fine,
18.1 MB for x3d_coarse) plus one scalar for the loss normaliser.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of a 13 MB message is latency,
not bandwidth, bound (2*(N-1)/N * 13 MB / 153 GB/s ~ 0.15 ms), so gradients are packed into a few
multi-MB buckets (fewer, larger collectives) that are launched on a side stream as soon as the last
gradient of a bucket has been accumulated, overlapping with the rest of backward.

Works on any torch.distributed backend (``nccl`` = RCCL on ROCm; ``gloo`` for the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*); returns (rank, world, device)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    use_cuda = torch.cuda.is_available()
    # CFN_DIST_BACKEND / CFN_SHARE_GPU: test hooks -- several ranks on ONE GPU over gloo (RCCL needs one device per rank), so
    # that the whole multi-rank path (parameter sync, bucketed all-reduce behind backward, global loss normaliser) can be
    # exercised with the real kernels on a one-GPU box
    backend = backend or os.environ.get('CFN_DIST_BACKEND') or None
    if use_cuda and os.environ.get('CFN_SHARE_GPU'):
        local = local % torch.cuda.device_count()
    dev = torch.device('cuda', local) if use_cuda else torch.device('cpu')
    if use_cuda:
        torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group(backend or ('nccl' if use_cuda else 'gloo'), rank=rank, world_size=world)
    return rank, world, dev


class _BucketHook(object):
    """post-accumulate hook of GradReducer: counts a bucket's gradients in and launches its all-reduce when the last one
    exists.  It flushes the lazily cast weight gradients before it reads them (ops.flush_grad_casts)."""

    def __init__(self, reducer):
        self.reducer = reducer

    def __call__(self, p):
        self.reducer._on_grad(p)


class GradReducer(object):
    """Bucketed, backward-overlapped gradient averaging for a replica's parameters.

    Collectives are matched across ranks by issue order, so buckets are launched strictly in bucket order.
    Every bucket owns ONE static flat buffer (gradients + one "somebody had a gradient" flag per parameter), allocated at the
    first use and reused every step: no allocation inside backward.  Every step issues exactly one all-reduce per bucket, in
    bucket order, on every rank -- a bucket whose parameters got no gradient on this rank (data-dependent branches: the rw6
    dropout path, multi-crop) is sent zero-filled by finish(), so the ranks can never disagree on the number of collectives.
    A parameter without a local gradient receives the averaged one iff some rank had one (its flag came back non-zero)."""

    def __init__(self, params, bucket_bytes=4 << 20, group=None, force=False):
        """force=True registers the bucket hooks even for a single rank (the collective is then an identity): used by the
        GPU test of the backward-overlapped path, which cannot have two ranks on a one-GPU box"""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = bool(force) and dist.is_initialized()
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []            # list of lists of params, in reverse registration order; one dtype / device per bucket
        cur, size = [], 0
        for p in reversed(self.params):   # backward produces gradients roughly in reverse order
            if cur and (p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._bucket_of[p] = bi
        nb = len(self.buckets)
        self._pending = [len(b) for b in self.buckets]
        self._next = 0               # buckets [0, _next) have been launched this step
        self._flat = [None] * nb     # static flat buffer per bucket: [gradients ..., one flag per parameter]
        self._views = [None] * nb
        self._inflight = []
        self._rerun = False          # more than one backward pass since the last finish(): every bucket is sent again with the accumulated gradients
        self._passes = 0             # backward passes completed since the last finish() (counted by an engine callback: the SAME number on every rank,
        self._in_pass = False        # whatever gradients a rank's data produced -- ADVICE r4: the decision to re-send must not depend on hook arrival)
        self._begun = 0              # begin_pass() calls since the last finish(): the rank-independent pass count -- the engine-callback count
                                     # above misses a pass that produced no gradient for any of this reducer's parameters on THIS rank
        self.filled = []
        self._stream = None
        self._hooks = []
        self.suspended = False       # True while a backward is being CAPTURED (cfn_hip.graph.GraphedDPStep): no collective may enter the graph
        # the lazily cast weight gradients (ops._GradCast) are handed out only for parameters whose every reader flushes
        # first: this reducer does (see _launch), so it opts its own parameters in.  Parameters nobody opted in -- a model
        # wrapped in torch DDP / FSDP, whose reducer hooks sit on the AccumulateGrad node where they cannot be seen -- get an
        # immediate cast.
        from . import ops
        ops.allow_lazy_grad_cast(self.params)
        if self.world > 1 or self.force:
            hook = _BucketHook(self)
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(hook))

    def _comm_stream(self, dev):
        if dev.type != 'cuda':
            return None
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        return self._stream

    def _buffers(self, bi):
        if self._flat[bi] is None:
            b = self.buckets[bi]
            flat = torch.zeros(sum(p.numel() for p in b) + len(b), dtype=b[0].dtype, device=b[0].device)
            views, off = [], 0
            for p in b:
                views.append(flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
            self._flat[bi], self._views[bi] = flat, views
        return self._flat[bi], self._views[bi]

    def no_sync(self):
        """Context manager for gradient accumulation (the reference's `num_steps_per_update` knob, train_fine.py:65): backward
        passes inside it only accumulate into p.grad, nothing is sent; the LAST micro-batch's backward runs outside it and is
        followed by finish(), which reduces the accumulated gradients.  Every rank must use it the same way."""
        reducer = self

        class _NoSync(object):
            def __enter__(self):
                self.prev, reducer.suspended = reducer.suspended, True

            def __exit__(self, *exc):
                reducer.suspended = self.prev
                reducer._in_pass = False      # (a backward that raised drops the engine's callbacks: never leave the flag set)
                return False
        return _NoSync()

    def begin_pass(self):
        """Call right before every backward() this reducer takes part in (the train steps of this package do).  It makes the number of
        backward passes since the last finish() -- which decides whether the buckets are sent again with accumulated gradients -- a
        count of CALLS, identical on every rank by construction; without it the count comes from autograd-engine callbacks queued by the
        first gradient hook of a pass, which a rank whose pass produced no gradient for any reducer parameter never queues."""
        if not self.suspended:       # (passes under no_sync() / inside a graph capture only accumulate: they are not counted, as in _end_pass)
            self._begun += 1

    def _completed_passes(self):
        return self._begun - 1 if self._begun > 0 else self._passes

    def close(self):
        """Detach from the parameters: removes the bucket hooks and withdraws the lazy-gradient-cast opt-in (it is only safe
        while THIS reducer, which flushes before it reads, is the one reading the gradients: a torch DDP wrapped around the
        same model afterwards reads them from hooks nobody can see)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        from . import ops
        ops.allow_lazy_grad_cast(self.params, False)

    def _on_grad(self, p):
        if self.suspended:
            return
        if not self._in_pass:        # first gradient of this backward pass: count the pass when the engine finishes it
            self._in_pass = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_pass)
        if self._completed_passes() > 0:
            # gradient accumulation without no_sync(): a second (third ...) backward before finish().  The first pass's collectives are in flight with
            # ITS gradients; p.grad holds the local sum of all passes (results are only written back in finish()), so nothing is launched from inside
            # this pass (a bucket would travel with partial sums) and finish() sends every bucket again, in bucket order on every rank -- decided by
            # the pass COUNT, which every rank shares, not by which hooks happened to fire here.
            self._rerun = True
            return
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if bi < self._next or self._pending[bi] < 0:     # (a parameter that accumulates twice inside one pass: treated like a second pass)
            self._rerun = True
            return
        # collectives are matched across ranks by issue order: buckets are launched strictly in bucket order, a complete
        # bucket behind an incomplete one waits (for that one, or for finish())
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)

    def _end_pass(self):
        self._in_pass = False
        if not self.suspended:
            self._passes += 1

    def _launch(self, bi):
        b = self.buckets[bi]
        flat, views = self._buffers(bi)
        from . import ops
        ops.flush_grad_casts()   # weight gradients are cast fp64 -> fp32 lazily; make this bucket's valid
        have = [i for i, p in enumerate(b) if p.grad is not None]
        if len(have) == len(b):
            flat[-len(b):].fill_(1.0)
        else:                    # gradient-less parameters travel as zeros with a zero flag
            flat.zero_()
            if have:
                flat[-len(b):].copy_(torch.tensor([1.0 if b[i].grad is not None else 0.0 for i in range(len(b))], dtype=flat.dtype),
                                     non_blocking=True)
        if have:
            torch._foreach_copy_([views[i] for i in have], [b[i].grad for i in have])
        flat.div_(self.world)
        st = self._comm_stream(flat.device)
        if st is not None:
            st.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(st):
                work = dist.all_reduce(flat, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(flat, group=self.group, async_op=True)
        assert bi == self._next
        self._next = bi + 1
        self._inflight.append((work, bi))

    def finish(self):
        """Call after backward(): launches the buckets backward did not complete (so that every rank issues one collective
        per bucket, always), waits for the collectives and writes the averaged gradients back."""
        if self.world == 1 and not self.force:
            return
        if self.suspended:           # captured backward: the reduction runs eagerly after the replay (finish() is called again there)
            return
        while self._next < len(self.buckets):
            self._launch(self._next)
        if (self._begun if self._begun > 0 else self._passes) > 1:
            self._rerun = True
        if self._rerun:              # several backward passes since the last finish(): reduce the accumulated gradients
            for work, bi in self._inflight:
                work.wait()
            self._inflight, self._next, self._rerun = [], 0, False
            while self._next < len(self.buckets):
                self._launch(self._next)
        self.filled = []             # parameters that had no local gradient and received another rank's (GraphedDPStep checks)
        for work, bi in self._inflight:
            work.wait()
            b, flat, views = self.buckets[bi], self._flat[bi], self._views[bi]
            have = [i for i, p in enumerate(b) if p.grad is not None]
            if have:    # one multi-tensor launch per bucket, not one per parameter
                torch._foreach_copy_([b[i].grad for i in have], [views[i] for i in have])
            if len(have) != len(b):
                flags = flat[-len(b):].tolist()          # rare path (a host sync): who, anywhere, had a gradient
                for i, p in enumerate(b):
                    if p.grad is None and flags[i] > 0:
                        p.grad = views[i].clone()
                        self.filled.append(p)
        self._inflight = []
        self._pending = [len(b) for b in self.buckets]
        self._next = 0
        self._passes = 0
        self._begun = 0
        self._in_pass = False        # (engine callbacks are dropped when a backward raises: the flag must not survive the step)


def global_mask_count(masks, group=None, local=False):
    """sum(masks) over the GLOBAL batch: the loc-loss normaliser of train_fine.py:212 is taken over the
    batch DataParallel gathered on GPU 0, so a sharded run has to all-reduce it.  local=True (evaluation: ranks hold
    different numbers of videos, nothing is averaged across ranks) skips the collective."""
    tot = masks.sum().detach().clone()
    if not local and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(tot, group=group)
    return tot


def _broadcast_flat(tensors, src, group):
    """one collective per dtype instead of one per tensor"""
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for ts in by_dtype.values():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src, group=group)
        off = 0
        with torch.no_grad():
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n


def sync_module(module, src=0, group=None):
    """Make every replica identical to rank `src`: parameters AND buffers.  nn.DataParallel replicates ONE model each
    iteration (train_fine.py:123); with one process per GPU the replicas are built independently, and anything not in the
    pretrained checkpoint (`replace_logits`' fresh fc2, rw2-6, mix2-5, pool_1) would otherwise start different on every
    rank -- gradient averaging alone never brings them together.  Call once after the model is complete."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    _broadcast_flat(list(module.parameters()) + list(module.buffers()), src, group)


def all_agree(flag, device, group=None):
    """True only if `flag` is true on EVERY rank (a collective: every rank must call it the same number of times).
    Used for the reference's "skip a short last batch" rule (train_fine.py:180-181), so that no rank skips alone and
    leaves the others waiting in a gradient all-reduce."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def gather_objects(obj, dst=0, group=None):
    """list of every rank's `obj` on rank `dst` (None elsewhere); validation scores / targets / CSV rows"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [obj]
    out = [None] * dist.get_world_size(group) if dist.get_rank(group) == dst else None
    dist.gather_object(obj, out, dst=dst, group=group)
    return out


def mean_over_ranks(values, device, group=None):
    """element-wise mean of a short list of python floats over the ranks (logging only)"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    return (t / dist.get_world_size(group)).tolist()


def describe(group=None):
    """one line for logs / bench output: world size as torch.distributed sees it, backend, RCCL version"""
    if not dist.is_initialized():
        return {'world_size': 1, 'backend': None}
    out = {'world_size': dist.get_world_size(group), 'backend': dist.get_backend(group)}
    try:
        out['rccl'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        pass
    return out


def broadcast_buffers(module, src=0, group=None):
    """BN running statistics stay per replica during training (as under DataParallel); before
    evaluation / checkpointing rank 0's buffers are the ones that survive in the reference."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    _broadcast_flat(list(module.buffers()), src, group)
