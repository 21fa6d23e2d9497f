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
    """post-accumulate hook of GradReducer.  The marker tells cfn_hip.ops that this hook flushes the lazily cast weight
    gradients before it reads them (ops._lazy_ok); any other post-accumulate hook disables the lazy cast for its
    parameter."""
    _cfn_flushes_grad_casts = True

    def __init__(self, reducer):
        self.reducer = reducer

    def __call__(self, p):
        self.reducer._on_grad(p)


class GradReducer(object):
    """Bucketed, backward-overlapped gradient averaging for a replica's parameters."""

    def __init__(self, params, bucket_bytes=4 << 20, group=None, force=False):
        """force=True registers the bucket hooks even for a single rank (the collective is then an identity): used by the
        GPU test of the backward-overlapped path, which cannot have two ranks on a one-GPU box"""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = bool(force) and dist.is_initialized()
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []            # list of lists of params, in reverse registration order
        cur, size = [], 0
        for p in reversed(self.params):   # backward produces gradients roughly in reverse order
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
        self._pending = [len(b) for b in self.buckets]
        self._inflight = []
        self._stream = None
        self._hooks = []
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

    def _on_grad(self, p):
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        ps = [p for p in self.buckets[bi] if p.grad is not None]
        if not ps:
            return
        from . import ops
        ops.flush_grad_casts()   # weight gradients are cast fp64 -> fp32 lazily; make this bucket's valid
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        flat.div_(self.world)
        st = self._comm_stream(flat.device)
        if st is not None:
            st.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(st):
                work = dist.all_reduce(flat, group=self.group, async_op=True)
            flat.record_stream(st)
        else:
            work = dist.all_reduce(flat, group=self.group, async_op=True)
        self._inflight.append((work, flat, ps))

    def finish(self):
        """Call after backward(): waits for the collectives and writes the averaged gradients back."""
        if self.world == 1 and not self.force:
            return
        for bi, left in enumerate(self._pending):   # buckets whose params got no gradient this step
            if left != 0 and left != len(self.buckets[bi]):
                self._launch(bi)
        for work, flat, ps in self._inflight:
            work.wait()
            views, off = [], 0
            for p in ps:
                n = p.numel()
                views.append(flat[off:off + n].view_as(p.grad))
                off += n
            torch._foreach_copy_([p.grad for p in ps], views)     # one multi-tensor launch per bucket, not one per parameter
        self._inflight = []
        self._pending = [len(b) for b in self.buckets]


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
