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
    dev = torch.device('cuda', local) if use_cuda else torch.device('cpu')
    if use_cuda:
        torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group(backend or ('nccl' if use_cuda else 'gloo'), rank=rank, world_size=world)
    return rank, world, dev


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
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

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


def global_mask_count(masks, group=None):
    """sum(masks) over the GLOBAL batch: the loc-loss normaliser of train_fine.py:212 is taken over the
    batch DataParallel gathered on GPU 0, so a sharded run has to all-reduce it."""
    tot = masks.sum().detach().clone()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(tot, group=group)
    return tot


def broadcast_buffers(module, src=0, group=None):
    """BN running statistics stay per replica during training (as under DataParallel); before
    evaluation / checkpointing rank 0's buffers are the ones that survive in the reference."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for b in module.buffers():
        dist.broadcast(b, src, group=group)
