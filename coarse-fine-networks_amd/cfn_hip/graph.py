"""hipGraph capture of a whole train step (forward + loss + backward + optimizer).

A Coarse-Fine train step is ~1.5k (fine) to ~2.5k (coarse) kernel launches, each reached through autograd + ctypes
(~8-12 us of host work apiece): below ~4 clips per GPU the step is launch bound (the coarse step spends 28 of 31 ms
on the host at 8 clips).  The C ABI is capture safe by construction (no allocation, no synchronisation, no host<->device
copy inside an entry point; every launch goes to the caller's stream), so the whole step is recorded ONCE per input shape
into a hipGraph and replayed with a single launch per step.

    step = GraphedStep(lambda x, labels, masks: train_step(net, reducer, opt, x, labels, masks), optimizer=opt)
    cls, loc, probs = step(x, labels, masks)        # call 1: eager (a real step); call 2: capture + replay; then replays

Rules the captured callable has to respect (checked where possible):
  * tensors in, tensors out; inputs are copied into static buffers before each replay, outputs are static buffers that
    the next replay overwrites -- clone what has to survive;
  * no .item() / float(tensor) / host<->device copies inside; no data-dependent Python control flow;
  * gradients live in the graph's memory pool: never `zero_grad(set_to_none=True)` between replays from outside (the
    captured step may do it internally: backward then re-creates them at the same addresses at capture time);
  * the learning rate is baked into the optimizer kernels: a changed `lr` (scheduler, warm-up) triggers a re-capture;
  * a graph is REPLAYED ON THE STREAM IT WAS CAPTURED ON (`_replay`): the library's per-(device, stream) scratch buffers (pre-split weight
    images, weight-gradient partial tiles: csrc/pwstream.hip, csrc/pwsplitw.hip) are baked into the graph by address and are only ordered
    against other users by in-stream order -- replayed on another stream, eager work on the capture stream (or a second graph captured on it)
    could overwrite them mid-contraction.  Every GraphedStep owns its side stream, so two graphed steps never share scratch;
  * multi-GPU: collectives are NOT captured, and GradReducer's hooks would launch RCCL on a side stream inside the capture:
    with torch.distributed initialised at world_size > 1 a GraphedStep refuses to capture (RuntimeError).  Use GraphedDPStep:
    graph A = forward + loss + backward (reducer suspended), then the bucketed all-reduce EAGERLY on the reducer's static flat
    buffers, then graph B = optimizer step.
"""
import torch

from . import ops


def _capture_mode():
    """Stream-capture error mode.  With a process group alive, ProcessGroupNCCL's watchdog THREAD polls the events of earlier collectives
    (hipEventQuery); under the default 'global' mode that call is illegal while any thread captures and the watchdog aborts the process
    ("operation not permitted when stream is capturing": tests/test_hip_train.py failed that way in about half of the runs).  'thread_local'
    restricts the check to the capturing thread, which is the one that matters here."""
    import torch.distributed as dist
    return 'thread_local' if (dist.is_available() and dist.is_initialized()) else 'global'


def _lr_signature(optimizer):
    return tuple((g.get('lr'), g.get('momentum'), g.get('weight_decay')) for g in optimizer.param_groups) if optimizer else ()


class GraphedStep(object):
    """eager_first: the first calls run eagerly as ordinary steps -- they create the lazily allocated state a capture must
    not contain (SGD momentum buffers are CLONED from the gradient on the first step: captured, every replay would reset
    them) -- so no hidden extra optimisation steps are ever taken: every call, eager or replayed, is exactly one step."""

    def __init__(self, fn, optimizer=None, eager_first=1, pool=None):
        self.fn, self.optimizer, self.eager_first, self.pool = fn, optimizer, eager_first, pool
        self.calls = 0
        self.stream = None      # side stream shared by the eager first steps and every capture (see _on_side_stream)
        self._graphs = {}       # shape signature -> (graph, static inputs, static outputs, lr signature)

    @staticmethod
    def _sig(args):
        return tuple((tuple(a.shape), a.dtype, a.device) if torch.is_tensor(a) else
                     tuple(sorted((k, tuple(v.shape)) for k, v in a.items())) if isinstance(a, dict) else a for a in args)

    @staticmethod
    def _clone(a):
        if torch.is_tensor(a):
            return a.clone()
        if isinstance(a, dict):
            return {k: v.clone() for k, v in a.items()}
        return a

    @staticmethod
    def _copy(dst, src):
        if torch.is_tensor(dst):
            dst.copy_(src, non_blocking=True)
        elif isinstance(dst, dict):
            for k in dst:
                dst[k].copy_(src[k], non_blocking=True)

    def _side_stream(self, args):
        """Autograd remembers the stream an AccumulateGrad node was created on and synchronises with it in every later
        backward; a node born on the default stream would drag the (uncapturable) default stream into the capture.  So the
        eager first steps run on the same side stream the captures use."""
        if self.stream is None:
            dev = next(a.device for a in args if torch.is_tensor(a)) if any(torch.is_tensor(a) for a in args) else None
            self.stream = torch.cuda.Stream(device=dev)
        return self.stream

    def _eager(self, args):
        st = self._side_stream(args)
        st.wait_stream(torch.cuda.current_stream(st.device))
        with torch.cuda.stream(st):
            out = self.fn(*args)
        torch.cuda.current_stream(st.device).wait_stream(st)
        return out

    def _capture(self, args):
        static_in = [self._clone(a) for a in args]
        torch.cuda.synchronize()
        ops.reset_scratch()                    # the zero-filled scratch must be allocated (and zeroed) INSIDE the graph
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, pool=self.pool, stream=self._side_stream(args), capture_error_mode=_capture_mode()):
            static_out = self.fn(*static_in)
        ops.reset_scratch()                    # ... and must not leak into later eager calls
        return graph, static_in, static_out, _lr_signature(self.optimizer)

    def _replay(self, graph):
        """launch `graph` on the capture stream, ordered behind the caller's stream and in front of its later work"""
        st = self.stream
        cur = torch.cuda.current_stream(st.device)
        if cur == st:
            graph.replay()
            return
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            graph.replay()
        cur.wait_stream(st)

    def __call__(self, *args):
        if self.calls < self.eager_first:
            self.calls += 1
            return self._eager(args)
        sig = self._sig(args)
        entry = self._graphs.get(sig)
        if entry is not None and entry[3] != _lr_signature(self.optimizer):
            entry = None                       # learning rate changed: the optimizer kernels hold the old value
        if entry is None:
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_world_size() > 1:
                raise RuntimeError('GraphedStep: collectives are not captured; with world_size > 1 run the step eagerly')
            entry = self._capture(args)
            self._graphs[sig] = entry
            # the capture itself executed nothing: fall through to a replay so that this call performs one real step
        graph, static_in, static_out, _ = entry
        for dst, src in zip(static_in, args):
            self._copy(dst, src)
        self._replay(graph)
        return static_out


class GraphedDPStep(GraphedStep):
    """Data-parallel step as two hipGraphs around an eager gradient all-reduce (world_size >= 1):

        step = GraphedDPStep(lambda x, l, m, tot: train_fine.forward_backward(net, x, l, m, mask_total=tot), reducer, optimizer,
                             pre=lambda x, l, m: (cdist.global_mask_count(m),), post_reduce=lambda: train_fine.post_reduce(net))
        cls, loc, probs = step(x, labels, masks)

    `fwd_bwd(*args, *pre(*args))` must do forward + loss + backward and NOTHING collective (`pre` runs eagerly before it every step:
    the global loss normaliser); the reducer's hooks are suspended while it is captured; after every replay `reducer.finish()`
    packs / all-reduces / unpacks on its static flat buffers (gradients live at fixed addresses in the graph's pool) and a second
    graph replays `optimizer.step()`.  Gradients are never set to None between replays.

    ONE input signature is live at a time: the gradients are re-created (new addresses) by every capture, so capturing a new
    signature DROPS the graphs of every earlier one (they and their optimizer graphs hold the old addresses: replayed, they would
    write buffers p.grad no longer points to and the ranks would silently diverge); a signature that comes back is re-captured.
    Every rank must produce the same set of gradients: a parameter that got its gradient from another rank only
    (`reducer.filled`) is not part of this rank's captured optimizer graph -- that raises instead of letting the replicas drift.
    The all-reduce runs AFTER the replayed backward (the reducer is suspended inside the capture), i.e. it is not overlapped with
    backward as in the eager path: 13-18 MB per step, ~0.2 ms against >= 10 ms of step."""

    def __init__(self, fwd_bwd, reducer, optimizer, pre=None, eager_first=1, pool=None, post_reduce=None):
        super(GraphedDPStep, self).__init__(fwd_bwd, optimizer=optimizer, eager_first=eager_first, pool=pool)
        self.reducer, self.pre = reducer, pre
        # runs between the all-reduce and optimizer.step(), eagerly in the first step(s) and as the head of the captured optimizer graph
        # afterwards: the un-scaling of an fp16 net's gradients (train_fine.post_reduce) -- forward_backward scales the loss, so without
        # this hook the optimizer would step on gradients `scale` times too large
        self.post_reduce = post_reduce
        self._opt_graphs = {}

    def _params(self):
        return [p for g in self.optimizer.param_groups for p in g['params']]

    def __call__(self, *args):
        extra = tuple(self.pre(*args)) if self.pre is not None else ()
        full = tuple(args) + extra
        if self.calls < self.eager_first:          # ordinary eager step(s): momentum buffers come into being here
            self.calls += 1
            st = self._side_stream(full)
            st.wait_stream(torch.cuda.current_stream(st.device))
            with torch.cuda.stream(st):
                out = self.fn(*full)
                self.reducer.finish()
                if self.post_reduce is not None:
                    self.post_reduce()
                self.optimizer.step()
                self.optimizer.zero_grad(set_to_none=True)
            torch.cuda.current_stream(st.device).wait_stream(st)
            return out
        sig = self._sig(full)
        lr = _lr_signature(self.optimizer)
        entry = self._graphs.get(sig)
        if entry is None:
            self._graphs.clear()                   # see the class docstring: earlier signatures hold the gradient addresses of THEIR capture
            self._opt_graphs.clear()
            for p in self._params():
                p.grad = None                      # backward then creates the gradients inside the graph's pool, at fixed addresses
            self.reducer.suspended = True
            try:
                entry = self._capture(full)
            finally:
                self.reducer.suspended = False
            self._graphs[sig] = entry
        og = self._opt_graphs.get(sig)
        if og is None or og[1] != lr:              # the learning rate is baked into the optimizer kernels
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self.pool, stream=self._side_stream(full), capture_error_mode=_capture_mode()):
                if self.post_reduce is not None:
                    self.post_reduce()
                self.optimizer.step()
            og = (g, lr)
            self._opt_graphs[sig] = og
            # the two captures executed nothing: the replays below perform this call's one real step
        graph, static_in, static_out, _ = entry
        for dst, src in zip(static_in, full):
            self._copy(dst, src)
        self._replay(graph)
        self.reducer.finish()                      # eager: bucket order, static flat buffers, RCCL on the reducer's side stream
        if self.reducer.filled:
            names = len(self.reducer.filled)
            for p in self.reducer.filled:
                p.grad = None
            raise RuntimeError('GraphedDPStep: %d parameter(s) have a gradient on another rank but none in this rank\'s captured '
                               'step; the captured optimizer graph would skip them and the replicas would diverge -- run this '
                               'model with the eager GradReducer path' % names)
        self._replay(og[0])
        return static_out
