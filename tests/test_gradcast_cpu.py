"""The lazily cast weight gradients of cfn_hip.ops (_GradCast) on CPU tensors: the hand-out rules that keep a gradient from
being read before it is filled.  No HIP calls: a stand-in autograd Function uses the same buffers the
conv ops use."""
import pytest
import torch

from cfn_hip import ops


class _Scale(torch.autograd.Function):
    """y = w * x with the weight gradient produced the way the conv ops do: fp64 accumulator + deferred fp32 view"""

    @staticmethod
    def forward(ctx, x, w, boom):
        ctx.save_for_backward(x)
        ctx.wparam, ctx.boom = w, boom
        return x * w

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        g64, fin = ops._gw_buffers(ctx.wparam, 1, x.numel(), x.device)
        g64.view(-1).add_((gy * x).double().view(-1))
        if ctx.boom:
            raise RuntimeError('backward failed')
        return gy * ctx.wparam, fin(), None


def _param(vals, lazy=True):
    w = torch.nn.Parameter(torch.tensor(vals))
    if lazy:
        ops.allow_lazy_grad_cast([w])      # what GradReducer does for its parameters
    return w


def test_lazy_view_is_filled_at_end_of_backward():
    w = _param([1.0, 2.0, 3.0])
    assert ops._lazy_ok_probe(w)
    x = torch.tensor([2.0, 2.0, 2.0], requires_grad=True)
    _Scale.apply(x, w, False).sum().backward()
    assert torch.equal(w.grad, torch.tensor([2.0, 2.0, 2.0]))


def test_weight_used_twice_in_one_graph_sums_both_terms():
    """autograd adds the two gradients mid-pass: the second use must not get an unfilled view"""
    w = _param([1.0, 2.0, 3.0])
    x1 = torch.tensor([2.0, 2.0, 2.0], requires_grad=True)
    x2 = torch.tensor([5.0, 6.0, 7.0], requires_grad=True)
    (_Scale.apply(x1, w, False).sum() + _Scale.apply(x2, w, False).sum()).backward()
    assert torch.equal(w.grad, torch.tensor([7.0, 8.0, 9.0]))


def test_foreign_post_accumulate_hook_sees_a_valid_gradient():
    w = _param([1.0, 2.0, 3.0])
    seen = []
    w.register_post_accumulate_grad_hook(lambda p: seen.append(p.grad.clone()))
    x = torch.tensor([2.0, 3.0, 4.0], requires_grad=True)
    _Scale.apply(x, w, False).sum().backward()
    assert torch.equal(seen[0], torch.tensor([2.0, 3.0, 4.0]))


def test_failed_backward_does_not_poison_the_next_pass():
    w = _param([1.0, 2.0, 3.0])
    x = torch.tensor([2.0, 2.0, 2.0], requires_grad=True)
    with pytest.raises(RuntimeError):
        _Scale.apply(x, w, True).sum().backward()
    w.grad = None
    _Scale.apply(x, w, False).sum().backward()
    assert torch.equal(w.grad, torch.tensor([2.0, 2.0, 2.0]))


def test_eager_cast_switch():
    ops.LAZY_GRAD_CAST = False
    try:
        w = _param([1.0, 2.0])
        x = torch.tensor([3.0, 4.0], requires_grad=True)
        _Scale.apply(x, w, False).sum().backward()
        assert torch.equal(w.grad, torch.tensor([3.0, 4.0]))
    finally:
        ops.LAZY_GRAD_CAST = True


def test_parameter_that_nobody_opted_in_gets_an_immediate_cast():
    """hooks on the AccumulateGrad NODE (torch DDP's reducer, grad_accumulator.register_hook users) are
    invisible in the parameter's hook dicts, so the lazy cast is opt-in: without allow_lazy_grad_cast such a hook reads a
    filled gradient."""
    w = _param([1.0, 2.0, 3.0], lazy=False)
    assert not ops._lazy_ok_probe(w)
    x = torch.tensor([2.0, 3.0, 4.0], requires_grad=True)
    y = _Scale.apply(x, w, False)
    acc = y.grad_fn.next_functions[1][0]          # w's AccumulateGrad node, as DDP's Reducer finds it
    assert type(acc).__name__ == 'AccumulateGrad'
    seen = []
    acc.register_hook(lambda *_: seen.append(w.grad.clone()))
    y.sum().backward()
    assert torch.equal(seen[0], torch.tensor([2.0, 3.0, 4.0]))


def test_grad_reducer_opts_its_parameters_in():
    from cfn_hip import dist as cdist
    w = _param([1.0, 2.0, 3.0], lazy=False)
    cdist.GradReducer([w])
    assert ops._lazy_ok_probe(w)


def test_reshaped_view_of_a_parameter_gets_the_lazy_view_too():
    """round 6: the fusion modules hand `conv1d.weight.view(O, I, 1, 1, 1)` to the conv op (x3d_coarse._w5) -- not a leaf, so every such weight
    gradient used to be cast on the spot (one conversion kernel per conv and step, 38 per coarse step).  A pure reshape of a whole leaf follows the
    leaf's rules: lazy view, filled at the end of the pass; a second use in the same pass still gets a valid gradient; a SLICE does not qualify."""
    w = _param([1.0, 2.0, 3.0, 4.0])
    x = torch.tensor([[2.0, 2.0], [3.0, 3.0]], requires_grad=True)
    wv = w.view(2, 2)
    assert ops._grad_owner(wv) is w
    with torch.no_grad():
        assert ops._lazy_ok(ops._grad_owner(wv))
    _Scale.apply(x, wv, False).sum().backward()
    assert torch.equal(w.grad, torch.tensor([2.0, 2.0, 3.0, 3.0]))
    w.grad = None
    (_Scale.apply(x, w.view(2, 2), False).sum() + _Scale.apply(2 * x.detach(), w.view(2, 2), False).sum()).backward()
    assert torch.equal(w.grad, torch.tensor([6.0, 6.0, 9.0, 9.0]))
    sl = w[:2]
    assert ops._grad_owner(sl) is sl                       # part of a parameter: ordinary (immediate) cast
    w.grad = None
    _Scale.apply(torch.tensor([5.0, 7.0], requires_grad=True), sl, False).sum().backward()
    assert torch.equal(w.grad, torch.tensor([5.0, 7.0, 0.0, 0.0]))
