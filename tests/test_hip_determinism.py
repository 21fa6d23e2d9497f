"""Deterministic mode of the library (include/cfn_hip.h cfn_deterministic; SURVEY 8(b): "offer a deterministic mode"): every cross-workgroup
fp64 accumulation is recorded and committed in a canonical order instead of being added atomically.  Checked here: the switch itself; that an
operator's results in the mode agree with the atomic mode to fp64-summation rounding; that train-mode passes of x3d_fine, x3d_coarse and the joint
two-stream step repeat bit for bit under it; and that a kernel which accumulates from thousands of workgroups in arbitrary order (the generic
pointwise weight gradient with the workspace path off) is bit-identical over many launches in the mode."""
import os

import pytest
import torch

from test_hip_ops import DEV, ops, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture
def det():
    import cfn_hip
    prev = cfn_hip.deterministic(True)
    yield
    cfn_hip.deterministic(prev)


def test_deterministic_switch():
    import cfn_hip
    prev = cfn_hip.deterministic()
    assert cfn_hip.deterministic(True) == prev
    assert cfn_hip.deterministic() is True
    assert cfn_hip.deterministic(True) is True          # switching on twice is a no-op
    assert cfn_hip.deterministic(False) is True
    assert cfn_hip.deterministic() is False
    cfn_hip.deterministic(prev)


@pytest.mark.parametrize('N,Cin,Cout,T,H,W,act', [(2, 24, 54, 4, 56, 56, 1), (2, 108, 48, 8, 28, 28, 2), (2, 192, 432, 9, 7, 7, 0)])
def test_operator_results_agree_with_atomic_mode(N, Cin, Cout, T, H, W, act):
    """statistics, coefficient gradients and weight gradient of a pointwise conv and of the depthwise conv behind it: the canonical-order commit and the
    fp64 atomics sum the same fp32 partials, so they agree to fp64 rounding of the sums (1e-12 relative), not just to fp32"""
    import cfn_hip
    o = ops()
    x = rnd(1, N, Cin, T, H, W).to(DEV)
    w = rnd(2, Cout, Cin, 1, 1, 1, scale=(2.0 / Cin) ** 0.5).to(DEV)
    wd = (0.3 * rnd(5, Cout, 1, 3, 3, 3)).to(DEV)
    A, B = (1 + 0.2 * rnd(3, N, Cin)).to(DEV), (0.3 * rnd(4, N, Cin)).to(DEV)

    def run():
        leaves = [v.clone().requires_grad_(True) for v in (x, w, A, B, wd)]
        y, s, q = o.pwconv(leaves[0], leaves[1], leaves[2], leaves[3], act, 1, True)
        y2, s2, q2 = o.dwconv3d(y, leaves[4], None, None, 0, 1, True)
        ((y2 * y2).sum() + s.sum() + 0.1 * q.sum() + s2.sum() + 0.01 * q2.sum()).backward()
        return [s.detach().double(), q.detach().double(), s2.detach().double(), q2.detach().double()] + [v.grad.double() for v in leaves]

    prev = cfn_hip.deterministic(False)
    try:
        a = run()
        cfn_hip.deterministic(True)
        b = run()
        c = run()
    finally:
        cfn_hip.deterministic(prev)
    for u, v, w_ in zip(a, b, c):
        assert torch.equal(v, w_)                                              # the mode repeats bit for bit
        assert float((u - v).abs().max()) <= 1e-6 * float(u.abs().max()) + 1e-30    # (fp32 outputs of the casts: one ulp of fp32 at most)


def test_many_workgroup_accumulation_repeats_bit_for_bit(det, monkeypatch):
    """the generic fp32-MFMA weight gradient (split arithmetic off): hundreds of workgroups add their tiles to the same M x K addresses"""
    import cfn_hip
    prev = cfn_hip.query('cfn_pw_split_terms', 0)
    try:
        o = ops()
        x = rnd(1, 4, 96, 16, 14, 14).to(DEV).requires_grad_(True)
        w = rnd(2, 216, 96, 1, 1, 1, scale=0.1).to(DEV).requires_grad_(True)
        y, s, q = o.pwconv(x, w, None, None, 0, 1, True)
        gy = rnd(3, *y.shape).to(DEV)
        ref = None
        for _ in range(20):
            g = torch.autograd.grad((y,), (w, x), (gy,), retain_graph=True)
            if ref is None:
                ref = [v.clone() for v in g]
            assert torch.equal(g[0], ref[0]) and torch.equal(g[1], ref[1])
    finally:
        cfn_hip.query('cfn_pw_split_terms', prev)


def _repeat(step, passes):
    ref, bad = None, {}
    for _ in range(passes):
        cur = step()
        if ref is None:
            ref = cur
            continue
        for k, v in cur.items():
            if not torch.equal(v, ref[k]):
                bad[k] = bad.get(k, 0) + 1
    return bad


def test_fine_train_pass_repeats_bit_for_bit(det):
    import x3d_fine
    from oracle import spec
    net = x3d_fine.generate_model('M', n_classes=157, task='loc', base_bn_splits=1, dropout=0.0)
    spec.fill_module_(net)
    net.to(DEV).train(True)
    x = spec.rand_input(5, (2, 3, 8, 112, 112)).to(DEV)
    r = spec.rand_input(6, (2, 157, 8)).to(DEV)

    def step():
        net.zero_grad(set_to_none=True)
        y = net([x, None])
        (y * r).sum().backward()
        out = {'<logits>': y.detach().clone()}
        out.update({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
        return out
    assert not _repeat(step, 6)


def test_coarse_train_pass_repeats_bit_for_bit(det):
    from test_hip_models import _coarse_inputs, _coarse_model
    from oracle import spec
    x, feat, fm, meta, depth = _coarse_inputs(130, 2, 16, 12)
    m = _coarse_model(depth, dropout=0.0)
    m.train(True)
    m.rw6.dropout.p = 0.0
    r = spec.rand_input(131, (2, 157, 16)).to(DEV)
    inp = [x.to(DEV), {k: v.to(DEV) for k, v in feat.items()}, fm.to(DEV), 0, meta.to(DEV)]

    def step():
        m.zero_grad(set_to_none=True)
        y = m(inp)
        (y * r).sum().backward()
        out = {'<logits>': y.detach().clone()}
        out.update({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
        return out
    assert not _repeat(step, 6)


def test_joint_train_pass_repeats_bit_for_bit(det):
    from test_hip_train import _joint_nets
    from oracle import spec
    tj, fine, coarse = _joint_nets()
    fine.train(True)
    coarse.train(True)
    for mod in coarse.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    clip = spec.rand_input(7, (2, 3, 16, 224, 224)).to(DEV)
    r = spec.rand_input(8, (2, 157, 8)).to(DEV)

    def step():
        fine.zero_grad(set_to_none=True)
        coarse.zero_grad(set_to_none=True)
        logits, _ = tj.joint_forward(fine, coarse, clip)
        (logits * r).sum().backward()
        out = {'<logits>': logits.detach().clone()}
        for tag, net in (('fine.', fine), ('coarse.', coarse)):
            out.update({tag + k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
        return out
    assert not _repeat(step, 4)


def test_mode_from_the_environment():
    """CFN_DETERMINISTIC=1 switches the mode on when the library is loaded (a parity run of a whole script)"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); import cfn_hip; cfn_hip.load(); print(int(cfn_hip.deterministic()))"
            % os.path.join(os.path.dirname(here), 'coarse-fine-networks_amd'))
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, CFN_DETERMINISTIC='1'), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == '1'


def test_two_threads_are_serialised_in_deterministic_mode(det):
    """the mode keeps ONE record buffer per process and commits it per entry point: while it is on, the binding lets one thread at a time into the
    library (round 6: the two-thread model test of tests/test_hip_models.py aborted the process under CFN_DETERMINISTIC=1).  Two threads run
    forward + backward of a bottleneck stack concurrently: no error, and each thread's results equal its own sequential run bit for bit."""
    import threading
    import x3d_fine
    from oracle import spec
    dev = torch.device('cuda:0')

    def make():
        net = x3d_fine.generate_model('S', n_classes=157, task='loc', base_bn_splits=1, dropout=0.0)
        spec.fill_module_(net)
        return net.to(dev).train(True)
    xs = [spec.rand_input(70 + i, (2, 3, 8, 64, 64)).to(dev) for i in range(2)]

    def run(net, x, out):
        for _ in range(2):
            net.zero_grad(set_to_none=True)
            y = net([x, None])
            y.square().mean().backward()
        out.append((y.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}))
    ref = []
    for i in range(2):
        o = []
        run(make(), xs[i], o)
        torch.cuda.synchronize()
        ref.append(o[0])
    nets, outs, errs = [make(), make()], [[], []], []

    def worker(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                run(nets[i], xs[i], outs[i])
                torch.cuda.current_stream().synchronize()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for i in range(2):
        assert torch.equal(outs[i][0][0], ref[i][0])
        for k, g in ref[i][1].items():
            assert torch.equal(outs[i][0][1][k], g), k
