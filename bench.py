#!/usr/bin/env python3
"""Headline benchmark: clips/sec (fwd + loss + bwd + SGD step) of x3d_fine X3D-M on synthetic
Bx3x256x224x224 clips (B = 8 clips per GPU by default), one process per GPU, gradients all-reduced over RCCL.

    python bench.py --gpus 1 --steps 8 --warmup 2
    python bench.py --gpus N --steps K --warmup W          # N > 1 outside torchrun: re-launches itself as N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  ``value`` = clips all ranks processed / max-over-ranks wall time of the K
timed steps (inputs already resident in HBM).  ``roofline`` is measured live: HIP events bracket every
launch of the depthwise-conv forward kernels (the headline kernel family) on the stream they
run on; achieved = algorithmic bytes (N_in + N_out elements x 4 B + weights, per launch) / device time.
``cpu_baseline`` times the CPU oracle (plain torch fp32; 16 threads by default -- torch's CPU conv kernels stop scaling far
below the GPU box's 256 logical CPUs; CFN_CPU_THREADS overrides, `cores` in the JSON is what was used) on a bounded sample
of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'coarse-fine-networks_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch                                     # noqa: E402
import torch.distributed as dist                 # noqa: E402
import torch.optim as optim                      # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 achievable


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(frames_full, sample_frames=None, repeats=3, stream='fine'):
    """CPU oracle (oracle/x3d_ref.py, stock torch CPU ops) fwd+bwd on a bounded sample of the same workload, SURVEY 8d protocol:
    one untimed warm-up, then the MEDIAN of `repeats` timed runs.
    fine: one clip of 3 x sample_frames x 224 x 224 per run (default sample_frames = frames_full: ~8 s per run at T = 256 on 16 threads, ~35 s in all; a
    shorter sample is scaled by sample_frames / frames_full and the JSON says so).  coarse: one clip at the metric's T with T'=128 fine features (the full unit)."""
    from oracle import spec, x3d_ref
    sample_frames = sample_frames or frames_full          # default: the metric's own clip length, nothing scaled
    # torch's CPU conv kernels stop scaling (and thrash) far below the hardware threads of the GPU box: 16 threads is near
    # the measured optimum.  `cores` = threads used; the box's logical CPU count and model are stated next to it
    threads = min(os.cpu_count() or 1, int(os.environ.get('CFN_CPU_THREADS', '16')))
    torch.set_num_threads(threads)
    host = {'cores': threads, 'host_logical_cpus': os.cpu_count(), 'cpu_model': _cpu_model(), 'kind': 'port'}

    def leaves(sd):
        for k, v in sd.items():
            if v.is_floating_point() and 'running' not in k:
                v.requires_grad_(True)
        return sd

    def median(ts):
        ts = sorted(ts)
        return ts[len(ts) // 2]

    if stream == 'coarse':
        import train_coarse_fineFEAT as tc
        sd = leaves(spec.procedural_fill(spec.coarse_keys('M', 157, 1)))
        cf = frames_full                                   # the full unit at the metric's T (a 64-frame sample scaled by T overestimates -- the fusion branch does not scale with T)
        x, labels, masks, feat, fm, meta, _, _ = next(iter(tc.SyntheticCoarse(1, 1, cf)))
        ts = []
        for r in range(repeats + 1):                      # run 0 = warm-up
            for v in sd.values():
                v.grad = None
            t0 = time.time()
            y = x3d_ref.x3d_coarse_forward(sd, [x[:, 0], feat, fm, 0, meta], 'M', training=True)
            y.square().mean().backward()
            ts.append(time.time() - t0)
        dt = median(ts[1:])
        return dict(host, value=round((1.0 / dt) * cf / frames_full, 5), unit='clips/s',
                    sample='1 warm-up + median of %d runs of 1 clip 3x%dx224x224 + fine features T\'=128, fwd+bwd fp32 (%.1f s each)%s'
                           % (repeats, cf, dt, '' if cf == frames_full else ', scaled by %d/%d to T=%d' % (cf, frames_full, frames_full)))
    sd = leaves(spec.procedural_fill(spec.fine_keys('M', 157, 1)))
    ts = []
    for r in range(repeats + 1):                          # run 0 = warm-up (a quarter-length clip: thread pool, allocator, oneDNN primitives)
        T = max(sample_frames // 4, 8) if r == 0 else sample_frames
        x = spec.rand_input(r, (1, 3, T, 224, 224))
        for v in sd.values():
            v.grad = None
        t0 = time.time()
        y = x3d_ref.x3d_fine_forward(sd, x, 'M', training=True)
        y.square().mean().backward()
        ts.append(time.time() - t0)
        del y
    dt = median(ts[1:])
    return dict(host, value=round((1.0 / dt) * sample_frames / frames_full, 5), unit='clips/s',
                sample='1 warm-up (quarter-length clip) + median of %d runs of 1 clip 3x%dx224x224 fwd+bwd fp32 (%.1f s each)%s'
                       % (repeats, sample_frames, dt, '' if sample_frames == frames_full else
                          ', scaled by %d/%d to T=%d clips (linear in T: every op of the fine stream is per frame)' % (sample_frames, frames_full, frames_full)))


# kernel families whose HIP-event times make up the roofline leg (cfn_prof_*): the metric's kernel set is the depthwise-conv
# stack forward (fine stream, SURVEY 8d figure A) plus, for the coarse stream, the Grid Pool forward (figure B)
ROOFLINE_FAMILIES = {'fine': ('dwconv_fwd',), 'coarse': ('dwconv_fwd', 'gridpool', 'dense_fwd'),
                     'joint': ('dwconv_fwd', 'gridpool', 'dense_fwd')}


def coarse_roofline(dev, B, T, steps=3, warmup=1):
    """Figure (B) of the headline: depthwise-conv forward family + Grid Pool forward (dense saliency convs,
    `time_sample_fwd`) of the COARSE stream at the metric's T, timed live by the same HIP events while a few x3d_coarse train
    steps run.  Carried by the default (fine-stream) line as `roofline_coarse` so that one driver run records both figures."""
    import cfn_hip
    import train_coarse_fineFEAT as tc
    from cfn_hip import dist as cdist
    net = tc.build_model(dev, pretrained=None)
    optimizer = optim.SGD(tc.param_groups(net, 0.02), lr=0.02, momentum=0.9, weight_decay=1e-5)
    x, labels, masks, feat, fm, meta, _, _ = next(iter(tc.SyntheticCoarse(B, 1, T, seed=4321)))
    x = x[:, 0].contiguous().to(dev)
    labels, masks, fm, meta = labels.to(dev), masks.to(dev), fm.to(dev), meta.to(dev)
    feat = {k: v.to(dev) for k, v in feat.items()}
    net.train(True)
    reducer = cdist.GradReducer(net.parameters())
    for _ in range(warmup):
        tc.train_step(net, reducer, optimizer, x, labels, masks, feat, fm, meta)
    torch.cuda.synchronize()
    fams = ROOFLINE_FAMILIES['coarse']
    for f in fams:
        cfn_hip.prof_enable(f, True)
    t0 = time.perf_counter()
    for _ in range(steps):
        tc.train_step(net, reducer, optimizer, x, labels, masks, feat, fm, meta)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = {}
    ms = by = 0.0
    launches = 0
    for f in fams:
        cfn_hip.prof_enable(f, False)
        m_, n_, b_ = cfn_hip.prof_collect(f)
        per[f] = {'launches': n_, 'ms': round(m_, 3), 'GB': round(b_ / 1e9, 3)}
        ms, launches, by = ms + m_, launches + n_, by + b_
    reducer.close()
    achieved = (by / 1e9) / (ms / 1e3) if ms > 0 else 0.0
    return {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
            'kernel': 'coarse stream (x3d_coarse train step, %dx3x%dx224x224 + fine features T\'=128): depthwise conv forward family '
                      '(conv1_t + layer-1 at T, layers 2-4 at K = T/4 + 1 frames) + Grid Pool forward (salconv.hip saliency convs, 1x3x3 '
                      'conv3, time_sample_fwd)' % (B, T),
            'family': per, 'launches': launches, 'avg_launch_ms': round(ms / max(launches, 1), 4),
            'algorithmic_bytes_per_launch': round(by / max(launches, 1)), 'steps': steps,
            'coarse_ms_per_step': round(dt / steps * 1e3, 3)}


def _flat(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for k in obj:
            yield from _flat(obj[k])
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _flat(v)


def staged_leg(step, resident, dev, steps, warmup, world):
    """The same K steps with the inputs arriving from HOST memory every step: a pinned host copy of the batch is handed to
    cfn_hip.staging.HostStager each step -- what a DataLoader(pin_memory=True) / a collate into pinned memory delivers -- and reaches HBM on the copy
    stream one batch ahead of the step that consumes it.  Returns (seconds for `steps` steps, bytes per step)."""
    from cfn_hip.staging import HostStager, _map_tensors
    host = _map_tensors(resident, lambda t: t.detach().cpu().pin_memory())
    stager = HostStager(dev)

    def loader():
        for _ in range(warmup + steps):
            yield host
    it = stager.stage(loader())
    keep = []
    for _ in range(warmup):
        keep.append(step(next(it))[:2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        keep.append(step(next(it))[:2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    it.close()
    assert all(float(v) == float(v) for pair in keep[-1:] for v in pair)
    return dt, stager.bytes_staged // max(stager.batches, 1)


def _self_launch(n):
    """`python bench.py --gpus N` outside torchrun: one rank per GPU under torch.distributed.run (as train_fine._spawn)"""
    import socket
    import subprocess
    port = os.environ.get('MASTER_PORT')
    if not port:
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--stream', choices=('fine', 'coarse', 'joint'), default='fine',
                    help='fine: x3d_fine X3D-M train step at T=256 (the headline metric); coarse: x3d_coarse fineFEAT-fusion '
                         'train step on 64-frame clips + T\'=128 fine features (BASELINE configs[3] per-GPU shard); joint: both streams end '
                         'to end, fine tower on 128 frames feeding the coarse stream on the centre 64 (BASELINE configs[4] per-GPU shard)')
    ap.add_argument('--frames', type=int, default=None, help='default 256 (fine) / 64 (coarse) / 128 fine frames (joint)')
    ap.add_argument('--batch', type=int, default=8, help='clips per GPU per step')
    ap.add_argument('--dtype', choices=('f32', 'bf16', 'fp16'), default='f32',
                    help='storage type of activations / activation gradients (fine stream; coarse stream: stem + layer 1): f32 = the reference precision (the '
                         'headline), bf16 = BASELINE configs[1] (bf16 MFMA pointwise, fp32 accumulation and statistics), fp16 = BASELINE configs[4] (IEEE-half '
                         'storage, fp16 MFMA pointwise, static loss scale)')
    ap.add_argument('--graph', action='store_true', help='replay the step from captured hipGraphs (one graph on a single GPU; with --gpus > 1: forward+backward graph, eager all-reduce, optimizer graph).  '
                                                         'DEFAULT for --stream coarse / joint on one GPU (their eager step is host bound: ~1,200 launches in front of ~25 ms of kernels)')
    ap.add_argument('--eager', action='store_true', help='--stream coarse / joint: launch every kernel from Python instead of replaying a hipGraph')
    ap.add_argument('--staged', action='store_true', help='after the resident measurement, time the same steps with the inputs arriving from (pinned) HOST memory every step through '
                                                          'cfn_hip.staging (copy stream, double buffered) and report pcie_inclusive_clips_per_s beside `value`')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-coarse-roofline', action='store_true', help='skip the coarse-stream roofline leg (figure B) of the default line')
    ap.add_argument('--cpu-sample-frames', type=int, default=None, help='frames of the CPU baseline clip (default: the metric\'s T -- no scaling; a shorter sample is scaled linearly and says so)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(_self_launch(args.gpus))
    coarse, joint = args.stream == 'coarse', args.stream == 'joint'
    if (coarse or joint) and args.gpus == 1 and not args.eager:
        args.graph = True
    assert not (args.graph and args.eager), '--graph and --eager exclude each other'
    if args.frames is None:
        args.frames = 64 if coarse else (128 if joint else 256)

    from cfn_hip import dist as cdist
    import cfn_hip
    import train_fine
    import train_coarse_fineFEAT as tc
    rank, world, dev = cdist.init_from_env()
    assert world == args.gpus, '--gpus %d but WORLD_SIZE is %d: launch one rank per GPU (or let bench.py do it)' % (args.gpus, world)
    assert torch.cuda.is_available(), 'bench.py measures the HIP path; it needs a GPU'
    cfn_hip.load()

    torch.manual_seed(0)
    B, T = args.batch, args.frames
    g = torch.Generator().manual_seed(1234 + rank)
    if joint:
        import train_joint as tj
        fine_net, net = tj.build_models(dev, fine_act_dtype=args.dtype, coarse_act_dtype=args.dtype)
        groups = tj.param_groups(fine_net, net, 0.02)
        optimizer = optim.SGD(groups, lr=0.02, momentum=0.9, weight_decay=1e-5)
        x = torch.randn(B, 3, T, 224, 224, generator=g).to(dev)
        tl = (T // 2) * 10
        labels = (torch.rand(B, 157, tl, generator=g) < 0.05).float().to(dev)
        masks = torch.ones(B, tl, device=dev)
        fine_net.train(True)
        cdist.sync_module(fine_net)
    elif coarse:
        net = tc.build_model(dev, pretrained=None, act_dtype=args.dtype)
        optimizer = optim.SGD(tc.param_groups(net, 0.02), lr=0.02, momentum=0.9, weight_decay=1e-5)
        x, labels, masks, feat, fm, meta, _, _ = next(iter(tc.SyntheticCoarse(B, 1, T, seed=1234 + rank)))
        x = x[:, 0].contiguous().to(dev)
        labels, masks, fm, meta = labels.to(dev), masks.to(dev), fm.to(dev), meta.to(dev)
        feat = {k: v.to(dev) for k, v in feat.items()}
    else:
        net = train_fine.build_model(dev, pretrained=None, act_dtype=args.dtype)
        optimizer = optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-5)
        x = torch.randn(B, 3, T, 224, 224, generator=g).to(dev)
        tl = T * 10
        labels = (torch.rand(B, 157, tl, generator=g) < 0.05).float().to(dev)
        masks = torch.ones(B, tl, device=dev)
    net.train(True)
    cdist.sync_module(net)            # one model on every rank, as under DataParallel
    reducer = cdist.GradReducer([p for gr in groups for p in gr['params']] if joint else net.parameters())

    resident = (x, labels, masks, feat, fm, meta) if coarse else (x, labels, masks)

    def step(inp=None):
        inp = resident if inp is None else inp
        if joint:
            return tj.train_step(fine_net, net, reducer, optimizer, *inp)
        if coarse:
            return tc.train_step(net, reducer, optimizer, *inp)
        return train_fine.train_step(net, reducer, optimizer, *inp)


    eager_step = step
    if args.graph and world > 1:
        # two hipGraphs around the eager bucketed all-reduce (cfn_hip/graph.py GraphedDPStep): fine stream only
        assert not (coarse or joint), '--graph with --gpus > 1 covers the fine stream'
        from cfn_hip.graph import GraphedDPStep
        graphed = GraphedDPStep(lambda x_, l_, m_, tot: train_fine.forward_backward(net, x_, l_, m_, mask_total=tot)[:2], reducer, optimizer,
                                pre=lambda x_, l_, m_: (cdist.global_mask_count(m_),), post_reduce=lambda: train_fine.post_reduce(net))

        def step(inp=None):
            return tuple(v.clone() for v in graphed(*(resident if inp is None else inp)))
    elif args.graph:
        from cfn_hip.graph import GraphedStep
        graphed = GraphedStep(lambda: eager_step()[:2], optimizer=optimizer)

        def step(inp=None):           # static loss buffers are overwritten by the next replay: keep copies
            if inp is not None:       # staged batch -> the graph's resident input tensors (device to device, on the step's stream)
                torch._foreach_copy_(list(_flat(resident)), list(_flat(inp)), non_blocking=True)
            return tuple(v.clone() for v in graphed())

    losses = []                       # device scalars; read after the timed region (no sync inside it)
    for _ in range(args.warmup):
        losses.append(step()[:2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    fams = ROOFLINE_FAMILIES[args.stream]
    for f in fams:
        cfn_hip.prof_enable(f, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step()[:2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    staged = None
    if args.staged:
        sdt, sbytes = staged_leg(step, resident, dev, args.steps, max(args.warmup, 2), world)
        st = torch.tensor([sdt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
        staged = (float(st.item()), sbytes)
    roofline_from = 'the timed steps'
    if args.graph and world == 1:
        # the HIP events that time the roofline kernels are recorded by the C ABI at launch time: a replayed graph holds none.  The roofline leg of
        # a graphed line therefore comes from two EAGER steps of the same model behind the timed region (same kernels, same shapes)
        for f in fams:                    # (nothing was collected under the replays; one untimed eager step first: it allocates the
            cfn_hip.prof_enable(f, False)  # default stream's kernel workspaces and warms the allocator, which the replays never touched)
            cfn_hip.prof_collect(f)
        eager_step()
        torch.cuda.synchronize()
        for f in fams:
            cfn_hip.prof_enable(f, True)
        for _ in range(2):
            eager_step()
        torch.cuda.synchronize()
        roofline_from = '2 eager steps behind the timed (replayed) ones'
    ms = by = 0.0
    launches = 0
    for f in fams:
        cfn_hip.prof_enable(f, False)
        m_, n_, b_ = cfn_hip.prof_collect(f)
        ms, launches, by = ms + m_, launches + n_, by + b_
    loss_first = [float(v) for v in losses[0]]
    loss_last = [float(v) for v in losses[-1]]
    assert all(v == v and abs(v) != float('inf') for v in loss_first + loss_last), ('non-finite loss', loss_first, loss_last)

    if rank == 0:
        achieved = (by / 1e9) / (ms / 1e3) if ms > 0 else 0.0
        # HBM traffic of the same kernels from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
        # passes, gfx950 correction applied) is measured offline -- bench.py cannot run under the counter tool -- and
        # committed in profiles/; quoted only for the configuration it was measured on
        traffic = traffic_note = None
        import glob
        for name in sorted((os.path.basename(f) for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_dwfwd.json'))), reverse=True):     # the newest round's
            pmc = os.path.join(ROOT, 'profiles', name)
            if not coarse and args.dtype == 'f32' and os.path.exists(pmc):
                doc = json.load(open(pmc))
                if doc.get('frames') == T:   # per launch, like `achieved`; measured at doc['batch'] clips, linear in the batch
                    traffic = round(doc['traffic_bytes_per_launch'] * B / doc['batch'])
                    traffic_note = 'HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE), profiles/' + name
                break
        if joint:
            metric = 'clips/sec (fwd+bwd+SGD) joint Coarse-Fine two-stream, fine T=%d + coarse T=%d, 224x224' % (T, T // 2)
            workload = ('joint two-stream train step: x3d_fine tower (global_tower) on %dx3x%dx224x224 -> 5 feature maps -> x3d_coarse '
                        '(Grid Pool + learned fusion) on the centre %d frames, one backward through both, random-init weights' % (B, T, T // 2))
            kernel = 'depthwise conv forward family of both streams (dw3d_flat*_fwd (fp32) / dw3d_cp_fwd (bf16) / dw3d_small_fwd / dwt5_fwd_flat) + Grid Pool forward'
        elif coarse:
            metric = 'clips/sec (fwd+bwd+SGD) x3d_coarse fineFEAT fusion T=%dx224x224, T\'=128' % T
            workload = ('x3d_coarse X3D-M (Grid Pool + learned Multi-stage Fusion) train step, %dx3x%dx224x224 clips + fine features '
                        '(T\'=128, 7x7) per GPU, random-init weights' % (B, T))
            kernel = 'depthwise conv forward family (dw3d_flat*_fwd / dw3d_small_fwd / dwt5_fwd_flat) + Grid Pool forward (dense saliency convs, time_sample_fwd)'
        else:
            metric = 'clips/sec (fwd+bwd+SGD) x3d_fine X3D-M T=%dx224x224' % T
            workload = 'x3d_fine X3D-M train step (fwd+loss+bwd+SGD), %dx3x%dx224x224 clips per GPU, random-init weights' % (B, T)
            kernel = 'depthwise conv stack forward: dw3d_flat_fwd_kernel / dw3d_flat14_fwd_kernel (56/28/14 planes, stride 1), dw3d_flat_s2_fwd_kernel / dw3d_flat14to7_fwd_kernel (stride 2), dw3d_small_fwd_kernel (7x7), dwt5_fwd_flat_kernel (conv1_t)'
        out = {
            'metric': metric,
            'value': round(world * B * args.steps / dt, 4),
            'unit': 'clips/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'dtype_note': ('fp32 tensors everywhere; pointwise contractions with >= 48 channels run as split-bf16 (each fp32 operand = 3 bf16 '
                           'terms, 6 bf16 MFMAs per k-block, fp32 accumulation: an fp32-accurate product), all other arithmetic fp32'
                           if args.dtype == 'f32' else
                           ('fp16 (IEEE half) activations / activation gradients with a static loss scale, v_mfma_f32_32x32x16_f16 pointwise products, '
                            'fp32 weights, statistics and accumulation' if args.dtype == 'fp16' else
                            'bf16 activations / activation gradients, fp32 weights, statistics and accumulation')
                           + (' (Fine tower, and the stem + layer 1 of the Coarse stream -- every tensor with the clip\'s full frame count; Grid Pool, layers 2-4 at '
                              'T/4 + 1 frames, the fusion and the heads stay fp32)' if joint else
                              (' (stem + layer 1, the full-frame-count part of the stream; Grid Pool, layers 2-4 at T/4 + 1 frames, fusion and head stay fp32)' if coarse else ''))),
            'config': {'workload': workload, 'clips_per_gpu': B, 'frames': T, 'parallelism': 'dp%d' % world,
                       'launch': ('hipGraph replay' if args.graph else 'eager') +
                                 (' (two graphs around an eager bucketed all-reduce: the all-reduce runs AFTER the replayed backward, not '
                                  'overlapped with it as in the eager path)' if args.graph and world > 1 else ''),
                       'dist': cdist.describe()},
            'loss': {'first_step_cls_loc': [round(v, 6) for v in loss_first], 'last_step_cls_loc': [round(v, 6) for v in loss_last]},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                         'traffic_note': traffic_note,
                         'kernel': kernel,
                         'launches': launches, 'avg_launch_ms': round(ms / max(launches, 1), 4),
                         'algorithmic_bytes_per_launch': round(by / max(launches, 1)), 'measured_over': roofline_from},
        }
        if staged is not None:
            sdt, sbytes = staged
            out['pcie_inclusive_clips_per_s'] = round(world * B * args.steps / sdt, 4)
            out['staged'] = {'ms_per_step': round(sdt / args.steps * 1e3, 3), 'host_bytes_per_step_per_gpu': int(sbytes),
                             'h2d_GBps_needed': round(sbytes / (sdt / args.steps) / 1e9, 2),
                             'vs_resident': round((sdt / args.steps) / (dt / args.steps), 4),
                             'note': 'inputs (clip, labels, masks%s) leave PINNED host memory every step through cfn_hip.staging: one device slab per batch, '
                                     'copy stream, double buffered, event ordered (no host synchronisation); `value` above is the resident measurement'
                                     % (', 5 fine feature maps, feature masks, meta' if coarse else '')}
        if world == 1 and args.stream == 'fine' and args.dtype == 'f32' and not args.graph and not args.no_coarse_roofline and not args.staged:
            del losses
            net = optimizer = reducer = x = labels = masks = None          # the fine step's tensors go back to the allocator first
            torch.cuda.empty_cache()
            out['roofline_coarse'] = coarse_roofline(dev, B, T)
        if world == 1 and not args.no_cpu_baseline and not joint:       # joint: the CPU legs of the two streams are reported by their own runs
            out['cpu_baseline'] = cpu_baseline(T, args.cpu_sample_frames, stream=args.stream)   # joint: none (see below)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
