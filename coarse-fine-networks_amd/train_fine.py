"""train_fine -- entry point #1 of the reference (train_fine.py) on the MI355X engine.

    python train_fine.py -gpu 0,1,...            one process per listed GPU is spawned (RCCL all-reduce)
    torchrun --nproc-per-node N train_fine.py    same thing under an external launcher

Same ``run()`` keyword arguments, loss, optimiser, LR schedule, phase pattern and checkpoint dict as the
reference (train_fine.py:56-263).  The Charades JPEG pipeline is out of scope: batches come
from ``SyntheticCharades`` which yields the same structures and shapes ``mt_collate_fn`` produces
(charades_fine.py:201-224); plug a real loader in through ``run(dataloaders=...)``.

DataParallel -> one-process-per-GPU differences that are handled explicitly:
  * the loc-loss normaliser sum(masks) is taken over the GLOBAL batch (all-reduced scalar);
  * BN statistics stay per replica; rank 0's running stats are broadcast before eval / checkpoints;
  * short last batches are skipped as in the reference (train_fine.py:180-181).
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import x3d_fine                                   # noqa: E402
from cfn_hip import dist as cdist                 # noqa: E402
from cfn_hip import staging                       # noqa: E402
from apmeter import APMeter                       # noqa: E402

BS = 8
BS_UPSCALE = 1
INIT_LR = 0.01 * BS_UPSCALE
X3D_VERSION = 'M'
CHARADES_TR_SIZE = 7900
CHARADES_VAL_SIZE = 1850
NUM_CLASSES = 157


class SyntheticCharades(object):
    """Iterable of collated batches shaped like the reference loader's output for task='loc':
    inputs (B,1,3,T,H,W) fp32, labels (B,157,TL) in {0,1}, masks (B,TL), names."""

    def __init__(self, batch_size, iters, frames=64, crop=224, stride=10, seed=0, device='cpu', label_p=0.05):
        self.bs, self.iters, self.T, self.crop, self.stride = batch_size, iters, frames, crop, stride
        self.seed, self.device, self.p = seed, device, label_p

    def __len__(self):
        return self.iters

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        tl = self.T * self.stride
        for i in range(self.iters):
            x = torch.randn(self.bs, 1, 3, self.T, self.crop, self.crop, generator=g)
            labels = (torch.rand(self.bs, NUM_CLASSES, tl, generator=g) < self.p).float()
            masks = torch.ones(self.bs, tl)
            yield x, labels, masks, ['synthetic_%d' % i] * self.bs


def detection_loss(per_frame_logits, labels, masks, align_corners=True, group=None, crops=1, local_norm=False, mask_total=None):
    """cls + loc loss of train_fine.py:199-213 for one rank's shard.

    ``loc_loss`` is normalised by the GLOBAL sum(masks) and multiplied by the world size, so that the
    average of the ranks' gradients equals the gradient of the reference's gathered-batch loss.
    crops = n > 1: validation-time multi-crop, logits are (b*n, C, T) against (b, ...) labels / masks; the per-frame
    probability is the max over the n crops of a video (train_fine.py:204-207).  local_norm=True (evaluation): this
    rank's own sum(masks) and no world factor -- no collective, ranks may hold different numbers of videos.
    mask_total: the global sum(masks) computed beforehand (cfn_hip.graph.GraphedDPStep takes the collective out of the captured part)."""
    tl = labels.size(2)
    if per_frame_logits.is_cuda:
        from cfn_hip import ops
        logits = ops.time_resize(per_frame_logits, tl, align_corners)   # = F.interpolate(mode='linear', align_corners=...)
    else:
        logits = F.interpolate(per_frame_logits, tl, mode='linear', align_corners=align_corners)
    if crops > 1:
        logits = logits.view(labels.shape[0], crops, logits.shape[1], tl)
        probs = torch.max(torch.sigmoid(logits), dim=1)[0] * masks.unsqueeze(1)
    else:
        probs = torch.sigmoid(logits) * masks.unsqueeze(1)
    cls_loss = F.binary_cross_entropy(torch.max(probs, dim=2)[0], torch.max(labels, dim=2)[0])
    world = 1
    if not local_norm and torch.distributed.is_initialized():
        world = torch.distributed.get_world_size(group)
    norm = (cdist.global_mask_count(masks, group, local=local_norm) if mask_total is None else mask_total) * labels.shape[1]
    loc_loss = F.binary_cross_entropy(probs, labels, reduction='sum') / norm * world
    return cls_loss, loc_loss, probs


def _mean_ap(apm):
    v = apm.value()
    return float(v.mean()) if torch.is_tensor(v) else float(v)


def lr_warmup(init_lr, cur_steps, warmup_steps, opt):
    start_after = 1
    if cur_steps < warmup_steps and cur_steps > start_after:
        lr_scale = min(1., float(cur_steps + 1) / warmup_steps)
        for pg in opt.param_groups:
            pg['lr'] = lr_scale * init_lr


def build_model(device, n_classes=NUM_CLASSES, pretrained=None, dropout=0.5, act_dtype=None):
    net = x3d_fine.generate_model(x3d_version=X3D_VERSION, n_classes=400, n_input_channels=3, task='loc',
                                  dropout=dropout, base_bn_splits=1, t_downsample=False, extract_feat=False,
                                  act_dtype=act_dtype)
    if pretrained:      # partial state.update as train_fine.py:104-107; a missing file raises, as in the reference
        ckpt = torch.load(pretrained, map_location='cpu')
        state = net.state_dict()
        state.update(ckpt['model_state_dict'])
        net.load_state_dict(state)
    net.replace_logits(n_classes)
    return net.to(device)


def forward_backward(net, inputs, labels, masks, gamma_tau=5, mask_total=None):
    """forward + loss + backward of one shard (no collective when mask_total is given, no optimizer): the capturable part of a
    data-parallel step (cfn_hip.graph.GraphedDPStep).  With fp16 activations the loss is multiplied by the net's loss scale:
    EVERY caller runs `post_reduce(net)` between the gradient reduction and the optimizer (train_step below does; GraphedDPStep takes it
    as its `post_reduce` hook and captures it in front of the optimizer graph)."""
    masks_clip = masks[:, ::gamma_tau * 2]
    logits = net([inputs, masks_clip])
    cls_loss, loc_loss, probs = detection_loss(logits, labels, masks, True, mask_total=mask_total)
    loss = (cls_loss + loc_loss) / 2
    scaler = loss_scaler(net)
    (loss if scaler is None else scaler.scale_loss(loss)).backward()
    return cls_loss.detach(), loc_loss.detach(), probs.detach()


# fp16 activation path (x3d_fine act_dtype='fp16', BASELINE configs[4]): activation gradients are stored as IEEE half (5 exponent bits), so the
# backward pass runs on a scaled loss; weight gradients accumulate in fp64 / fp32 and are divided by the scale before the optimizer step.
# Initial scale (gradients of this loss at the logits are ~1 / (B x 157 x T)); it is halved on the device whenever a gradient overflowed.
LOSS_SCALE_FP16 = 4096.0


class LossScaler(object):
    """Loss scale of the fp16 path, kept ON THE DEVICE: no host synchronisation, so the whole sequence survives hipGraph capture.

    `scale_loss` multiplies by the current scale; `unscale_` divides the gradients by it, and if ANY gradient is inf / nan (a half-stored
    activation gradient saturated) it zeroes ALL of them -- the optimizer step that follows then applies momentum and weight decay only,
    instead of poisoning the weights -- and halves the scale; after `interval` clean steps the scale doubles (torch's GradScaler rule,
    `torch._amp_update_scale_`, without its `.item()`)."""

    def __init__(self, device, init=LOSS_SCALE_FP16, backoff=0.5, growth=2.0, interval=2000):
        self.scale = torch.full((1,), float(init), dtype=torch.float32, device=device)
        self.found_inf = torch.zeros(1, dtype=torch.float32, device=device)
        self.tracker = torch.zeros(1, dtype=torch.int32, device=device)
        self.backoff, self.growth, self.interval = backoff, growth, interval

    def scale_loss(self, loss):
        return loss * self.scale.to(loss.dtype).squeeze(0)

    def unscale_(self, params):
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        self.found_inf.zero_()
        by_kind = {}
        for g in grads:
            by_kind.setdefault((g.device, g.dtype), []).append(g)
        inv = self.scale.double().reciprocal().float()
        for gs in by_kind.values():
            torch._amp_foreach_non_finite_check_and_unscale_(gs, self.found_inf, inv)
        # all-or-nothing without a host decision and without ever multiplying an inf / nan (inf * 0 = nan): the gradients' BIT PATTERNS are
        # multiplied by an int32 0 / 1 -- two multi-tensor launches for the whole parameter set (a per-tensor nan_to_num_ was ~300 launches)
        keep = (1.0 - self.found_inf).to(torch.int32).squeeze(0)          # 1 = clean step, 0 = an overflow somewhere
        torch._foreach_mul_([g.view(torch.int32) for g in grads if g.dtype == torch.float32], keep)
        rest = [g for g in grads if g.dtype != torch.float32]
        if rest:                                                          # (no such gradient on this path; kept correct)
            for g in rest:
                g.nan_to_num_(nan=0.0, posinf=0.0, neginf=0.0)
            torch._foreach_mul_(rest, (1.0 - self.found_inf).squeeze(0))
        torch._amp_update_scale_(self.scale, self.tracker, self.found_inf, self.growth, self.backoff, self.interval)


def loss_scale(*nets):
    """the INITIAL loss scale of a set of nets (1 unless one of them stores fp16 activations)"""
    return LOSS_SCALE_FP16 if any(getattr(n, 'act_dtype', None) == torch.float16 for n in nets) else 1.0


def loss_scaler(*nets):
    """the LossScaler shared by `nets` (created on first use, kept on the first fp16 net); None when no net stores fp16 activations"""
    for n in nets:
        if getattr(n, 'act_dtype', None) == torch.float16:
            sc = getattr(n, '_cfn_loss_scaler', None)
            if sc is None:
                sc = LossScaler(next(n.parameters()).device)
                object.__setattr__(n, '_cfn_loss_scaler', sc)
            return sc
    return None


def unscale_grads(params, scale):
    """scale: a LossScaler (overflow-checked, see there), None / 1.0 (nothing to do) or a plain factor"""
    if isinstance(scale, LossScaler):
        scale.unscale_(list(params))
    elif scale is not None and scale != 1.0:
        grads = [p.grad for p in params if p.grad is not None]
        if grads:
            torch._foreach_mul_(grads, 1.0 / scale)


def post_reduce(net):
    """what has to run between the gradient reduction and optimizer.step() -- the other half of forward_backward's loss scale"""
    unscale_grads(net.parameters(), loss_scaler(net))


def train_step(net, reducer, optimizer, inputs, labels, masks, gamma_tau=5, pre_step=None):
    """one optimisation step on this rank's shard; returns (cls_loss, loc_loss, probs).  pre_step() runs between the
    gradient reduction and optimizer.step() -- where the reference adjusts the warm-up learning rate
    (train_fine.py:241-244)."""
    reducer.begin_pass()
    cls_loss, loc_loss, probs = forward_backward(net, inputs, labels, masks, gamma_tau)
    reducer.finish()
    post_reduce(net)
    if pre_step is not None:
        pre_step()
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    return cls_loss, loc_loss, probs


def _ap_rows(probs, labels, valid_t):
    """per video: (scores (v,157), targets (v,157)) numpy over the valid frames -- what APMeter.add takes"""
    rows = []
    for i in range(labels.shape[0]):
        v = int(valid_t[i])
        rows.append((probs[i][:, :v].transpose(0, 1).cpu().numpy(), labels[i][:, :v].transpose(0, 1).cpu().numpy()))
    return rows


def run(init_lr=INIT_LR, warmup_steps=0, max_epochs=200, mode='rgb', root=None, train_split=None,
        batch_size=BS * BS_UPSCALE, frames=80 * 4, dataloaders=None, max_steps=None, save_model='models/fine_charades_',
        pretrained='models/x3d_multigrid_kinetics_fb_pretrained.pt', log=print, phase_hook=None):
    rank, world, dev = cdist.init_from_env()
    gamma_tau = {'S': 6, 'M': 5, 'XL': 5}[X3D_VERSION]
    crop = {'S': 160, 'M': 224, 'XL': 312}[X3D_VERSION]
    clip_frames = frames * 2 // (gamma_tau * 2)          # 640-frame window at stride 10 -> 64 frames
    local_bs = max(batch_size // world, 1)
    iters = CHARADES_TR_SIZE // batch_size
    val_bs = max(batch_size // 2 // world, 1)
    val_iters = CHARADES_VAL_SIZE // max(batch_size // 2, 1)
    if dataloaders is None:
        dataloaders = {'train': SyntheticCharades(local_bs, iters, clip_frames, crop, gamma_tau * 2, seed=rank),
                       'val': SyntheticCharades(val_bs, val_iters, clip_frames, crop, gamma_tau * 2, seed=1000 + rank)}
    net = build_model(dev, pretrained=pretrained)
    cdist.sync_module(net)        # one model on every rank (DataParallel replicates rank 0's; fc2 was just re-drawn)
    optimizer = optim.SGD(net.parameters(), lr=init_lr, momentum=0.9, weight_decay=1e-5)
    lr_sched = optim.lr_scheduler.MultiStepLR(optimizer, [15, 20, 25])
    reducer = cdist.GradReducer(net.parameters())
    tr_apm, val_apm = APMeter(), APMeter()
    # batches reach HBM through a pinned slab on a copy stream, one batch ahead of the step (the reference: a synchronous `.cuda()` per tensor
    # in front of every step, train_fine.py:184-197); the `.to(dev)` calls below are no-ops on staged batches
    stager = staging.HostStager(dev) if dev.type == 'cuda' else None
    steps, epochs = 0, 0
    while epochs < max_epochs:
        for phase in 4 * ['train'] + ['val']:
            train = phase == 'train'
            net.train(train)
            if train:
                epochs += 1
            else:
                cdist.broadcast_buffers(net)
                net.aggregate_sub_bn_stats()
            tot_loc = tot_cls = 0.0
            n_it = 0
            val_rows = []
            for inputs, labels, masks, _name in (stager.stage(dataloaders[phase]) if stager else dataloaders[phase]):
                want = local_bs if train else val_bs
                ok = inputs.shape[0] == want
                if train:      # the skip is a collective decision: no rank may leave the others alone in an all-reduce
                    ok = cdist.all_agree(ok, dev) if world > 1 else ok
                if not ok:
                    continue
                b, n = inputs.shape[:2]          # n crops per video (1 in training, train_fine.py:176-185)
                inputs = inputs.view((b * n,) + tuple(inputs.shape[2:])).to(dev, non_blocking=True)
                labels, masks = labels.to(dev), masks.to(dev)
                valid_t = masks.sum(1).int()
                n_it += 1
                if train:
                    warm = (lambda: lr_warmup(init_lr, steps, warmup_steps, optimizer))
                    cls_loss, loc_loss, probs = train_step(net, reducer, optimizer, inputs, labels, masks, gamma_tau, pre_step=warm)
                    steps += 1
                    for sc, tg in _ap_rows(probs, labels, valid_t):
                        tr_apm.add(sc, tg)
                else:
                    with torch.no_grad():
                        logits = net([inputs, masks[:, ::gamma_tau * 2]])
                        cls_loss, loc_loss, probs = detection_loss(logits, labels, masks, True, crops=n, local_norm=True)
                    val_rows.extend(_ap_rows(probs, labels, valid_t))
                tot_cls += float(cls_loss)
                tot_loc += float(loc_loss)
                if train and steps % max(iters // 2, 1) == 0:
                    m_loc, m_cls = cdist.mean_over_ranks([tot_loc / n_it, tot_cls / n_it], dev)
                    if rank == 0:
                        log(' Epoch:{} {} steps: {} Loc Loss: {:.4f} Cls Loss: {:.4f} mAP: {:.4f}'.format(
                            epochs, phase, steps, m_loc, m_cls, _mean_ap(tr_apm)))
                    tr_apm.reset()
                if train and steps % 1000 == 0 and rank == 0:
                    os.makedirs(os.path.dirname(save_model) or '.', exist_ok=True)
                    torch.save({'model_state_dict': net.state_dict(), 'optimizer_state_dict': optimizer.state_dict(),
                                'scheduler_state_dict': lr_sched.state_dict()}, save_model + str(steps).zfill(6) + '.pt')
                if max_steps is not None and steps >= max_steps:
                    return net
            if not train:
                # every rank validated its own shard: the reported mAP / losses cover the WHOLE validation set
                gathered = cdist.gather_objects((val_rows, tot_loc, tot_cls, n_it))
                if rank == 0:
                    g_loc = g_cls = 0.0
                    g_it = 0
                    for rows, r_loc, r_cls, r_it in gathered:
                        for sc, tg in rows:
                            val_apm.add(sc, tg)
                        g_loc, g_cls, g_it = g_loc + r_loc, g_cls + r_cls, g_it + r_it
                    log(' Epoch:{} val Loc Loss: {:.4f} Cls Loss: {:.4f} mAP: {:.4f}'.format(
                        epochs, g_loc / max(g_it, 1), g_cls / max(g_it, 1), _mean_ap(val_apm)))
                val_apm.reset()
                lr_sched.step()            # once per val phase, as the reference (not per epoch, not per step)
            if phase_hook is not None:         # test / logging hook: end of a phase (not in the reference's signature)
                phase_hook(phase, epochs, optimizer)
    return net


def _spawn(gpus, argv):
    """`-gpu 0,1,2` -> one worker process per GPU (what nn.DataParallel did with threads)."""
    import subprocess
    env = dict(os.environ, CUDA_VISIBLE_DEVICES=gpus, HSA_ENABLE_IPC_MODE_LEGACY='0')
    n = len(gpus.split(','))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', os.environ.get('MASTER_PORT', '29511'),
           os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('-gpu', default='0', type=str)
    parser.add_argument('--max-steps', type=int, default=None)
    parser.add_argument('--batch-size', type=int, default=BS * BS_UPSCALE)
    args = parser.parse_args()
    if 'RANK' not in os.environ and len(args.gpu.split(',')) > 1:
        sys.exit(_spawn(args.gpu, ['--batch-size', str(args.batch_size)] +
                        (['--max-steps', str(args.max_steps)] if args.max_steps else [])))
    if 'RANK' not in os.environ:
        os.environ['CUDA_VISIBLE_DEVICES'] = args.gpu
    run(batch_size=args.batch_size, max_steps=args.max_steps)
