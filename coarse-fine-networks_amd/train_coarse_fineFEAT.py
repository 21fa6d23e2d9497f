"""train_coarse_fineFEAT -- entry point #2 of the reference (train_coarse_fineFEAT.py) on the MI355X engine:
trains the Coarse stream + Multi-stage Fusion on clips and pre-extracted fine features.

    python train_coarse_fineFEAT.py -gpu 0,1,...      (one process per GPU, RCCL gradient all-reduce)

Kept from the reference (train_coarse_fineFEAT.py:59-301): ``run()`` kwargs, model constructor arguments
(:107-109), the two parameter groups with 10x learning rate on every parameter whose name contains 'rw' or
'mix' (:137-141), SGD(0.9, 1e-5), MultiStepLR([15,25,35]) stepped once per val phase, the 2 x train + val
phase pattern, the loss with ``F.interpolate(..., mode='linear')`` WITHOUT align_corners (:226, unlike the
fine script), val-time chunking of videos longer than 1005 frames (:215-224), the 25-frame Charades_v1_localize
CSV rows (:249-263) and the checkpoint dict (:289-293).  The JPEG / feature-file dataset is out of scope
(SURVEY 2.1); ``SyntheticCoarse`` yields batches with the structure ``mt_collate_fn`` builds
(charades_coarse_fineFEAT.py:208-252), SURVEY 8d cfg4 conventions."""
import argparse
import csv
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import x3d_coarse                                 # noqa: E402
from cfn_hip import staging                       # noqa: E402
from cfn_hip import dist as cdist                 # noqa: E402
from apmeter import APMeter                       # noqa: E402
from train_fine import lr_warmup                  # noqa: E402

BS = 6
BS_UPSCALE = 1
INIT_LR = 0.02 * BS_UPSCALE
X3D_VERSION = 'M'
CHARADES_TR_SIZE = 7900
CHARADES_VAL_SIZE = 1850
NUM_CLASSES = 157
FEAT_KEYS = ('layer1', 'layer2', 'layer3', 'layer4', 'conv5')
FEAT_DEPTH = {'layer1': 24, 'layer2': 48, 'layer3': 96, 'layer4': 192, 'conv5': 432}


class SyntheticCoarse(object):
    """collated batches: inputs (B,1,3,T,224,224), labels (B,157,TL), masks (B,TL), feat{k: (B,C_k,T',7,7) >= 0},
    feat_masks (B,T'), meta (B,4) int64 [start/10, frames/10, nf/10, stride/10], names, durations."""

    def __init__(self, batch_size, iters, frames=64, fine_len=128, crop=224, stride=10, seed=0):
        self.bs, self.iters, self.T, self.Tf, self.crop, self.stride, self.seed = batch_size, iters, frames, fine_len, crop, stride, seed

    def __len__(self):
        return self.iters

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        tl = self.T * self.stride
        for i in range(self.iters):
            x = torch.randn(self.bs, 1, 3, self.T, self.crop, self.crop, generator=g)
            labels = (torch.rand(self.bs, NUM_CLASSES, tl, generator=g) < 0.05).float()
            masks = torch.ones(self.bs, tl)
            feat = {k: torch.relu(torch.randn(self.bs, c, self.Tf, 7, 7, generator=g)) for k, c in FEAT_DEPTH.items()}
            fm = torch.ones(self.bs, self.Tf)
            meta = torch.zeros(self.bs, 4, dtype=torch.int64)
            for b in range(self.bs):
                nf = int(torch.randint(min(self.T, self.Tf), self.Tf + 1, (1,), generator=g))
                fm[b, nf:] = 0
                start = int(torch.randint(0, max(nf - self.T, 0) + 1, (1,), generator=g))
                meta[b] = torch.tensor([start, self.T, nf, 1])
            yield x, labels, masks, feat, fm, meta, ['synthetic_%d' % i] * self.bs, torch.full((self.bs,), tl / 24.0)


def _mean_ap(apm):
    v = apm.value()
    return float(v.mean()) if torch.is_tensor(v) else float(v)


def detection_loss(per_frame_logits, labels, masks, group=None, crops=1, local_norm=False):
    """train_coarse_fineFEAT.py:226-240 (F.interpolate WITHOUT align_corners); multi-crop / normaliser conventions as in
    train_fine.detection_loss"""
    from train_fine import detection_loss as _loss
    return _loss(per_frame_logits, labels, masks, False, group, crops, local_norm)


def build_model(device, n_classes=NUM_CLASSES, pretrained=None, dropout=0.5, act_dtype=None):
    """act_dtype 'bf16' / 'fp16': 16-bit stem + layer 1 (x3d_coarse.ResNet); None / 'f32': the reference's fp32"""
    net = x3d_coarse.generate_model(x3d_version=X3D_VERSION, n_classes=400, n_input_channels=3, feat_depth=FEAT_DEPTH,
                                    task='loc', dropout=dropout, base_bn_splits=1, learnedMixing=True, isMixing=True,
                                    t_pool='grid', act_dtype=act_dtype)
    if pretrained:      # a missing file raises, as in the reference (train_coarse_fineFEAT.py:112)
        ckpt = torch.load(pretrained, map_location='cpu')
        state = net.state_dict()
        state.update(ckpt['model_state_dict'])
        net.load_state_dict(state)
    net.replace_logits(n_classes)
    return net.to(device)


def param_groups(net, lr):
    """names containing 'rw' or 'mix' train with 10x learning rate (train_coarse_fineFEAT.py:137-141)"""
    rw, base = [], []
    for name, p in net.named_parameters():
        (rw if ('rw' in name or 'mix' in name) else base).append(p)
    return [{'params': base}, {'params': rw, 'lr': lr * 10}]


def forward_video(net, inputs, feat, feat_masks, i, meta, t_lim=1000):
    """whole-video inference with the reference's chunking of long videos (:215-224)"""
    if inputs.shape[2] < t_lim + 5:
        return net([inputs, feat, feat_masks, i, meta])
    outs = []
    meta = meta.clone()
    for t_ind in range(0, inputs.shape[2] // t_lim + 1):
        chunk = inputs[:, :, t_ind * t_lim:min(inputs.shape[2], (t_ind + 1) * t_lim)].contiguous()
        outs.append(net([chunk, feat, feat_masks, i, meta]))
        meta[:, 0] += t_lim
    return torch.cat(outs, dim=2)


def train_step(net, reducer, optimizer, inputs, labels, masks, feat, feat_masks, meta, i=0, pre_step=None):
    logits = net([inputs, feat, feat_masks, i, meta])
    cls_loss, loc_loss, probs = detection_loss(logits, labels, masks)
    from train_fine import loss_scaler, unscale_grads
    scaler = loss_scaler(net)      # fp16 stem + layer 1: device-side loss scale (train_fine.LossScaler); None otherwise
    loss = (cls_loss + loc_loss) / 2
    reducer.begin_pass()
    (loss if scaler is None else scaler.scale_loss(loss)).backward()
    reducer.finish()
    unscale_grads(net.parameters(), scaler)
    if pre_step is not None:       # warm-up learning rate is set before optimizer.step() (train_coarse_fineFEAT.py:274-277)
        pre_step()
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    return cls_loss.detach(), loc_loss.detach(), probs.detach()


def localize_rows(probs, labels, valid_t, names, dur):
    """Charades_v1_localize rows (:249-263): 25 equally spaced frames per video.
    -> list of (csv rows [name, time, '157 scores'], scores (25,157), targets (25,157))"""
    out = []
    for bb in range(labels.shape[0]):
        v = int(valid_t[bb])
        step = max(int(v / 25.), 1)
        p1 = probs[bb][:, :v][:, 1::step][:, :25]
        l1 = labels[bb][:, :v][:, 1::step][:, :25]
        a = p1.transpose(0, 1).cpu().numpy()
        rows = [[names[0], 1 + r * float(dur[bb]) / 25., ' '.join(str(s) for s in a[r])] for r in range(a.shape[0])]
        out.append((rows, a, l1.transpose(0, 1).cpu().numpy()))
    return out


def run(init_lr=INIT_LR, warmup_steps=0, max_epochs=200, mode='rgb', root=None, train_split=None,
        batch_size=BS * BS_UPSCALE, frames=80 * 4, dataloaders=None, max_steps=None,
        save_model='models/coarse_fineFEAT_charades_', pretrained='models/x3d_multigrid_kinetics_fb_pretrained.pt',
        csv_path='localize_corr_v1.csv', log=print, phase_hook=None):
    rank, world, dev = cdist.init_from_env()
    gamma_tau = 5
    clip_frames = frames * 2 // (gamma_tau * 2)
    local_bs = max(batch_size // world, 1)
    iters = CHARADES_TR_SIZE // batch_size
    if dataloaders is None:
        dataloaders = {'train': SyntheticCoarse(local_bs, iters, clip_frames, seed=rank),
                       'val': SyntheticCoarse(1, CHARADES_VAL_SIZE // world, clip_frames, seed=1000 + rank)}
    net = build_model(dev, pretrained=pretrained)
    cdist.sync_module(net)   # rw2-6, mix2-5, pool_1 and the new fc2 are not in the checkpoint: rank 0's draw everywhere
    optimizer = optim.SGD(param_groups(net, init_lr), lr=init_lr, momentum=0.9, weight_decay=1e-5)
    lr_sched = optim.lr_scheduler.MultiStepLR(optimizer, [15, 25, 35])
    reducer = cdist.GradReducer(net.parameters())
    tr_apm, val_apm = APMeter(), APMeter()
    writer = write_file = None
    if rank == 0 and csv_path:
        write_file = open(csv_path, 'w', newline='\n')
        writer = csv.writer(write_file)
    # clip, labels, masks, the five fine feature maps, their masks and meta travel as ONE pinned slab on a copy stream, one batch ahead of the
    # step (the reference: a synchronous `.cuda()` per tensor, train_coarse_fineFEAT.py:205-224); the `.to(dev)` calls below are then no-ops
    stager = staging.HostStager(dev) if dev.type == 'cuda' else None
    steps, epochs = 0, 0
    while epochs < max_epochs:
        for phase in 2 * ['train'] + ['val']:
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
            for i, (inputs, labels, masks, feat, feat_masks, meta, name, dur) in enumerate(stager.stage(dataloaders[phase]) if stager else dataloaders[phase]):
                if train:     # collective skip of a short last batch (:193-194)
                    ok = inputs.shape[0] == local_bs
                    if not (cdist.all_agree(ok, dev) if world > 1 else ok):
                        continue
                b, n = inputs.shape[:2]            # n crops per video at validation time (:198-201)
                inputs = inputs.view((b * n,) + tuple(inputs.shape[2:])).to(dev, non_blocking=True)
                labels, masks, feat_masks, meta = labels.to(dev), masks.to(dev), feat_masks.to(dev), meta.to(dev)
                feat = {k: v.to(dev) for k, v in feat.items()}
                valid_t = masks.sum(1).int()
                n_it += 1
                if train:
                    warm = (lambda: lr_warmup(init_lr, steps, warmup_steps, optimizer))
                    cls_loss, loc_loss, probs = train_step(net, reducer, optimizer, inputs, labels, masks, feat, feat_masks, meta,
                                                           i, pre_step=warm)
                    steps += 1
                    for bb in range(labels.shape[0]):
                        v = int(valid_t[bb])
                        tr_apm.add(probs[bb][:, :v].transpose(0, 1).cpu().numpy(), labels[bb][:, :v].transpose(0, 1).cpu().numpy())
                else:
                    with torch.no_grad():
                        logits = forward_video(net, inputs, feat, feat_masks, i, meta)
                        cls_loss, loc_loss, probs = detection_loss(logits, labels, masks, crops=n, local_norm=True)
                    val_rows.extend(localize_rows(probs, labels, valid_t, name, dur))
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
                    if write_file:
                        write_file.close()
                    return net
            if not train:
                # each rank evaluated its shard of the videos; rank 0 writes the CSV and the mAP for ALL of them
                gathered = cdist.gather_objects((val_rows, tot_loc, tot_cls, n_it))
                if rank == 0:
                    g_loc = g_cls = 0.0
                    g_it = 0
                    for rows_r, r_loc, r_cls, r_it in gathered:
                        for rows, sc, tg in rows_r:
                            if writer is not None:
                                writer.writerows(rows)
                            val_apm.add(sc, tg)
                        g_loc, g_cls, g_it = g_loc + r_loc, g_cls + r_cls, g_it + r_it
                    log(' Epoch:{} val Loc Loss: {:.4f} Cls Loss: {:.4f} mAP: {:.4f}'.format(
                        epochs, g_loc / max(g_it, 1), g_cls / max(g_it, 1), _mean_ap(val_apm)))
                if write_file:          # the reference closes the CSV after the first val phase (:295)
                    write_file.close()
                    write_file = writer = None
                val_apm.reset()
                lr_sched.step()            # once per val phase, as the reference (not per epoch, not per step)
            if phase_hook is not None:         # test / logging hook: end of a phase (not in the reference's signature)
                phase_hook(phase, epochs, optimizer)
    return net


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('-gpu', default='0', type=str)
    parser.add_argument('--max-steps', type=int, default=None)
    parser.add_argument('--batch-size', type=int, default=BS * BS_UPSCALE)
    args = parser.parse_args()
    if 'RANK' not in os.environ and len(args.gpu.split(',')) > 1:
        from train_fine import _spawn
        import subprocess
        env = dict(os.environ, CUDA_VISIBLE_DEVICES=args.gpu, HSA_ENABLE_IPC_MODE_LEGACY='0')
        n = len(args.gpu.split(','))
        sys.exit(subprocess.call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
                                  '--master-addr', '127.0.0.1', '--master-port', os.environ.get('MASTER_PORT', '29512'),
                                  os.path.abspath(__file__), '--batch-size', str(args.batch_size)] +
                                 (['--max-steps', str(args.max_steps)] if args.max_steps else []), env=env))
    if 'RANK' not in os.environ:
        os.environ['CUDA_VISIBLE_DEVICES'] = args.gpu
    run(batch_size=args.batch_size, max_steps=args.max_steps)
