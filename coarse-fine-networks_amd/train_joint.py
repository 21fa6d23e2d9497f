"""train_joint -- Coarse and Fine streams trained END TO END in one graph (BASELINE.json configs[4]).

The reference trains the Coarse stream on fine features extracted beforehand and stored on disk
(extract_fineFEAT.py -> train_coarse_fineFEAT.py); its README.md:31 notes that the two streams can also be trained
jointly.  Here the Fine stream runs as the feature tower (``x3d_fine.generate_model(..., global_tower=True)``,
x3d_fine.py:339-363) on the long fine clip, its five (B,C,T',7,7) maps go straight into the Coarse stream's
Multi-stage Fusion (x3d_coarse.py:628-727) -- no disk round trip, no detach -- and ONE backward pass reaches both
parameter sets; one ``GradReducer`` averages both over the ranks.

    python train_joint.py -gpu 0,1,...          one process per GPU (RCCL gradient all-reduce)

Time bases: both streams sample frames at the same stride, so coarse frame j of a window that starts at
fine index s corresponds to fine index s + j; ``meta = [s, T_coarse, T_fine, 1]`` is what the Gaussian alignment reads
(x3d_coarse.py:259,275).  By default the coarse clip is the centre T_fine/2 frames of the fine clip.
"""
import argparse
import os
import sys

import torch
import torch.optim as optim

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import x3d_fine                                   # noqa: E402
import train_coarse_fineFEAT as tc
import train_fine                # noqa: E402
from cfn_hip import staging                       # noqa: E402
from cfn_hip import dist as cdist                 # noqa: E402
from train_fine import lr_warmup                  # noqa: E402

BS = 8
INIT_LR = 0.02
NUM_CLASSES = 157


class SyntheticJoint(object):
    """collated batches: fine clip (B,3,T_fine,224,224), labels (B,157,TL), masks (B,TL) for the coarse window"""

    def __init__(self, batch_size, iters, fine_frames=128, coarse_frames=64, crop=224, stride=10, seed=0):
        self.bs, self.iters, self.Tf, self.Tc, self.crop, self.stride, self.seed = (batch_size, iters, fine_frames,
                                                                                   coarse_frames, crop, stride, seed)

    def __len__(self):
        return self.iters

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        tl = self.Tc * self.stride
        for _ in range(self.iters):
            x = torch.randn(self.bs, 3, self.Tf, self.crop, self.crop, generator=g)
            labels = (torch.rand(self.bs, NUM_CLASSES, tl, generator=g) < 0.05).float()
            yield x, labels, torch.ones(self.bs, tl)


def build_models(device, pretrained_fine=None, pretrained_coarse=None, dropout=0.5, fine_act_dtype=None, coarse_act_dtype=None):
    """(fine tower, coarse net).  The tower has no classifier of its own on this path (fc1 / fc2 get no gradient).
    fine_act_dtype='fp16' / 'bf16': the Fine stream -- 2/3 of the joint step's bytes and flops -- stores its activations as IEEE half / bf16 and
    runs its pointwise convs on v_mfma_f32_32x32x16_{f16,bf16} (BASELINE configs[4] "fp16 MFMA pointwise"; both kinds run at the same MFMA rate
    on CDNA4; fp16 adds a static loss scale, train_step); its pooled feature maps are fp32, so the Coarse stream and the fusion are unchanged."""
    fine = x3d_fine.generate_model(x3d_version='M', n_classes=NUM_CLASSES, n_input_channels=3, task='loc', dropout=dropout,
                                   base_bn_splits=1, global_tower=True, act_dtype=fine_act_dtype)
    if pretrained_fine:
        state = fine.state_dict()
        state.update(torch.load(pretrained_fine, map_location='cpu')['model_state_dict'])
        fine.load_state_dict(state)
    # coarse_act_dtype: the stem + layer 1 of the Coarse stream in 16 bits as well (x3d_coarse.ResNet) -- the joint step then runs 16-bit in both
    # trunks wherever a tensor has the clip's full frame count
    coarse = tc.build_model(device, pretrained=pretrained_coarse, dropout=dropout, act_dtype=coarse_act_dtype)
    return fine.to(device), coarse


def coarse_window(clip, coarse_frames=None, start=None):
    """(coarse clip, start index): the centre `coarse_frames` (default T_fine/2) frames of the fine clip"""
    tf = clip.shape[2]
    tcn = coarse_frames or tf // 2
    s = (tf - tcn) // 2 if start is None else start
    return clip[:, :, s:s + tcn].contiguous(), s


def joint_forward(fine, coarse, clip, coarse_frames=None, start=None):
    """fine tower on the whole clip -> feature dict -> coarse stream on the window; returns (logits (B,157,T_coarse), feat)"""
    b, _, tf = clip.shape[:3]
    xc, s = coarse_window(clip, coarse_frames, start)
    feat, _ = fine([clip, None])
    feat_masks = torch.ones(b, tf, device=clip.device)
    # filled ON the device (four fill kernels): a host list -> device copy is not permitted while the stream is being captured into a hipGraph
    meta = torch.empty(b, 4, dtype=torch.int64, device=clip.device)
    for col, val in enumerate((s, xc.shape[2], tf, 1)):
        meta[:, col] = val
    return coarse([xc, feat, feat_masks, 0, meta]), feat


def param_groups(fine, coarse, lr):
    """fusion parameters ('rw' / 'mix' in the name) at 10x, as in train_coarse_fineFEAT.py:137-141; the Fine stream and the
    coarse trunk at the base rate.  Parameters the joint graph never reaches (the tower's fc1 / fc2) are left out."""
    groups = tc.param_groups(coarse, lr)
    groups[0]['params'] = groups[0]['params'] + [p for n, p in fine.named_parameters() if not n.startswith(('fc1.', 'fc2.'))]
    return groups


def train_step(fine, coarse, reducer, optimizer, clip, labels, masks, pre_step=None):
    logits, _ = joint_forward(fine, coarse, clip)
    cls_loss, loc_loss, probs = tc.detection_loss(logits, labels, masks)
    scaler = train_fine.loss_scaler(fine, coarse)  # fp16 fine tower / fp16 coarse layer 1 (BASELINE configs[4]): device-side loss scale, see train_fine.LossScaler
    loss = (cls_loss + loc_loss) / 2
    reducer.begin_pass()
    (loss if scaler is None else scaler.scale_loss(loss)).backward()
    reducer.finish()
    train_fine.unscale_grads([p for g in optimizer.param_groups for p in g['params']], scaler)
    if pre_step is not None:
        pre_step()
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    return cls_loss.detach(), loc_loss.detach(), probs.detach()


def run(init_lr=INIT_LR, warmup_steps=0, max_steps=None, batch_size=BS, fine_frames=128, coarse_frames=64, dataloader=None,
        pretrained_fine=None, pretrained_coarse=None, save_model='models/joint_charades_', log=print, fine_act_dtype=None, coarse_act_dtype=None):
    rank, world, dev = cdist.init_from_env()
    local_bs = max(batch_size // world, 1)
    if dataloader is None:
        dataloader = SyntheticJoint(local_bs, tc.CHARADES_TR_SIZE // batch_size, fine_frames, coarse_frames, seed=rank)
    fine, coarse = build_models(dev, pretrained_fine, pretrained_coarse, fine_act_dtype=fine_act_dtype, coarse_act_dtype=coarse_act_dtype)
    cdist.sync_module(fine)
    cdist.sync_module(coarse)
    groups = param_groups(fine, coarse, init_lr)
    optimizer = optim.SGD(groups, lr=init_lr, momentum=0.9, weight_decay=1e-5)
    reducer = cdist.GradReducer([p for g in groups for p in g['params']])     # ONE reducer over both parameter sets
    fine.train(True)
    coarse.train(True)
    steps = 0
    # (the clip, labels and masks reach HBM one batch ahead of the step, on a copy stream: cfn_hip/staging.py)
    for clip, labels, masks in staging.stage(dataloader, dev):
        ok = clip.shape[0] == local_bs
        if not (cdist.all_agree(ok, dev) if world > 1 else ok):
            continue
        warm = (lambda: lr_warmup(init_lr, steps, warmup_steps, optimizer))
        cls_loss, loc_loss, _ = train_step(fine, coarse, reducer, optimizer, clip.to(dev), labels.to(dev), masks.to(dev), warm)
        steps += 1
        if steps % 50 == 0 or max_steps is not None:
            m_loc, m_cls = cdist.mean_over_ranks([float(loc_loss), float(cls_loss)], dev)
            if rank == 0:
                log(' joint steps: {} Loc Loss: {:.4f} Cls Loss: {:.4f}'.format(steps, m_loc, m_cls))
        if steps % 1000 == 0 and rank == 0:
            os.makedirs(os.path.dirname(save_model) or '.', exist_ok=True)
            torch.save({'fine_state_dict': fine.state_dict(), 'model_state_dict': coarse.state_dict(),
                        'optimizer_state_dict': optimizer.state_dict()}, save_model + str(steps).zfill(6) + '.pt')
        if max_steps is not None and steps >= max_steps:
            break
    return fine, coarse


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('-gpu', default='0', type=str)
    parser.add_argument('--max-steps', type=int, default=None)
    parser.add_argument('--batch-size', type=int, default=BS)
    args = parser.parse_args()
    if 'RANK' not in os.environ and len(args.gpu.split(',')) > 1:
        import subprocess
        env = dict(os.environ, CUDA_VISIBLE_DEVICES=args.gpu, HSA_ENABLE_IPC_MODE_LEGACY='0')
        n = len(args.gpu.split(','))
        sys.exit(subprocess.call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
                                  '--master-addr', '127.0.0.1', '--master-port', os.environ.get('MASTER_PORT', '29513'),
                                  os.path.abspath(__file__), '--batch-size', str(args.batch_size)] +
                                 (['--max-steps', str(args.max_steps)] if args.max_steps else []), env=env))
    if 'RANK' not in os.environ:
        os.environ['CUDA_VISIBLE_DEVICES'] = args.gpu
    run(batch_size=args.batch_size, max_steps=args.max_steps)
