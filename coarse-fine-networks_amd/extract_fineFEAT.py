"""extract_fineFEAT -- runs the trained Fine stream as a feature tower over whole videos and stores the five
multi-level feature maps per video, in the reference's on-disk format (extract_fineFEAT.py:153-173; read back by
charades_coarse_fineFEAT.py:84-87):

    <save_dir>/<key>/<vid>     one ``torch.save``d fp32 CPU tensor of shape (1, C_key, T', 7, 7)
    keys = layer1 (24), layer2 (48), layer3 (96), layer4 (192), conv5 (432)

``x3d_fine.generate_model(..., global_tower=True)`` produces them with the HIP path (adaptive (None,7,7)
average pooling of every stage output, x3d_fine.py:339-363).  ``extract(videos)`` takes any iterable of
(vid, clip (1,3,T,224,224)); the Charades frame reader itself is out of scope."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import x3d_fine                                   # noqa: E402

FEAT_KEYS = ('layer1', 'layer2', 'layer3', 'layer4', 'conv5')


def build_tower(device, ckpt=None, n_classes=157):
    net = x3d_fine.generate_model('M', n_classes=n_classes, n_input_channels=3, task='loc', dropout=0.5,
                                  base_bn_splits=1, global_tower=True)
    if ckpt and os.path.exists(ckpt):
        net.load_state_dict(torch.load(ckpt, map_location='cpu')['model_state_dict'])
    net.to(device).train(False)
    net.aggregate_sub_bn_stats()          # extract_fineFEAT.py:136-139
    return net


@torch.no_grad()
def extract(net, videos, save_dir, device='cuda'):
    for k in FEAT_KEYS:
        os.makedirs(os.path.join(save_dir, k), exist_ok=True)
    n = 0
    for vid, clip in videos:
        feat, _ = net([clip.to(device), None])
        for k in FEAT_KEYS:
            torch.save(feat[k].data.cpu(), os.path.join(save_dir, k, vid))
        n += 1
    return n


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('-gpu', default='0')
    ap.add_argument('--save-dir', default='fine_feat')
    ap.add_argument('--ckpt', default='models/fine_charades_039000_SAVE.pt')
    ap.add_argument('--synthetic', type=int, default=2, help='number of synthetic videos to run')
    a = ap.parse_args()
    os.environ.setdefault('CUDA_VISIBLE_DEVICES', a.gpu)
    g = torch.Generator().manual_seed(0)
    vids = (('synthetic_%03d' % i, torch.randn(1, 3, 64, 224, 224, generator=g)) for i in range(a.synthetic))
    print('wrote', extract(build_tower('cuda', a.ckpt), vids, a.save_dir), 'videos to', a.save_dir)
