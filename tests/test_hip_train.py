"""GPU: the drop-in entry points run end to end on the HIP path (synthetic Charades-shaped batches)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_train_fine_two_steps(tmp_path):
    import train_fine
    loaders = {'train': train_fine.SyntheticCharades(2, 3, frames=8, crop=64),
               'val': train_fine.SyntheticCharades(1, 1, frames=8, crop=64)}
    logs = []
    net = train_fine.run(batch_size=2, dataloaders=loaders, max_steps=2, pretrained=None, log=logs.append,
                         save_model=str(tmp_path / 'fine_'))
    assert all(torch.isfinite(p).all() for p in net.parameters())
    assert int(net.bn1.split_bn.num_batches_tracked) == 2


def test_train_coarse_step_and_val(tmp_path):
    import train_coarse_fineFEAT as tc
    loaders = {'train': tc.SyntheticCoarse(1, 1, frames=8, fine_len=12), 'val': tc.SyntheticCoarse(1, 1, frames=8, fine_len=12)}
    csvp = str(tmp_path / 'localize.csv')
    net = tc.run(batch_size=1, dataloaders=loaders, max_epochs=2, pretrained=None, csv_path=csvp, log=lambda *_: None,
                 save_model=str(tmp_path / 'coarse_'))
    assert all(torch.isfinite(p).all() for p in net.parameters())
    rows = open(csvp).read().strip().splitlines()
    assert len(rows) == 25 and len(rows[0].split(',')[2].split(' ')) == 157     # Charades_v1_localize row format


def test_extract_fine_features_roundtrip(tmp_path):
    import extract_fineFEAT as ex
    net = ex.build_tower(DEV, ckpt=None)
    g = torch.Generator().manual_seed(0)
    n = ex.extract(net, [('vidA', torch.randn(1, 3, 8, 224, 224, generator=g))], str(tmp_path))
    assert n == 1
    for k, c in (('layer1', 24), ('layer2', 48), ('layer3', 96), ('layer4', 192), ('conv5', 432)):
        f = torch.load(os.path.join(str(tmp_path), k, 'vidA'))
        assert f.shape == (1, c, 8, 7, 7) and f.dtype == torch.float32 and bool((f >= 0).all())


def test_smoke_entry():
    import __graft_entry__
    __graft_entry__.smoke()
