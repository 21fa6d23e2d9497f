#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of
tools/dwfwd_only.py into profiles/rNN_pmc_dwfwd.json.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [B] [T]

Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are KiB; on gfx950
FETCH_SIZE counts wide coalesced reads at half weight, so it is doubled; WRITE_SIZE is used as is.  The first launch of
every distinct (kernel, grid) is the warm-up and is dropped."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    """[(kernel, grid, [bytes per launch, ...]), ...] in dispatch order; one entry per layer shape: a run of launches of
    one (kernel, grid) is split where the counter jumps by > 25 % (two shapes can share kernel variant and grid)"""
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    out = []
    for r in rows:
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        if not (name.startswith('dw3d_kernel<0') or name.startswith('dwt5_kernel<0') or name.startswith('dw3d_small_fwd_kernel')
                or name.startswith('dw3d_cp_fwd_kernel') or name.startswith('dwt5_fwd_stream_kernel') or name.startswith('dw3d_flat')
                or name.startswith('dwt5_fwd_flat_kernel')):
            continue
        v = float(r['Counter_Value']) * 1024.0
        if out and out[-1][0] == name and out[-1][1] == r['Grid_Size'] and abs(v - out[-1][2][0]) <= 0.25 * out[-1][2][0]:
            out[-1][2].append(v)
        else:
            out.append((name, r['Grid_Size'], [v]))
    return out


def main():
    fetch, write, dst = sys.argv[1:4]
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    T = int(sys.argv[5]) if len(sys.argv) > 5 else 256
    f, w = per_kernel(fetch, 'FETCH_SIZE'), per_kernel(write, 'WRITE_SIZE')
    layers, tot, launches = [], 0.0, 0
    assert len(f) == len(w) and all(a[:2] == b[:2] and len(a[2]) == len(b[2]) for a, b in zip(f, w)), 'passes disagree'
    for (kn, grid, fa), (_, _, wa) in zip(f, w):
        key = (kn, grid)
        fv, wv = fa[1:], wa[1:]                  # drop the warm-up launch
        fb, wb = 2.0 * sum(fv) / len(fv), sum(wv) / len(wv)
        layers.append({'kernel': key[0], 'grid': int(key[1]), 'launches_per_step': len(fv),
                       'fetch_bytes_x2_per_launch': fb, 'write_bytes_per_launch': wb})
        tot += (fb + wb) * len(fv)
        launches += len(fv)
    doc = {
        'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python tools/dwfwd_only.py, '
                  'MI355X, T=%d, B=%d' % (T, B),
        'correction': 'counter values are KiB; FETCH_SIZE doubled (gfx950 counts wide coalesced reads at 1/2, '
                      'MI355X_MICROARCH.md HBM section); WRITE_SIZE as is',
        'kernels': 'dwt5_fwd_flat_kernel (conv1_t) + dw3d_flat*_fwd_kernel / dw3d_small_fwd_kernel (26 conv2 launches) per x3d_fine forward',
        'batch': B, 'frames': T, 'layers': layers, 'launches_per_step': launches,
        'traffic_bytes_per_step': tot, 'traffic_bytes_per_launch': tot / max(launches, 1),
    }
    json.dump(doc, open(dst, 'w'), indent=1)
    print(json.dumps({k: doc[k] for k in ('launches_per_step', 'traffic_bytes_per_step', 'traffic_bytes_per_launch')}))


if __name__ == '__main__':
    main()
