#!/usr/bin/env python3
"""Sweep of the CPU baseline's thread count: bench.py's `cpu_baseline` leg (the oracle port, one 3 x 256 x 224 x 224 clip
fwd+bwd) at CFN_CPU_THREADS = 8 / 16 / 32 / 64 / 128 on the GPU box's host; one warm-up + `--repeats` timed runs each.

    python tools/cpu_threads.py [--threads 8,16,32,64,128] [--repeats 1] [--frames 256] > profiles/r06_cpu_threads.txt
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', default='8,16,32,64,128')
    ap.add_argument('--repeats', type=int, default=1)
    ap.add_argument('--frames', type=int, default=256)
    args = ap.parse_args()
    print('# cpu_baseline (oracle port, 1 clip 3x%dx224x224 fwd+bwd fp32) against the thread count; host: %s, %d logical CPUs' % (args.frames, bench._cpu_model(), os.cpu_count()))
    best = None
    for th in [int(v) for v in args.threads.split(',')]:
        if th > (os.cpu_count() or 1):
            continue
        os.environ['CFN_CPU_THREADS'] = str(th)
        r = bench.cpu_baseline(args.frames, repeats=args.repeats)
        print('threads %4d   %.5f clips/s   (%s)' % (th, r['value'], r['sample']), flush=True)
        if best is None or r['value'] > best[1]:
            best = (th, r['value'])
    print('# best: %d threads, %.5f clips/s' % best)
