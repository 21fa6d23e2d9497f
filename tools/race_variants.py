#!/usr/bin/env python3
"""ISA-level bisection of the round-3 weight-gradient race: takes the device assembly of the ROUND-3
pwsplitw.hip (hipcc -S --cuda-device-only, see tools/asm_variant.sh) and writes variants in which ONE kind of instruction is
inserted at fixed sites of pws_wgrad_staged_kernel<2,2,2,true,3,4> (layer-2 conv3 weight gradient, the failing launch);
every other byte of the library stays what round 3 shipped.

Sites (source: csrc/pwsplitw.hip main loop `convert / barrier / issue / mfma`):
  A  entry of convert(h, buf0, set0)      (both the loop-entry copy and the back-edge copy)
  B  entry of convert(h+1, buf1, set1)
  C  right behind the six buffer loads of issue(h+2, set0)
  D  right behind the six buffer loads of issue(h+3, set1)
"""
import re
import sys

KERNEL = '_Z23pws_wgrad_staged_kernelILi2ELi2ELi2ELb1ELi3ELi4EEv7WssArgs'

VARIANTS = {
    'same': {},
    'vm0': {'A': ['s_waitcnt vmcnt(0)'], 'B': ['s_waitcnt vmcnt(0)']},            # H1: operand registers read before the loads landed
    'nop': {'A': ['s_nop 0'], 'B': ['s_nop 0']},                                  # control for vm0 / lgkm0: same size, same place
    'lgkm0': {'A': ['s_waitcnt lgkmcnt(0)'], 'B': ['s_waitcnt lgkmcnt(0)']},
    'bar': {'A': ['s_waitcnt lgkmcnt(0)', 's_barrier'], 'B': ['s_waitcnt lgkmcnt(0)', 's_barrier']},   # no wave runs ahead of the multiplying waves
    'pad': {'C': ['s_nop 7', 's_nop 7'], 'D': ['s_nop 7', 's_nop 7']},            # H3: operands of the loads read after issue
    'nop2': {'C': ['s_nop 0', 's_nop 0'], 'D': ['s_nop 0', 's_nop 0']},           # control for pad
}


def main(src, outdir):
    lines = open(src).read().split('\n')
    beg = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ':'))
    end = next(i for i in range(beg, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    sites = {'A': [], 'B': [], 'C': [], 'D': []}
    loads = [i for i in range(beg, end) if 'buffer_load_dwordx4' in lines[i]]
    # the two in-loop issue groups: six loads each behind an s_barrier
    bars = [i for i in range(beg, end) if lines[i].strip() == 's_barrier']
    loop_bars = bars[-2:]                       # (the first s_barrier belongs to the coefficient tables)
    for name, b in zip('CD', loop_bars):
        grp = [i for i in loads if b < i < b + 20]
        assert len(grp) == 6, (name, grp)
        sites[name].append(grp[-1] + 1)         # insert BEHIND the last load
    for i in range(beg, end):
        s = lines[i].strip()
        if s == 's_and_saveexec_b64 s[8:9], s[0:1]':
            sites['A'].append(i)                # insert in front
        if s == 's_and_saveexec_b64 s[18:19], s[0:1]':
            sites['B'].append(i)
    assert len(sites['A']) == 2 and len(sites['B']) == 1, sites
    for name, edits in VARIANTS.items():
        out = list(lines)
        ins = []
        for site, instrs in edits.items():
            for at in sites[site]:
                ins.append((at, instrs))
        for at, instrs in sorted(ins, reverse=True):
            out[at:at] = ['\t' + x + '\t; race-variant %s' % name for x in instrs]
        with open('%s/pwsplitw_r3_%s.s' % (outdir, name), 'w') as fh:
            fh.write('\n'.join(out))
        print(name, {k: [v - beg for v in sites[k]] for k in edits})


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
