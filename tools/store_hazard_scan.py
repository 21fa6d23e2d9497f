"""List wide buffer stores whose data registers are overwritten right behind them.

Measured on gfx950 (hipcc 7.2, csrc/dwt5.hip): `buffer_store_dwordx4 v[a:a+3], v, s[..], sN offen` followed IMMEDIATELY by a VALU
write to one of v[a:a+3] stored the new value in some lanes.  LLVM's hazard recognizer inserts wait states for this pair only when
soffset is not a register (cfn_common.h: cfn_bst128 keeps the data registers alive for two wait states).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icsrc -I../include -S --cuda-device-only -o k.s csrc/<file>.hip
    python tools/store_hazard_scan.py k.s [...]
"""
import re
import sys


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    lines = [l for l in open(path).read().split('\n')]
    name, hits = None, []
    code = []
    for l in lines:
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name = m.group(1)
        t = l.strip()
        if t and not t.startswith(';') and not t.startswith('.') and not t.endswith(':'):
            code.append((name, t))
    for i, (nm, t) in enumerate(code):
        m = re.match(r'buffer_store_dwordx[34] (\S+), (\S+), (\S+), (\S+)', t)
        if not m or not re.match(r's\d+', m.group(4)):
            continue
        data = regs(m.group(1).rstrip(','))
        for k in (1, 2):                                   # the next two issue slots
            if i + k >= len(code) or code[i + k][0] != nm:
                break
            nt = code[i + k][1]
            if nt.startswith('s_nop') or nt.startswith('s_waitcnt'):
                break
            if nt.startswith('v_'):
                dst = regs(nt.split()[1].rstrip(','))
                if dst & data:
                    hits.append((nm, t, nt))
                    break
    return hits


if __name__ == '__main__':
    total = 0
    for p in sys.argv[1:]:
        for nm, st, nx in scan(p):
            print('%s: %s\n    %s\n    %s' % (p.split('/')[-1], nm[:70], st, nx))
            total += 1
    print('unprotected wide-store / overwrite pairs:', total)
