"""List the kernels whose gfx950 ISA contains "waterfall" loops: a buffer / global access whose descriptor or scalar offset
the compiler could not prove wave uniform is wrapped in  readfirstlane / v_cmp_eq / s_and_saveexec / <access> / s_xor exec /
s_cbranch_execnz  (about ten extra instructions and a taken branch per access; see cfn_common.h: cfn_uni).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icsrc -I../include -S --cuda-device-only -o k.s csrc/<file>.hip
    python tools/waterfall_scan.py k.s [...]
"""
import re
import subprocess
import sys


def scan(path):
    lines = open(path).read().split('\n')
    name, counts, labels = None, {}, {}
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name = m.group(1)
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
        m = re.match(r'\s+s_cbranch_execnz (\.LBB\d+_\d+)', l)
        if m and name and m.group(1) in labels and 0 < i - labels[m.group(1)] <= 24:
            body = '\n'.join(lines[labels[m.group(1)]:i])
            if 'v_readfirstlane' in body and 'saveexec' in body and re.search(r'(buffer|global|flat)_(load|store|atomic)', body):
                counts[name] = counts.get(name, 0) + 1
    return counts


if __name__ == '__main__':
    total = 0
    for path in sys.argv[1:]:
        for k, v in scan(path).items():
            d = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
            print('%-20s %4d  %s' % (path.split('/')[-1], v, d[:120]))
            total += v
    print('waterfall loops:', total)
