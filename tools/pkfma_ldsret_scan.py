#!/usr/bin/env python3
"""Scan every kernel's gfx950 ISA for the instruction pattern of DESIGN 4.1 (the round-3 weight-gradient race).

What 4.1 pinned down at the ISA level: in a kernel whose SIMD partner wave is MFMA bound, the LOW half of a `v_pk_fma_f32` read the
`op_sel`-ed HIGH register of a register pair that a wide LDS read (`ds_read_b64` / `b96` / `b128`) had JUST returned, as zero, in lanes 48-63 --
with `s_waitcnt lgkmcnt(0)` between the two.  The mechanism below the ISA is not known, so the product must not rely on the pattern being
safe: this tool lists every place where it occurs.

Pattern (per kernel that issues at least one `v_mfma_*`; straight-line scan in layout order, state kept across labels and conditional branches
-- a superset of the hits along real paths -- and reset behind unconditional branches):

    ds_read_{b64,b96,b128} / ds_read2{,st64}_b32 / ds_read2{,st64}_b64   vD[lo:hi], ...       (an LDS read that returns >= 2 registers)
    ... (anything that does not overwrite vD)
    s_waitcnt ... lgkmcnt(N)                                                                   (the read is complete afterwards)
    ... (anything that does not READ or overwrite the registers)
    v_pk_{fma,mul,add}_f32  ..., v[a:a+1], ...  op_sel:[..1..]      <- FIRST VALU reader of a+1, which is one of the returned registers, and
                                                                       op_sel[i] = 1: the LOW half takes the HIGH register of the pair

class S (strict, the 4.1 signature): as above.   class W (wide): the first VALU reader of a just-returned register is ANY packed fp32
instruction that takes it as the high register of a source pair (op_sel_hi[i] = 1, the default) -- reported as counts per kernel only.
`gap` = VALU / other instructions issued between the s_waitcnt and the reader (4.1's two sites had gap 0 and gap 3).  Every class-S hit is
tagged with where it sits: `mfma-loop` (inside a backward-branch span that also holds MFMAs: the reader can run while the SIMD's other wave is
matrix bound -- the condition 4.1 needed), `loop` (a loop without MFMAs) or `straight` (kernel prologue / epilogue: every wave of the workgroup
is in the same phase there).

    python tools/pkfma_ldsret_scan.py [--out profiles/r06_pkfma_scan.txt] [--jobs 8] [file.s | file.hip ...]

Without file arguments every csrc/*.hip is compiled to assembly (`hipcc -S --cuda-device-only`, the product's flags) into a temp directory.
"""
import argparse
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'coarse-fine-networks_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wno-unused-value', '-Wno-unused-result']

WIDE_LDS = re.compile(r'^ds_read(_b64|_b96|_b128|2_b32|2st64_b32|2_b64|2st64_b64|_b64_tr_b16|_b64_tr_b8|_b96_tr_b6)\b')
PK = re.compile(r'^v_pk_(fma|mul|add)_f32\b')


def vregs(tok):
    """the VGPR numbers an operand token names ('v12', 'v[4:7]'); empty for anything else"""
    tok = tok.strip().rstrip(',')
    m = re.match(r'^v\[(\d+):(\d+)\]$', tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'^v(\d+)$', tok)
    return [int(m.group(1))] if m else []


def split_operands(rest):
    """'v[0:1], v[2:3], s[4:5] op_sel:[0,1] op_sel_hi:[1,0]' -> (['v[0:1]', 'v[2:3]', 's[4:5]'], {'op_sel': [0, 1], 'op_sel_hi': [1, 0]})"""
    mods = {}
    for name in ('op_sel_hi', 'op_sel', 'neg_lo', 'neg_hi'):
        m = re.search(r'\b%s:\[([0-9,\s]*)\]' % name, rest)
        if m:
            mods[name] = [int(v) for v in m.group(1).replace(' ', '').split(',') if v != '']
            rest = rest[:m.start()] + rest[m.end():]
    rest = re.split(r'\s+(?:offset\d*:|gds|clamp|mul:|div:|dpp|row_|quad_|bank_|bound_|sdwa|dst_sel|src\d_sel|cbsz|abid|blgp|offen|idxen|glc|slc|nt|sc\d|format:)', rest)[0]
    ops, depth, cur = [], 0, ''
    for ch in rest:
        if ch == '[':
            depth += 1
        elif ch == ']':
            depth -= 1
        if ch == ',' and depth == 0:
            ops.append(cur.strip())
            cur = ''
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops, mods


def kernels(path):
    """-> [(mangled name, [instruction text, ...] with labels kept as 'LABEL:')]"""
    out, name, code = [], None, None
    for line in open(path, errors='replace'):
        m = re.match(r'^(_Z[\w.$]+):', line)
        if m:
            name, code = m.group(1), []
            out.append((name, code))
            continue
        if code is None:
            continue
        t = line.split(';')[0].strip()
        if not t:
            continue
        if t.startswith('.'):
            if t.startswith('.Lfunc_end') or t.startswith('.section') or t.startswith('.rodata') or t.startswith('.amdhsa_kernel'):
                code = None if not t.startswith('.L') else code
            m2 = re.match(r'^(\.L\w+):', t)
            if m2 and code is not None:
                code.append('LABEL:' + m2.group(1))
            continue
        if t.endswith(':'):
            code.append('LABEL:' + t[:-1])
            continue
        code.append(t)
    return out


def scan_kernel(code):
    """-> (strict hits [(ds_read text, waitcnt text, reader text, gap)], wide count, n_mfma, n_wide_lds, n_pk)"""
    n_mfma = sum(1 for t in code if t.startswith('v_mfma_') or t.startswith('v_smfmac_'))
    n_lds = sum(1 for t in code if WIDE_LDS.match(t))
    n_pk = sum(1 for t in code if PK.match(t))
    if n_mfma == 0 or n_lds == 0 or n_pk == 0:
        return [], 0, n_mfma, n_lds, n_pk
    # loops: a backward branch to a label spans [label, branch]; a hit is "in an MFMA loop" when such a span holds it AND an MFMA
    label_at = {t[6:]: i for i, t in enumerate(code) if t.startswith('LABEL:')}
    mfma_at = [i for i, t in enumerate(code) if t.startswith('v_mfma_') or t.startswith('v_smfmac_')]
    loops = []
    for i, t in enumerate(code):
        if t.startswith('s_cbranch') or t.startswith('s_branch'):
            tgt = t.split()[-1]
            j = label_at.get(tgt)
            if j is not None and j <= i:
                loops.append((j, i, any(j <= m <= i for m in mfma_at)))

    def where(i):
        inl = [lp for lp in loops if lp[0] <= i <= lp[1]]
        if not inl:
            return 'straight'
        return 'mfma-loop' if any(lp[2] for lp in inl) else 'loop'
    pending = {}      # reg -> ds_read text (issued, not yet waited for)
    fresh = {}        # reg -> (ds_read text, waitcnt text, gap counter list) returned by the last lgkmcnt wait and not read by a VALU op since
    strict, wide = [], 0
    for idx, t in enumerate(code):
        if t.startswith('LABEL:') or t.startswith('s_cbranch'):
            continue                  # fall-through continues in a straight line: keep the state (a superset of the true hits)
        if (t.startswith('s_branch') and label_at.get(t.split()[-1], len(code)) > idx) or t.startswith('s_setpc') or t.startswith('s_endpgm'):
            pending, fresh = {}, {}   # the next instruction is not reached from here
            continue
        parts = t.split(None, 1)
        op, rest = parts[0], (parts[1] if len(parts) > 1 else '')
        if op == 's_waitcnt':
            m = re.search(r'lgkmcnt\((\d+)\)', rest)
            # LDS reads of one wave return in order: lgkmcnt(N) completes all but the last N; be conservative -- any lgkmcnt wait "returns" every
            # pending read (a too-early classification only widens the window the scan looks at)
            if m is not None or rest.strip() in ('0', ''):
                for r, d in pending.items():
                    fresh[r] = [d, t, 0]
                pending = {}
            continue
        ops, mods = split_operands(rest)
        if WIDE_LDS.match(op):
            dst = vregs(ops[0]) if ops else []
            for r in dst:
                fresh.pop(r, None)
                pending[r] = t
            continue
        is_valu = op.startswith('v_')
        dst, srcs = [], []
        if is_valu and ops:
            first_src = 1
            dst = vregs(ops[0])
            if op.startswith('v_cmp') or op.startswith('v_cmpx'):
                dst = []
                first_src = 1 if not ops[0].startswith('v') else 0
            # carry-out forms (v_add_co_u32 v, vcc, a, b) and readlane-style ops keep extra non-VGPR operands: harmless for the register sets
            srcs = ops[first_src:]
        elif ops:
            # stores / buffer ops / ds_write read VGPRs but are not the VALU consumer 4.1 is about; an overwrite by a load kills freshness
            if op.startswith(('buffer_load', 'global_load', 'flat_load', 'scratch_load', 'ds_read', 'ds_bpermute', 'ds_permute', 'ds_swizzle')):
                dst = vregs(ops[0])
        if is_valu and fresh:
            if PK.match(op):
                sel = mods.get('op_sel', [0] * len(srcs))
                sel_hi = mods.get('op_sel_hi', [1] * len(srcs))
                for i, s in enumerate(srcs):
                    rs = vregs(s)
                    if len(rs) != 2:
                        continue
                    hi = rs[1]
                    if hi in fresh:
                        d, w, gap = fresh[hi]
                        if i < len(sel) and sel[i] == 1:
                            strict.append((d, w, t, gap, where(idx)))
                        elif i < len(sel_hi) and sel_hi[i] == 1:
                            wide += 1
            read = set()
            if PK.match(op):
                # a packed fp32 source pair is read per HALF: the low half takes register op_sel[i] of the pair, the high half register op_sel_hi[i]
                # (default 1).  `v_pk_fma_f32 .., v[2:3], .. op_sel_hi:[..0..]` reads v2 twice and v3 NOT AT ALL -- counting the whole pair as read
                # hid the sites of csrc/pwfuseds.hip's first build (a third packed FMA took v3 through op_sel, after two that had only read v2)
                sel = mods.get('op_sel', [0] * len(srcs))
                sel_hi = mods.get('op_sel_hi', [1] * len(srcs))
                for i, s in enumerate(srcs):
                    rs = vregs(s)
                    if len(rs) == 2:
                        read.add(rs[sel[i] if i < len(sel) else 0])
                        read.add(rs[sel_hi[i] if i < len(sel_hi) else 1])
                    else:
                        read.update(rs)
            else:
                for s in srcs:
                    read.update(vregs(s))
            for r in list(fresh):
                if r in read:
                    del fresh[r]
        for r in dst:
            fresh.pop(r, None)
            pending.pop(r, None)
        for r in fresh:
            fresh[r][2] += 1
    return strict, wide, n_mfma, n_lds, n_pk


def demangle(names):
    try:
        p = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), capture_output=True, text=True, check=True)
        return dict(zip(names, p.stdout.split('\n')))
    except Exception:
        return {n: n for n in names}


def compile_to_asm(src, outdir):
    out = os.path.join(outdir, os.path.basename(src)[:-4] + '.s')
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h') or f.endswith('.hip')]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    extra = []
    with open(src) as fh:
        for line in fh:
            if line.startswith('// hipcc-flags:'):
                extra += line.split(':', 1)[1].split()
    subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + extra + ['-S', '--cuda-device-only', '-I' + CSRC, src, '-o', out], check=True, cwd=CSRC)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('files', nargs='*')
    ap.add_argument('--out', default=None)
    ap.add_argument('--jobs', type=int, default=8)
    ap.add_argument('--asm-dir', default=None, help='keep / reuse the generated .s files here (reused when newer than their source and every header)')
    args = ap.parse_args()
    srcs = [os.path.abspath(f) for f in args.files] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    tmp = args.asm_dir or tempfile.mkdtemp(prefix='pkscan_')
    os.makedirs(tmp, exist_ok=True)
    asm = [s for s in srcs if s.endswith('.s')]
    hip = [s for s in srcs if s.endswith('.hip')]
    with concurrent.futures.ThreadPoolExecutor(max_workers=args.jobs) as ex:
        asm += list(ex.map(lambda s: compile_to_asm(s, tmp), hip))
    lines, fam = [], {}
    tot_k = tot_mfma_k = tot_s = tot_w = tot_sl = 0
    for path in sorted(asm):
        ks = kernels(path)
        names = demangle([n for n, _ in ks])
        for name, code in ks:
            tot_k += 1
            strict, wide, n_mfma, n_lds, n_pk = scan_kernel(code)
            if n_mfma:
                tot_mfma_k += 1
            if not strict and not wide:
                continue
            tot_s += len(strict)
            tot_sl += sum(1 for h in strict if h[4] == 'mfma-loop')
            tot_w += wide
            m = re.match(r'_Z(\d+)', name)
            base = name[2 + len(m.group(1)):][:int(m.group(1))] if m else name
            f = fam.setdefault((os.path.basename(path), base), [0, 0, 0, 0])
            f[0] += 1
            f[1] += 1 if strict else 0
            f[2] += len(strict)
            f[3] += wide
            if strict:      # class-S sites are listed one by one; class-W readers are counted per family below
                lines.append('%s: %s\n    mfma %d, wide LDS reads %d, packed fp32 %d | class S: %d | class W: %d'
                             % (os.path.basename(path), names[name][:160], n_mfma, n_lds, n_pk, len(strict), wide))
                for d, w, t, gap, wh in strict:
                    lines.append('      S %-9s gap %-3d %s  ->  %s  ->  %s' % (wh, gap, d, w, t))
    head = ['pkfma_ldsret_scan: %d kernels in %d files, %d of them issue MFMAs; class S hits: %d (%d of them inside a loop that also issues MFMAs), class W hits: %d'
            % (tot_k, len(asm), tot_mfma_k, tot_s, tot_sl, tot_w), '',
            '# class S = the round-3 signature (DESIGN 4.1): the FIRST reader of a register pair that a wide LDS read just returned is a v_pk_{fma,mul,add}_f32 whose op_sel takes',
            '#           the pair\'s HIGH register into the instruction\'s LOW half (sites listed one by one at the end).',
            '# class W = a packed fp32 instruction is the first reader and takes the returned register as the high half in the ordinary way (op_sel_hi, the default) -- not the',
            '#           failing half in round 3; counted per kernel family.', '']
    table = ['%-14s %-28s instantiations with packed first readers %4d (with class S: %3d)   class S %4d   class W %6d' % (f, k, v[0], v[1], v[2], v[3])
             for (f, k), v in sorted(fam.items())]
    text = '\n'.join(head + table + [''] + lines) + '\n'
    if args.out:
        with open(args.out, 'w') as fh:
            fh.write(text)
    sys.stdout.write(text if len(text) < 20000 else text[:20000] + '\n... (see --out)\n')


if __name__ == '__main__':
    main()
