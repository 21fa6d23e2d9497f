#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ counter passes (counter_collection.csv files) per kernel variant:

    python tools/pmc_sq.py <out.json> <pass1_counter_collection.csv> [<pass2...>] [--filter dw3d_,dwt5_]

Per kernel (first launch of each (kernel, grid) dropped as warm-up): mean of every counter per launch, plus derived
fractions of SQ_WAVE_CYCLES: issue-active (ACTIVE_INST_ANY), VALU-active, parked (WAIT_ANY: s_waitcnt / barrier),
issue-stall (WAIT_INST_ANY).  SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md)."""
import collections, csv, json, re, sys


def main():
    out = sys.argv[1]
    flt = None
    files = []
    it = iter(sys.argv[2:])
    for a in it:
        if a == '--filter':
            flt = next(it).split(',')
        else:
            files.append(a)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        seen = set()
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id']))
        for r in rows:
            name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
            if flt and not any(x in name for x in flt):
                continue
            key = (name, r['Grid_Size'])
            tag = (key, r['Counter_Name'])
            if tag not in seen:          # warm-up launch of this shape
                seen.add(tag)
                continue
            acc['%s grid %s' % key][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] == 'SQ_WAVE_CYCLES' and 'Start_Timestamp' in r:
                acc['%s grid %s' % key]['_ns'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    res = {}
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        ns = m.pop('_ns', None)
        cs = {c: v for c, v in cs.items() if c != '_ns'}
        wc = m.get('SQ_WAVE_CYCLES')
        d = {'launches': max(len(v) for v in cs.values()), 'counters': {c: round(v, 1) for c, v in m.items()}}
        if wc:
            for lab, c in (('active_inst_any', 'SQ_ACTIVE_INST_ANY'), ('active_inst_valu', 'SQ_ACTIVE_INST_VALU'), ('wait_any_parked', 'SQ_WAIT_ANY'),
                           ('wait_inst_any_issue_stall', 'SQ_WAIT_INST_ANY'), ('active_inst_lds', 'SQ_ACTIVE_INST_LDS'),
                           ('active_inst_vmem', 'SQ_ACTIVE_INST_VMEM'), ('wait_inst_lds', 'SQ_WAIT_INST_LDS')):
                if c in m:
                    d['frac_of_wave_cycles_' + lab] = round(m[c] / wc, 4)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'SQ_BUSY_CYCLES' in m and m['SQ_BUSY_CYCLES']:
            # MFMA-busy cycles (per SIMD, summed) over the cycles the SQs were busy: SQ_BUSY_CYCLES counts per SE-level SQ, so the
            # ratio is reported next to the per-SIMD figure derived from the wall time below
            d['mfma_busy_over_sq_busy'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CYCLES'], 4)
        if 'SQ_INSTS_VALU' in m and 'SQ_WAVES' in m and m['SQ_WAVES']:
            d['valu_insts_per_wave'] = round(m['SQ_INSTS_VALU'] / m['SQ_WAVES'], 1)
        if ns and wc:      # per-SIMD view at an ASSUMED 2.1 GHz (1024 SIMDs; SQ cycle counters are in quad-cycles)
            simd_cycles = 1024.0 * ns * 2.1
            d['avg_us'] = round(ns / 1e3, 1)
            d['per_simd_at_2.1GHz'] = {'valu_busy': round(4 * m.get('SQ_ACTIVE_INST_VALU', 0) / simd_cycles, 3),
                                       'inst_issue_busy': round(4 * m.get('SQ_ACTIVE_INST_ANY', 0) / simd_cycles, 3),
                                       'resident_waves': round(4 * wc / simd_cycles, 2)}
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in m:      # counts cycles, not quad-cycles (MI355X_MICROARCH.md)
                d['per_simd_at_2.1GHz']['mfma_busy'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles, 3)
        res[k] = d
    json.dump({'source': 'rocprofv3 --kernel-trace --pmc <SQ counters> (separate passes) -- python tools/dwfwd_only.py | tools/pw_only.py', 'kernels': res},
              open(out, 'w'), indent=1)
    for k, d in res.items():
        print(k, {x: y for x, y in d.items() if x.startswith('frac') or x.startswith('valu')})


if __name__ == '__main__':
    main()
