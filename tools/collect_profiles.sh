#!/bin/bash
# Copy the outputs of tools/refresh_profiles.sh (gpurun_out/refresh, merged back from the GPU box) into profiles/ as the
# round's tracked artefacts.  Usage: tools/collect_profiles.sh r02
set -e
TAG=${1:-r06}
R=gpurun_out/refresh
P=profiles
latest() { ls -t $1 | head -1; }
cp $(latest "$R/bench/runc/*kernel_stats.csv") $P/${TAG}_bench_kernel_stats.csv
cp $(latest "$R/bench/runc/*domain_stats.csv") $P/${TAG}_bench_domain_stats.csv
cp $(latest "$R/bench_bf16/runc/*kernel_stats.csv") $P/${TAG}_bench_bf16_kernel_stats.csv
for f in bench bench_rocprof bench_bf16 bench_fp16 bench_staged bench_coarse bench_coarse_eager bench_coarse_eager_rocprof bench_coarse_t256_eager bench_coarse_t256_eager_rocprof bench_coarse_bf16 bench_coarse_fp16 bench_coarse_eager_staged bench_joint bench_joint_bf16tower bench_joint_fp16tower; do tail -1 $R/$f.json > $P/${TAG}_$f.json; done    # bench = the default line (both rooflines + CPU leg), bench_rocprof = the fine step alone under rocprofv3 (matches ${TAG}_bench_kernel_stats.csv)
cp $(latest "$R/bench_coarse/runc/*kernel_stats.csv") $P/${TAG}_coarse_kernel_stats.csv          # (both coarse CSVs come from THIS refresh: they were stale in r02 / r03)
cp $(latest "$R/bench_coarse_t256/runc/*kernel_stats.csv") $P/${TAG}_coarse_t256_kernel_stats.csv
cp $R/membw.txt $P/${TAG}_membw.txt
cp $R/sal_bench.txt $P/${TAG}_sal_bench.txt
cp $R/salb_bench.txt $P/${TAG}_salb_bench.txt
for f in pwfs_bench pwfs_knockouts pwfs_stress pwfs_step_ab; do cp $R/$f.txt $P/${TAG}_$f.txt; done
cp $R/glue_coarse.txt $P/${TAG}_glue_coarse.txt
cp $R/sync_debug.txt $P/${TAG}_sync_debug.txt
cp $R/determinism_scan.txt $P/${TAG}_determinism_scan.txt
cp $R/power_clock.txt $P/${TAG}_power_clock.txt
cp $R/microbench_b8.txt $P/${TAG}_microbench_b8.txt
cp $R/microbench_bf16_b8.txt $P/${TAG}_microbench_bf16_b8.txt
cp $R/stream_probe.txt $P/${TAG}_stream_probe.txt
cp $R/mfma_rate_probe.txt $P/${TAG}_issue_rate_probe.txt
F=$(latest "$R/pmc_fetch/runc/*counter_collection.csv"); W=$(latest "$R/pmc_write/runc/*counter_collection.csv")
python tools/pmc_traffic.py $F $W $P/${TAG}_pmc_dwfwd.json 8 256
cp $F $P/${TAG}_pmc_dwfwd_FETCH_SIZE.csv; cp $W $P/${TAG}_pmc_dwfwd_WRITE_SIZE.csv
python tools/pmc_sq.py $P/${TAG}_pmc_dw_valu.json $(latest "$R/pmc_sq1/runc/*counter_collection.csv") $(latest "$R/pmc_sq2/runc/*counter_collection.csv") --filter dw3d_,dwt5_ > /dev/null
for v in fp32mfma split6 bf16; do
  python tools/pmc_sq.py $P/${TAG}_pmc_pw_${v}.json $(latest "$R/pmc_pw1_$v/runc/*counter_collection.csv") $(latest "$R/pmc_pw2_$v/runc/*counter_collection.csv") --filter pw_,pws_,pwb_,pwk_ > /dev/null
done
python - <<PY
import json
out = {'note': 'MFMA-busy evidence for the pointwise kernels of X3D-M layers 2-4 (8 clips x T=256, tools/pw_only.py under rocprofv3 --pmc, two passes per variant; every shape forced onto the named arithmetic). mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel wall time x 2.1 GHz assumed).', 'variants': {}}
for v in ('fp32mfma', 'split6', 'bf16'):
    d = json.load(open('$P/${TAG}_pmc_pw_%s.json' % v))['kernels']
    out['variants'][v] = {k: {'avg_us': x.get('avg_us'), 'mfma_busy_per_simd': x.get('per_simd_at_2.1GHz', {}).get('mfma_busy'), 'valu_busy_per_simd': x.get('per_simd_at_2.1GHz', {}).get('valu_busy'), 'SQ_VALU_MFMA_BUSY_CYCLES': x['counters'].get('SQ_VALU_MFMA_BUSY_CYCLES'), 'SQ_BUSY_CYCLES': x['counters'].get('SQ_BUSY_CYCLES'), 'SQ_INSTS_VALU_MFMA_MOPS_BF16': x['counters'].get('SQ_INSTS_VALU_MFMA_MOPS_BF16'), 'lds_conflict_over_active_lds': round(x['counters'].get('SQ_LDS_BANK_CONFLICT', 0) / max(x['counters'].get('SQ_ACTIVE_INST_LDS', 1), 1), 3)} for k, x in d.items()}
json.dump(out, open('$P/${TAG}_pmc_pw_mfma.json', 'w'), indent=1)
PY
tail -1 $R/bench_coarse_t256.json > $P/${TAG}_bench_coarse_t256.json
ls -la $P | tail -30
