#!/bin/bash
# Copy the outputs of tools/refresh_profiles.sh (gpurun_out/refresh, merged back from the GPU box) into profiles/ as the
# round's tracked artefacts.  Usage: tools/collect_profiles.sh r02
set -e
TAG=${1:-r02}
R=gpurun_out/refresh
P=profiles
latest() { ls -t $1 | head -1; }
cp $(latest "$R/bench/runc/*kernel_stats.csv") $P/${TAG}_bench_kernel_stats.csv
cp $(latest "$R/bench/runc/*domain_stats.csv") $P/${TAG}_bench_domain_stats.csv
cp $(latest "$R/bench_bf16/runc/*kernel_stats.csv") $P/${TAG}_bench_bf16_kernel_stats.csv
for f in bench bench_bf16 bench_coarse bench_joint bench_joint_bf16tower; do tail -1 $R/$f.json > $P/${TAG}_$f.json; done
cp $R/microbench_b8.txt $P/${TAG}_microbench_b8.txt
cp $R/microbench_bf16_b8.txt $P/${TAG}_microbench_bf16_b8.txt
cp $R/stream_probe.txt $P/${TAG}_stream_probe.txt
cp $R/mfma_rate_probe.txt $P/${TAG}_issue_rate_probe.txt
F=$(latest "$R/pmc_fetch/runc/*counter_collection.csv"); W=$(latest "$R/pmc_write/runc/*counter_collection.csv")
python tools/pmc_traffic.py $F $W $P/${TAG}_pmc_dwfwd.json 8 256
cp $F $P/${TAG}_pmc_dwfwd_FETCH_SIZE.csv; cp $W $P/${TAG}_pmc_dwfwd_WRITE_SIZE.csv
python tools/pmc_sq.py $P/${TAG}_pmc_dw_valu.json $(latest "$R/pmc_sq1/runc/*counter_collection.csv") $(latest "$R/pmc_sq2/runc/*counter_collection.csv") --filter dw3d_,dwt5_ > /dev/null
ls -la $P | tail -30
