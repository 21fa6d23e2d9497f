#!/bin/bash
# Regenerate the measurements kept under profiles/ (run on the GPU box through gpurun; writes into gpurun_out/refresh).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/tools/dwfwd_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/tools/dwfwd_only.py > /dev/null 2>&1
python $R/tools/microbench.py pw --bwd --batch 8 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib\|amdgpu.ids" > $O/microbench_b8.txt
python $R/tools/microbench.py dw --bwd --batch 8 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib\|amdgpu.ids" >> $O/microbench_b8.txt
python $R/tools/coarse_step.py 16 > $O/coarse_step.txt 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -la $O $O/*
