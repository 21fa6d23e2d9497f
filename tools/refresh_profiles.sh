#!/bin/bash
# Regenerate the measurements kept under profiles/ (run on the GPU box through gpurun; writes into gpurun_out/refresh).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FILT="^RCCL\|^HIP\|^ROCm\|^Host\|^Lib\|amdgpu.ids"
# kernel stats of the FINE step alone (the default line's coarse leg would mix x3d_coarse kernels into the table); then the default line itself
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python $R/bench.py --no-coarse-roofline --no-cpu-baseline > $O/bench_rocprof.json 2> $O/bench.err
python $R/bench.py > $O/bench.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_bf16 -- python $R/bench.py --dtype bf16 > $O/bench_bf16.json 2>> $O/bench.err
# coarse stream: the kernel tables come from the EAGER step (launch counts per step); the bench lines proper are the default hipGraph replays (round 6)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_coarse -- python $R/bench.py --stream coarse --eager --no-cpu-baseline > $O/bench_coarse_eager_rocprof.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_coarse_t256 -- python $R/bench.py --stream coarse --eager --frames 256 --no-cpu-baseline > $O/bench_coarse_t256_eager_rocprof.json 2>> $O/bench.err
python $R/bench.py --stream coarse --steps 20 --staged > $O/bench_coarse.json 2>> $O/bench.err
python $R/bench.py --stream coarse --steps 20 --eager --no-cpu-baseline > $O/bench_coarse_eager.json 2>> $O/bench.err
python $R/bench.py --stream coarse --frames 256 --no-cpu-baseline > $O/bench_coarse_t256.json 2>> $O/bench.err
python $R/bench.py --stream coarse --frames 256 --eager --no-cpu-baseline > $O/bench_coarse_t256_eager.json 2>> $O/bench.err
python $R/bench.py --stream coarse --steps 20 --dtype bf16 --no-cpu-baseline > $O/bench_coarse_bf16.json 2>> $O/bench.err
python $R/bench.py --stream coarse --steps 20 --dtype fp16 --no-cpu-baseline > $O/bench_coarse_fp16.json 2>> $O/bench.err
python $R/bench.py --stream coarse --steps 20 --eager --staged --no-cpu-baseline > $O/bench_coarse_eager_staged.json 2>> $O/bench.err
python $R/bench.py --stream joint --staged > $O/bench_joint.json 2>> $O/bench.err
python $R/bench.py --stream joint --dtype bf16 > $O/bench_joint_bf16tower.json 2>> $O/bench.err
python $R/bench.py --stream joint --dtype fp16 > $O/bench_joint_fp16tower.json 2>> $O/bench.err
python $R/bench.py --dtype fp16 --no-cpu-baseline > $O/bench_fp16.json 2>> $O/bench.err
python $R/bench.py --staged --no-cpu-baseline --no-coarse-roofline > $O/bench_staged.json 2>> $O/bench.err
# HBM traffic of the depthwise forward family: separate --pmc passes, kernel trace only
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/tools/dwfwd_only.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/tools/dwfwd_only.py > /dev/null 2>&1
# SQ counters of the same workload (two passes)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq1 -- python $R/tools/dwfwd_only.py > /dev/null 2> $O/pmc_sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_sq2 -- python $R/tools/dwfwd_only.py > /dev/null 2> $O/pmc_sq2.err
# MFMA-busy / VALU / LDS counters of the pointwise kernels of layers 2-4: fp32 MFMA, 6-term split bf16, bf16 activations
for v in "0 f32 fp32mfma" "6 f32 split6" "6 bf16 bf16"; do set -- $v
  CFN_PW_SPLIT=$1 DT=$2 CFN_PWS_MAXK=100000 CFN_PWS_MAXSLABS=100 REPS=2 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_pw1_$3 -- python $R/tools/pw_only.py > /dev/null 2> $O/pmc_pw1_$3.err
  CFN_PW_SPLIT=$1 DT=$2 CFN_PWS_MAXK=100000 CFN_PWS_MAXSLABS=100 REPS=2 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/pmc_pw2_$3 -- python $R/tools/pw_only.py > /dev/null 2> $O/pmc_pw2_$3.err
done
python $R/tools/membw.py 2>&1 | grep -v "$FILT" > $O/membw.txt
python $R/tools/sal_bench.py 2>&1 | grep -v "$FILT" > $O/sal_bench.txt
python $R/tools/salb_bench.py 2>&1 | grep -v "$FILT" > $O/salb_bench.txt
# the one-pass split-bf16 pointwise backward of layer 2 (csrc/pwfuseds.hip): same-process A/B against the separate kernels, knock-out table, bit-repeat stress
{ CFN_PWF_SPLIT=2 python $R/tools/pwfs_bench.py; L3=1 python $R/tools/pwfs_bench.py; L1=1 python $R/tools/pwfs_bench.py; echo '## layer-3 variant, knock-outs (CFN_PWFS_DBG: 1 weight gradient, 2 data gradient, 8 stores)'; for d in 1 2 3 11; do echo -n "dbg=$d  "; L3=1 CFN_PWFS_DBG=$d python $R/tools/pwfs_bench.py | grep fused; done; } 2>&1 | grep -v "$FILT" > $O/pwfs_bench.txt
CFN_PWF_SPLIT=2 bash $R/tools/pwfs_knockouts.sh 2>&1 | grep -v "$FILT" > $O/pwfs_knockouts.txt
RUNS=200 python $R/tools/pwfs_stress.py 2>&1 | grep -v "$FILT" > $O/pwfs_stress.txt
{ for i in 1 2; do for v in "0 1" "1 0" "1 1" "2 1"; do set -- $v; echo -n "CFN_PWF_SPLIT=$1 CFN_PWF_L3=$2  "; CFN_PWF_SPLIT=$1 CFN_PWF_L3=$2 python $R/bench.py --no-cpu-baseline --no-coarse-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/step', d['value'], 'clips/s  figure A', d['roofline']['frac'])"; done; done; } > $O/pwfs_step_ab.txt
{ echo "# one profiled step at the benchmarked shape (8 clips x 256 frames):"; FRAMES=256 python $R/tools/glue_profile_coarse.py 8 2>&1 | grep "in the step"
  echo "# 2 clips x 64 frames, with the call sites:"; python $R/tools/glue_profile_coarse.py 2 2>&1 | grep -v "$FILT\|Warn\|_warn_once\|ROCTracer"; } > $O/glue_coarse.txt
python $R/tools/sync_debug.py 2>&1 | grep -v "$FILT" > $O/sync_debug.txt
python $R/tools/determinism_scan.py --runs 12 2>&1 | grep -v "$FILT" > $O/determinism_scan.txt
python $R/tools/microbench.py pw --bwd --batch 8 2>&1 | grep -v "$FILT" > $O/microbench_b8.txt
python $R/tools/microbench.py dw --bwd --batch 8 2>&1 | grep -v "$FILT" >> $O/microbench_b8.txt
python $R/tools/microbench_bf16.py 2>&1 | grep -v "$FILT" > $O/microbench_bf16_b8.txt
# sustained single-kernel runs with rocm-smi sampled alongside (clock / power): the evidence behind DESIGN 4j
{ for w in copy t5 dw56s1 dw28s1 dw14s1 dw7s1 dw112s2 dw56s2 dw28s2 dw14s2; do $R/tools/clk_watch.sh $w python $R/tools/busy.py $w --secs 3; done
  for w in dw56s1 dw28s1 dw14s1 dw112s2 dw56s2 dw28s2 dw14s2; do CFN_DW_FLAT=0 $R/tools/clk_watch.sh marching_$w python $R/tools/busy.py $w --secs 3; done
  for h in 7 14 28 56; do python $R/tools/dwbwd_busy.py $h; done
  CFN_DW_FLATB=0 python $R/tools/dwbwd_busy.py 7
  python $R/tools/stem_wg_bench.py; CFN_STEM_WG_OFF=1 python $R/tools/stem_wg_bench.py
  python $R/tools/dwbwd_s2_time.py; CFN_DW_FLATB=8 python $R/tools/dwbwd_s2_time.py
} 2>&1 | grep -v "$FILT" > $O/power_clock.txt
$R/tools/probe/stream_probe > $O/stream_probe.txt 2>&1
$R/tools/probe/mfma_rate_probe > $O/mfma_rate_probe.txt 2>&1
find $O -name "*kernel_trace.csv" -size +4M -delete
find $O -name "*agent_info.csv" -delete
ls -la $O $O/*
