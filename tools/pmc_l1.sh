R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_l1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -- python $R/tools/microbench.py pw --bwd --batch 8 > /dev/null 2> $O/p1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p2 -- python $R/tools/microbench.py pw --bwd --batch 8 > /dev/null 2> $O/p2.err
cd $R
python tools/pmc_sq.py $O/l1.json $(ls $O/p1/runc/*counter_collection.csv | head -1) $(ls $O/p2/runc/*counter_collection.csv | head -1) --filter pw_bwd_fused,pw_gemm,pw_wgrad_kernel > /dev/null
find $O -name "*.csv" -size +2M -delete
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_l1/l1.json'))['kernels']
for k,x in d.items():
    f=x.get('fractions_of_wave_cycles',{}); p=x.get('per_simd_at_2.1GHz',{})
    print(k[:60], 'us', x.get('avg_us'), {a:round(b,3) for a,b in f.items()}, {a:round(b,3) for a,b in p.items()}, 'ldsconf', round(x['counters'].get('SQ_LDS_BANK_CONFLICT',0)/max(x['counters'].get('SQ_ACTIVE_INST_LDS',1),1),3))
PY
