O=gpurun_out/r3l; mkdir -p $O
python -m pytest tests/test_hip_bf16.py -q 2>&1 | tail -5 > $O/bf16_tests.txt; tail -4 $O/bf16_tests.txt
for d in 1 0; do echo "DIRECT=$d"; CFN_PWB_WG_DIRECT=$d python tools/microbench_bf16.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib"; done > $O/mb_bf16.txt; cat $O/mb_bf16.txt
for d in 1 0; do CFN_PWB_WG_DIRECT=$d python bench.py --dtype bf16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 step direct=$d', d['value'], d['ms_per_step'])"; done
