O=gpurun_out/r3p; mkdir -p $O
python -m pytest tests/test_hip_split.py -q 2>&1 | tail -2
python tools/microbench.py pw --bwd --batch 8 2>/dev/null | grep -A1 "^L[23]\.x"
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fine', d['value'], d['ms_per_step'])"; done
python bench.py --stream coarse --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coarse', d['value'], d['ms_per_step'])"
