O=gpurun_out/r3i; mkdir -p $O
for t in 0 6; do
CFN_PW_SPLIT=$t python bench.py --stream coarse --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coarse split=$t', d['value'], d['ms_per_step'])"
CFN_PW_SPLIT=$t python bench.py --stream joint --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('joint split=$t', d['value'], d['ms_per_step'])"
CFN_PW_SPLIT=$t python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fine split=$t', d['value'], d['ms_per_step'])"
done > $O/split_ab.txt; cat $O/split_ab.txt
