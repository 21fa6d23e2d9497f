O=gpurun_out/r3g; mkdir -p $O
python -m pytest tests/test_hip_split.py -q 2>&1 | tail -5 > $O/split_tests.txt
for s in "96 216 14" "216 96 14" "48 108 28" "108 48 28"; do for np in 1 2; do CFN_PW_SPLIT=6 CFN_PWS_NP=$np python tools/pw_matrix.py $s 2>/dev/null | grep "act=2\|stats=0 act=0"; done; echo; done > $O/matrix.txt
CFN_PW_SPLIT=6 python tools/microbench.py pw --bwd --batch 8 > $O/mb_6.txt 2>&1
tail -2 $O/split_tests.txt; cat $O/matrix.txt; grep -A1 "^L[234]\.x" $O/mb_6.txt
