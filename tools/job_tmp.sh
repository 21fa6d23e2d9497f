O=gpurun_out/r3k; mkdir -p $O
CFN_DW_CP_TALL=3 python -m pytest tests/test_hip_ops.py tests/test_hip_fullsize.py -q -k "dwconv3d" 2>&1 | tail -3
for t in 0 3; do echo "TALL=$t"; CFN_DW_CP_TALL=$t python tools/microbench.py dw --batch 8 2>/dev/null | grep "@56\|@28"; done
