O=gpurun_out/r3j; mkdir -p $O
python -m pytest tests/test_torchlib.py tests/test_hip_models.py -m gpu -q -x 2>&1 | tail -40 > $O/torchlib.txt; tail -40 $O/torchlib.txt
