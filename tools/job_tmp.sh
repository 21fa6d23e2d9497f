O=gpurun_out/r3h; mkdir -p $O
python -m pytest tests/test_hip_train.py -m gpu -q -k "forward_video" 2>&1 | tail -15 > $O/fv.txt; tail -5 $O/fv.txt
