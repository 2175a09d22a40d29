cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scratch/ab100.py default > gpurun_out/ab.txt 2>&1
timeout 900 python scratch/ab_small.py default >> gpurun_out/ab.txt 2>&1
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -8 >> gpurun_out/ab.txt
cat gpurun_out/ab.txt
