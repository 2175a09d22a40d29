cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -v "^  File \"/usr" | head -40 > gpurun_out/t_parity.txt
timeout 600 python scratch/ab100.py default 2>&1 | tail -20 >> gpurun_out/t_parity.txt
cat gpurun_out/t_parity.txt
