cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -v "^  File \"/usr" | tail -25 > gpurun_out/t_parity.txt
timeout 900 python scratch/ab100.py default default@VIREO_VAR_FORM=2 2>&1 | tail -20 >> gpurun_out/t_parity.txt
cat gpurun_out/t_parity.txt
