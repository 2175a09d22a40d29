cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/t_parity.txt
timeout 1500 python scratch/ab100.py default default@VIREO_VAR_FORM=2 > gpurun_out/ab_r3_28.txt 2>&1
cat gpurun_out/t_parity.txt gpurun_out/ab_r3_28.txt
