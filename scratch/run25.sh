cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/t_parity.txt
timeout 1500 python scratch/ab100.py default default@VIREO_LDS_XCD_PHASE=0 default default@VIREO_LDS_XCD_PHASE=0 > gpurun_out/ab_r3_25.txt 2>&1
cat gpurun_out/t_parity.txt gpurun_out/ab_r3_25.txt
