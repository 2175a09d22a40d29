cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/t_parity.txt
timeout 1200 python scratch/ab_bench.py scratch/lib_r3a.so default scratch/lib_probe.so default@VIREO_LDS_BLOCKS=512 default@VIREO_LDS_STAGE_TRIPS_X10=0 > gpurun_out/ab_r3_3.txt 2>&1
cat gpurun_out/t_parity.txt gpurun_out/ab_r3_3.txt
