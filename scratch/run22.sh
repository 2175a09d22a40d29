cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/t_parity.txt
timeout 1500 python scratch/ab_bench.py default default@VIREO_FUSE_ELBO=0,VIREO_FUSE_FINAL_MIN_PARTS=0 default default@VIREO_FUSE_ELBO=0,VIREO_FUSE_FINAL_MIN_PARTS=0 > gpurun_out/ab_r3_22.txt 2>&1
AB_CONFIG=c2 timeout 600 python scratch/ab_bench.py default default@VIREO_FUSE_ELBO=0 >> gpurun_out/ab_r3_22.txt 2>&1
cat gpurun_out/t_parity.txt gpurun_out/ab_r3_22.txt
