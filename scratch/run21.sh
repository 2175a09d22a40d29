cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python scratch/ab_bench.py default default@VIREO_THETA_BLOCKS_PER_CU=1 default@VIREO_THETA_BLOCKS_PER_CU=1,VIREO_GT_BLOCKS_PER_CU=4 default@VIREO_THETA_BLOCKS_PER_CU=1,VIREO_GT_BLOCKS_PER_CU=2 default@VIREO_THETA_BLOCKS_PER_CU=2,VIREO_GT_BLOCKS_PER_CU=4,VIREO_FUSE_THETA_MAX_PARTS=512 default@VIREO_GT_BLOCKS_PER_CU=16 > gpurun_out/ab_r3_21.txt 2>&1
cat gpurun_out/ab_r3_21.txt
