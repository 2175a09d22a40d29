cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python scratch/ab_bench.py default@VIREO_FUSE_ELBO=0 default@VIREO_FUSE_FINAL_MIN_PARTS=0 default@VIREO_FUSE_ELBO=0,VIREO_FUSE_FINAL_MIN_PARTS=0 default@VIREO_FUSE_ELBO=0 > gpurun_out/ab_r3_23.txt 2>&1
cat gpurun_out/ab_r3_23.txt
