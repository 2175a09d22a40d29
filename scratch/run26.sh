cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python scratch/ab100.py default@VIREO_LDS_STAGE_TRIPS_X10=30 default@VIREO_LDS_STAGE_TRIPS_X10=80 default@VIREO_LDS_STAGE_TRIPS_X10=120 default@VIREO_LDS_BLOCKS=512 default > gpurun_out/ab_r3_26.txt 2>&1
cat gpurun_out/ab_r3_26.txt
