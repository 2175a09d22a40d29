cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python scratch/ab_bench.py default@VIREO_LDS_STAGE_TRIPS_X10=50 default@VIREO_LDS_STAGE_TRIPS_X10=70 scratch/lib_rwc64.so@VIREO_LDS_STAGE_TRIPS_X10=50 > gpurun_out/ab_r3_8.txt 2>&1
cat gpurun_out/ab_r3_8.txt
