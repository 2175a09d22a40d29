cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python scratch/ab100.py default default@VIREO_LDS_STAGE_TRIPS_X10=200 default@VIREO_LDS_STAGE_TRIPS_X10=400 > gpurun_out/ab_r3_27.txt 2>&1
cat gpurun_out/ab_r3_27.txt
