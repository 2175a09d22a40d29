cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export VIREO_LDS_STAGE_TRIPS_X10=50
timeout 1500 python scratch/ab_bench.py scratch/lib_rwc64.so scratch/lib_v48.so scratch/lib_rwc64.so scratch/lib_v48.so > gpurun_out/ab_r3_10.txt 2>&1
cat gpurun_out/ab_r3_10.txt
