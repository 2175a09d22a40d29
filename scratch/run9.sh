cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export VIREO_LDS_STAGE_TRIPS_X10=50
timeout 1500 python scratch/ab_bench.py default scratch/lib_h1.so scratch/lib_h2.so scratch/lib_h4.so scratch/lib_h5.so scratch/lib_rwc64.so > gpurun_out/ab_r3_9.txt 2>&1
cat gpurun_out/ab_r3_9.txt
