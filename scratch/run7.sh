cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scratch/probe_repeat.py 2>&1 | head -4 > gpurun_out/probe_repeat.txt
VIREO_LDS_PLAN_REVERSE=1 timeout 600 python scratch/probe_repeat.py 2>&1 | head -4 >> gpurun_out/probe_repeat.txt
timeout 1200 python scratch/ab_bench.py default default@VIREO_LDS_STAGE_TRIPS_X10=50 default@VIREO_LDS_PLAN_REVERSE=1 >> gpurun_out/probe_repeat.txt 2>&1
cat gpurun_out/probe_repeat.txt
