cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles_r03
timeout 900 python scratch/collect_traffic.py c5 > gpurun_out/profiles_r03/r03_traffic_c5.log 2>&1
cp profiles/traffic_c5.json gpurun_out/profiles_r03/traffic_c5.json
tail -20 gpurun_out/profiles_r03/r03_traffic_c5.log
