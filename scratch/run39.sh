cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles_r03
timeout 900 python scratch/collect_traffic.py c3 > gpurun_out/profiles_r03/r03_traffic_c3.log 2>&1
cp profiles/traffic_c3.json gpurun_out/profiles_r03/traffic_c3.json
timeout 900 python scratch/collect_traffic.py c5 > gpurun_out/profiles_r03/r03_traffic_c5.log 2>&1
cp profiles/traffic_c5.json gpurun_out/profiles_r03/traffic_c5.json
timeout 3000 python -X faulthandler -m pytest tests -x -q -m gpu 2>&1 | grep -v "^  File \"/usr" | tail -15 > gpurun_out/t_full.txt
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/t_full.txt; tail -3 gpurun_out/profiles_r03/r03_traffic_c3.log; cut -c1-300 gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
