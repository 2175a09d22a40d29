cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -X faulthandler -m pytest tests -x -q -m gpu 2>&1 | grep -v "^  File \"/usr" | tail -15 > gpurun_out/t_full.txt
bash scratch/profile_round.sh r03 > gpurun_out/profile_round.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/t_full.txt; tail -3 gpurun_out/profile_round.log; cut -c1-300 gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
