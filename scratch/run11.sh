cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/t_full.txt
timeout 900 python bench.py --no-cpu > gpurun_out/bench_r3_a.json 2> gpurun_out/bench_r3_a.err
cat gpurun_out/t_full.txt; cut -c1-600 gpurun_out/bench_r3_a.json; tail -3 gpurun_out/bench_r3_a.err
