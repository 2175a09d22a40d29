cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/t_full3.txt
( time timeout 1200 python bench.py > gpurun_out/bench_r3_b.json 2> gpurun_out/bench_r3_b.err ) 2> gpurun_out/bench_r3_b.time
cat gpurun_out/t_full3.txt; cut -c1-400 gpurun_out/bench_r3_b.json; tail -3 gpurun_out/bench_r3_b.err; cat gpurun_out/bench_r3_b.time
