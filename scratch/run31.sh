cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -X faulthandler -m pytest tests -x -q -m gpu 2>&1 | grep -v "^  File \"/usr" | tail -25 > gpurun_out/t_full.txt
timeout 600 python bench.py > gpurun_out/bench_form3.json 2> gpurun_out/bench_form3.err
cat gpurun_out/t_full.txt; cat gpurun_out/bench_form3.json; tail -3 gpurun_out/bench_form3.err
