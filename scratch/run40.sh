cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/t.txt
timeout 600 python tests/perf/bench_bmm.py >> gpurun_out/t.txt 2>&1
timeout 600 python scratch/ab_small.py default >> gpurun_out/t.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c5 or bmm or clone" 2>&1 | tail -3 >> gpurun_out/t.txt
cat gpurun_out/t.txt | cut -c1-1500
