cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
VIREO_LIB=$PWD/scratch/lib_head.so timeout 600 python tests/perf/bench_bmm.py 2>&1 | head -1 | cut -c1-260 >> gpurun_out/t.txt
timeout 600 python tests/perf/bench_bmm.py 2>&1 | head -1 | cut -c1-260 >> gpurun_out/t.txt
done; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4 >> gpurun_out/t.txt
cat gpurun_out/t.txt
