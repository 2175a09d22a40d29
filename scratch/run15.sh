cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
VIREO_LIB=$GRAFT_REPO_ROOT/scratch/lib_pipe.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/t_pipe.txt
timeout 1500 python scratch/ab_bench.py default scratch/lib_pipe.so scratch/lib_pipeprobe.so > gpurun_out/ab_r3_15.txt 2>&1
cat gpurun_out/t_pipe.txt; cut -c1-900 gpurun_out/ab_r3_15.txt
