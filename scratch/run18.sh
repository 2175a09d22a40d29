cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_cli.py -x -q -m gpu -s 2>&1 | grep -v "^\[vireo\]" | tail -40 > gpurun_out/t_full2.txt
cat gpurun_out/t_full2.txt
