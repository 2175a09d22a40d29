cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_cli.py -x -q -m gpu -s 2>&1 | grep -v "^\[vireo\]" | grep "differ\|passed\|failed\|Error\|assert" | tail -50 > gpurun_out/t_cli.txt
cat gpurun_out/t_cli.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "whole_protocol" 2>&1 | grep "per-iteration\|passed\|failed" > gpurun_out/t_arb.txt
cat gpurun_out/t_arb.txt
