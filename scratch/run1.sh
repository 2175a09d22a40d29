set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/t_parity.txt
timeout 1200 python scratch/ab_bench.py scratch/lib_base.so default scratch/lib_lean.so scratch/lib_ahead.so scratch/lib_probe.so scratch/lib_w12.so > gpurun_out/ab_r3_1.txt 2>&1
cat gpurun_out/t_parity.txt gpurun_out/ab_r3_1.txt
