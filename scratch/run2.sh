cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/t_parity.txt
timeout 1200 python scratch/ab_bench.py default scratch/lib_ring1024.so scratch/lib_pf6.so scratch/lib_l2pf1.so scratch/lib_l2pf2.so scratch/lib_probe.so default > gpurun_out/ab_r3_2.txt 2>&1
cat gpurun_out/t_parity.txt gpurun_out/ab_r3_2.txt
