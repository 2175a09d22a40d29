cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/t_parity.txt
AB_DUMP=gpurun_out/probe_rec.npy timeout 600 python scratch/ab_bench.py scratch/lib_probe.so > gpurun_out/ab_r3_5p.txt 2>&1
timeout 1200 python scratch/ab_bench.py scratch/lib_r3a.so default default@VIREO_LDS_HEAD_FIRST=0 default@VIREO_LDS_BLOCKS_CELL=260 default@VIREO_LDS_BLOCKS_CELL=1040 > gpurun_out/ab_r3_5.txt 2>&1
cat gpurun_out/t_parity.txt gpurun_out/ab_r3_5.txt
