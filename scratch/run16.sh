cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python scratch/ab_bench.py default scratch/lib_c80.so scratch/lib_c96.so scratch/lib_v48c80.so scratch/lib_aheadc80.so default > gpurun_out/ab_r3_16.txt 2>&1
cat gpurun_out/ab_r3_16.txt
