cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python scratch/ab_bench.py default scratch/lib_nt.so default scratch/lib_nt.so > gpurun_out/ab_r3_20.txt 2>&1
cat gpurun_out/ab_r3_20.txt
