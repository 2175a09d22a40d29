cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB_DUMP=gpurun_out/probe_rec.npy timeout 600 python scratch/ab_bench.py scratch/lib_probe.so > gpurun_out/ab_r3_4.txt 2>&1
cat gpurun_out/ab_r3_4.txt | cut -c1-300
