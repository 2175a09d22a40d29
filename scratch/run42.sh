cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python scratch/ab100.py scratch/lib_cur.so scratch/lib_nostage.so > gpurun_out/ab.txt 2>&1
cat gpurun_out/ab.txt
