cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python scratch/batch_c3.py c3 12 1,4; timeout 600 python scratch/batch_c3.py c3 5 1,3,6; timeout 600 python scratch/batch_c3.py c3 8 1,2; timeout 600 python scratch/batch_c3.py c3 16 1,4) > gpurun_out/batch.txt 2>&1
cat gpurun_out/batch.txt
