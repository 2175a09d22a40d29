cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(python scratch/first_run.py a; python scratch/first_run.py b) > gpurun_out/first.txt 2>&1
cat gpurun_out/first.txt
