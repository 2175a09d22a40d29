cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scratch/probe_repeat.py > gpurun_out/probe_repeat.txt 2>&1
tail -4 gpurun_out/probe_repeat.txt
