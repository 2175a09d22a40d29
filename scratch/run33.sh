cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st_c2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_c2 -o c2 -- env VIREO_FUSE_THETA_MAX_PARTS=0 python $REPO/scratch/ab_small.py child > /tmp/c2.out 2>&1
f=$(find /tmp/st_c2 -name "*kernel_stats.csv" | head -1)
cp $f $REPO/gpurun_out/c2_small_stats.csv
python - <<'P'
import csv
for r in csv.DictReader(open('/root/repo/gpurun_out/c2_small_stats.csv')):
    print(r['Name'].split('(')[0][:50], r['Calls'], round(float(r['AverageNs'])/1e3,2), r['MinNs'])
P
