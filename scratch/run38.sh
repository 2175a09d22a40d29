cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st_c3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_c3 -o c3 -- python $REPO/bench.py --no-cpu --no-c4 --steps 100 --warmup 50 --config c3 > /tmp/c3.out 2>&1
f=$(find /tmp/st_c3 -name "*kernel_stats.csv" | head -1)
cp $f $REPO/gpurun_out/c3_stats.csv
python - <<'P'
import csv
for r in csv.DictReader(open('/root/repo/gpurun_out/c3_stats.csv')):
    if int(r['Calls'])>=100: print(r['Name'].split('(')[0][:50], r['Calls'], round(float(r['AverageNs'])/1e3,2), r['MinNs'])
P
