cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st_c3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-c4 --steps 20 --warmup 3 --config c3 > $GRAFT_REPO_ROOT/gpurun_out/prof/bench_c3_under_rocprof.json 2> /tmp/st_c3.err
f=$(find /tmp/st_c3 -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/prof/c3_kernel_stats.csv
cut -c1-90 $f | head -14; awk -F, 'NR>1{print $1}' $f | head -0
python3 - <<'PY'
import csv,sys,glob
f=glob.glob('/tmp/st_c3/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if n.startswith('vrx_') or 'vrx_' in n[:12]:
        print('%-40s calls %4s avg %10.1f us'%(n[:40],r['Calls'],float(r['AverageNs'])/1e3))
PY
