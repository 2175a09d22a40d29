#!/bin/bash
# Round profiles on the GPU box: kernel stats (c3, c2, c5), SQ counters and HBM traffic of c3.
# usage (from the repository root, through gpurun): bash scratch/profile_round.sh r02
set -u
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in c3 c2 c5; do
  rm -rf /tmp/st_$cfg
  if [ $cfg = c5 ]; then  # BinomMixtureVB clone mode (BASELINE.json configs[4])
    (cd $REPO && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$cfg -o $cfg -- \
        python tests/perf/bench_bmm.py > $OUT/${TAG}_bench_${cfg}_under_rocprof.json 2> /tmp/st_$cfg.err)
  else
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$cfg -o $cfg -- \
        python $REPO/bench.py --only-headline --steps 100 --warmup 50 --config $cfg > $OUT/${TAG}_bench_${cfg}_under_rocprof.json 2> /tmp/st_$cfg.err
  fi
  f=$(find /tmp/st_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${cfg}_kernel_stats.csv
done
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/pmc_sq_$i
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmc_sq_$i -o c3 -- \
      python $REPO/bench.py --only-headline --steps 3 --warmup 1 --config c3 > /dev/null 2> /tmp/pmc_sq_$i.err
done
mkdir -p /tmp/pmc_sq_all && cp -r /tmp/pmc_sq_1 /tmp/pmc_sq_2 /tmp/pmc_sq_all/ 2>/dev/null
python $REPO/scratch/pmc_summary.py /tmp/pmc_sq_all > $OUT/${TAG}_c3_pmc_sq.txt
cd $REPO && timeout 900 python scratch/collect_traffic.py c3 > $OUT/${TAG}_traffic_c3.log 2>&1
cp profiles/traffic_c3.json $OUT/traffic_c3.json
# the default build's stream beside it (contiguous slabs; not the record bench.py quotes)
cd $REPO && VIREO_BALANCE=0 VIREO_TRAFFIC_OUT=$OUT/traffic_c3_default_build.json timeout 900 python scratch/collect_traffic.py c3 > $OUT/${TAG}_traffic_c3_default_build.log 2>&1
cd $REPO && timeout 900 python scratch/collect_traffic.py c5 > $OUT/${TAG}_traffic_c5.log 2>&1
cp profiles/traffic_c5.json $OUT/traffic_c5.json
ls -la $OUT
