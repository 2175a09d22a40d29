cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['ms_per_step_repeats']['runs'], d['roofline']['frac'], d['roofline']['whole_iteration']['frac'])
P
tail -2 gpurun_out/bench_default.err
