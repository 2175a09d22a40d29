"""HBM traffic of the sparse passes from rocprofv3 PMC counters -> profiles/traffic_<config>.json.

Run on the GPU box from the repository root:
    python scratch/collect_traffic.py [c3] [profiles/r02]
Two separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md, HBM
section) over `python bench.py --only-headline --steps 3 --warmup 1`.  Units: both counters are
KiB.  gfx950 correction from the guide: FETCH_SIZE reports exactly half of the bytes of wide
(16 B / lane) coalesced streaming reads -- every read of both passes -- so it is doubled;
WRITE_SIZE is taken as is.  The record carries the hash of the kernel sources it was measured on;
bench.py quotes it only when that hash matches its own build.  Keys: vrx_spmm_lds<0> = the variant
pass, vrx_spmm_lds<1> = the cell pass."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "c3"
prefix = sys.argv[2] if len(sys.argv) > 2 else "profiles/r04"
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--only-headline", "--steps", "3", "--warmup", "1",
       "--config", config]
if config == "c5":      # BinomMixtureVB clone mode (BASELINE.json configs[4]): its own driver
    cmd = [sys.executable, os.path.join(ROOT, "tests", "perf", "bench_bmm.py"), "--passes-only"]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    out = "/tmp/pmc_%s_%s" % (config, counter)   # (one directory per config and counter, emptied first)
    subprocess.run(["rm", "-rf", out], check=False)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out,
                    "-o", config, "--"] + cmd, cwd=ROOT if config == "c5" else "/tmp", env=env, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # The two passes may be the SAME instantiation (the variant pass over virtual rows runs the cell
    # pass's kernel), so they are told apart by dispatch order: every iteration launches the
    # variant pass, then the cell pass (where the MODE template argument differs, it decides).
    rows = []
    for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.match(r"void vrx_spmm_lds<\d+, (\d)", r["Kernel_Name"])
            if m and r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), int(m.group(1)), float(r["Counter_Value"])))
    rows.sort()
    by_mode = any(mode == 0 for _, mode, _ in rows)  # (two instantiations: the name tells)
    for i, (_, mode, v) in enumerate(rows):
        which = ("cell" if mode else "variant") if by_mode else ("cell" if i % 2 else "variant")
        vals["vrx_spmm_lds<%d>" % (which == "cell")][counter].append(v)
kernels = {}
for k, c in vals.items():
    fetch = sum(c["FETCH_SIZE"]) / max(len(c["FETCH_SIZE"]), 1)
    write = sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)
    kernels[k] = dict(FETCH_SIZE_KiB=fetch, WRITE_SIZE_KiB=write, launches=len(c["FETCH_SIZE"]),
                      traffic_bytes=int(2 * fetch * 1024 + write * 1024))
doc = dict(_comment=__doc__.split("\n\n")[1].replace("\n", " "), config=config,
           kernel_source_hash=bench.kernel_source_hash(), stream=bench.stream_choice(), kernels=kernels)
path = os.environ.get("VIREO_TRAFFIC_OUT") or os.path.join(ROOT, "profiles", "traffic_%s.json" % config)
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps(doc, indent=1))
