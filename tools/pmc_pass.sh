#!/bin/bash
# One rocprofv3 counter pass over a bench_conv.py run, summed per kernel name:
#   tools/pmc_pass.sh <out-dir> "<counters>" <bench_conv args...>
export TMPDIR=/tmp
OUT=$1; shift
CNT=$1; shift
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -- python3 tools/bench_conv.py "$@" > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(1)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); n[k] += 1
for k in acc:
    if n[k] >= 1 and ("k3_" in k or "k2_" in k or "bn_" in k or "one_to" in k or "c_to" in k or "thin" in k):
        print(k, "dispatches", n[k], {c: f"{v / n[k]:.4g}" for c, v in acc[k].items()})
PY
