#!/usr/bin/env python3
"""Fabric traffic per kernel over one training iteration, from two rocprofv3 counter passes over the same
bench.py command (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, each with --kernel-trace):

    tools/fabric_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <iterations in the run>

Reads = 2 x FETCH_SIZE (the correction prescribed for gfx950: the counter counts 128-byte requests as 64),
both counters are in KB.  Prints GB per iteration, the kernel's time per iteration (from the trace of the
FETCH_SIZE pass) and the rate while the kernel runs."""
import collections
import csv
import glob
import sys


def newest(d, pat):
    f = sorted(glob.glob(f"{d}/**/*{pat}", recursive=True))
    if not f:
        sys.exit(f"no {pat} under {d}")
    return f[-1]


def short(name):
    return name.replace("void ", "").replace("siv::", "").split("(")[0]


def counter_sum(d, counter):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(newest(d, "counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    return acc


def main():
    dfetch, dwrite, iters = sys.argv[1], sys.argv[2], float(sys.argv[3])
    fetch, write = counter_sum(dfetch, "FETCH_SIZE"), counter_sum(dwrite, "WRITE_SIZE")
    dur, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(newest(dfetch, "kernel_trace.csv"))):
        k = short(r["Kernel_Name"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        cnt[k] += 1
    print(f"# {'kernel':<52}{'ms/iter':>9}{'read GB':>10}{'write GB':>10}{'TB/s':>7}{'launches/iter':>15}")
    tot_r = tot_w = tot_ms = 0.0
    for k in sorted(dur, key=lambda k: -dur[k]):
        rd, wr, ms = 2 * fetch[k] * 1024 / 1e9 / iters, write[k] * 1024 / 1e9 / iters, dur[k] / iters
        tot_r, tot_w, tot_ms = tot_r + rd, tot_w + wr, tot_ms + ms
        if ms < 0.1:
            continue
        print(f"{k[:54]:<54}{ms:>9.2f}{rd:>10.2f}{wr:>10.2f}{(rd + wr) / ms:>7.2f}{cnt[k] / iters:>15.1f}")
    print(f"# all kernels: {tot_ms:.1f} ms, {tot_r:.1f} GB read, {tot_w:.1f} GB written per iteration = "
          f"{(tot_r + tot_w) / tot_ms:.2f} TB/s on average")


if __name__ == "__main__":
    main()
