#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_round.sh (gpurun_out/prof_rNN/) into the files
kept under profiles/: the kernel-stats CSV, a markdown summary of the dominant kernel's trace
and PMC passes, and dominant_kernel_traffic.json (read by bench.py)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{TAG}")
DST = os.path.join(ROOT, "profiles")
DOM = "k3_fwd_wino2d"
DOM_GRID = "131072"           # one 512-thread block per CU (256): the kernel walks its bricks as one chunk stream
DOM_MIN_MS = 2.0              # ... a grid the 40x48x40 level shares (0.5 ms per launch); the 80x96x80 launches take ~4 ms
ALGO_BYTES = 2 * 8 * 80 * 96 * 80 * 64 * 4 + 27 * 64 * 64 * 4     # read x + write y + weights
ALGO_FLOP = 2.0 * 27 * 64 * 64 * 80 * 96 * 80 * 8


def one(pattern):
    """Newest match (gpurun merges successive runs into the same local directory)."""
    return max(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)


def counters(d):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(one(f"{d}/vm/*_counter_collection.csv"))):
        if DOM in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    bench = json.loads(open(os.path.join(SRC, "bench_default.json")).read().strip().splitlines()[-1])
    vps = bench["value"]
    for old in glob.glob(os.path.join(DST, f"{TAG}_bench_kernel_stats_*.csv")):
        os.remove(old)
    shutil.copy(one("trace/vm/*_kernel_stats.csv"), os.path.join(DST, f"{TAG}_bench_kernel_stats_{vps:.1f}vps.csv"))
    if glob.glob(os.path.join(SRC, "trace_serial/vm/*_kernel_stats.csv")):
        shutil.copy(one("trace_serial/vm/*_kernel_stats.csv"),
                    os.path.join(DST, f"{TAG}_bench_kernel_stats_serial.csv"))
    def dom_durations(d):
        return [t for t in ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                            for r in csv.DictReader(open(one(f"{d}/vm/*_kernel_trace.csv")))
                            if DOM in r["Kernel_Name"] and r["Grid_Size_X"] == DOM_GRID) if t >= DOM_MIN_MS]
    # In the default run the weight gradients share the chip with the backward launches of this kernel
    # (side stream), which stretches those; the serial trace has every launch alone on the chip, like the
    # forward launches bench.py times live.
    serial = bool(glob.glob(os.path.join(SRC, "trace_serial/vm/*_kernel_trace.csv")))
    durs = dom_durations("trace_serial" if serial else "trace")
    avg_ms = sum(durs) / len(durs)
    durs_default = dom_durations("trace")
    avg_default = sum(durs_default) / len(durs_default)
    fetch, write, sq = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_sq")
    fetch_b, write_b = fetch["FETCH_SIZE"] * 1024, write["WRITE_SIZE"] * 1024
    traffic = 2 * fetch_b + write_b            # gfx950: FETCH_SIZE counts 128-B requests as 64 B
    simd_cycles = sq["GRBM_GUI_ACTIVE"] / 8 * 1024       # per-XCD active cycles x 256 CUs x 4 SIMDs
    with open(os.path.join(DST, "dominant_kernel_traffic.json"), "w") as f:
        json.dump({"kernel": DOM, "workload": "3x3x3 conv 64->64 @80x96x80, batch 8, forward",
                   "hbm_bytes_per_launch": int(traffic), "fetch_size_bytes_raw": int(fetch_b),
                   "write_size_bytes": int(write_b), "algorithmic_bytes_per_launch": ALGO_BYTES,
                   "note": "traffic = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes); FETCH_SIZE doubled as "
                           "prescribed for gfx950; separate --pmc passes (tools/profile_round.sh)"}, f, indent=1)
    with open(os.path.join(DST, f"{TAG}_pmc_conv.md"), "w") as f:
        f.write(f"""# {TAG}: dominant kernel `{DOM}` (3x3x3 conv 64->64 @80x96x80, batch 8, fp32)

Produced by `tools/profile_round.sh` on one MI355X, summarised by `tools/summarize_profiles.py`.

## bench.py (default flags) on the same box
`{json.dumps({k: bench[k] for k in ("value", "unit", "ms_per_step", "roofline", "cpu_baseline")})}`

## rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1`
* launches of this kernel at the dominant shape (grid {DOM_GRID}; forward -- the input gradients run the F(2,3) x F(4,3) form, `k3_fwd_wino2x4`), weight
  gradients on the main stream (`SIV_OVERLAP_WGRAD=0`, every launch alone on the chip): {len(durs)}, average
  **{avg_ms:.3f} ms** (bench.py's live HIP-event figure over the forward launches: {bench["roofline"]["avg_launch_ms"]} ms)
  = {ALGO_FLOP / avg_ms / 1e9:.1f} TFLOP/s algorithmic; in the default run the backward launches share the chip with
  side-stream weight gradients and the same {len(durs_default)} launches average {avg_default:.3f} ms
* full per-kernel tables: `{TAG}_bench_kernel_stats_{vps:.1f}vps.csv` (default run), `{TAG}_bench_kernel_stats_serial.csv`

## PMC passes (tools/bench_conv.py --shapes v0_64_64 --ops fwd, one counter group per run)
| counter | per launch |
|---|---|
| FETCH_SIZE | {fetch["FETCH_SIZE"]:.0f} KB = {fetch_b / 1e9:.3f} GB raw, x2 (gfx950 correction) = {2 * fetch_b / 1e9:.3f} GB |
| WRITE_SIZE | {write["WRITE_SIZE"]:.0f} KB = {write_b / 1e9:.3f} GB (= the output tensor exactly) |
| HBM traffic (2*FETCH + WRITE) | **{traffic / 1e9:.2f} GB** vs algorithmic {ALGO_BYTES / 1e9:.2f} GB |
| SQ_VALU_MFMA_BUSY_CYCLES | {sq["SQ_VALU_MFMA_BUSY_CYCLES"]:.4g} |
| GRBM_GUI_ACTIVE (sum of 8 XCDs) | {sq["GRBM_GUI_ACTIVE"]:.4g} -> {sq["GRBM_GUI_ACTIVE"] / 8:.4g} cycles per launch |
| MFMA pipe busy | **{100 * sq["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles:.1f} %** of SIMD-cycles |
| SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY / SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY | {sq["SQ_WAVE_CYCLES"]:.4g} / {sq["SQ_WAIT_INST_ANY"]:.4g} / {sq["SQ_WAIT_ANY"]:.4g} / {sq["SQ_ACTIVE_INST_ANY"]:.4g} |

Reading: the kernel applies Winograd F(2x2,3x3) over (H, W), so it issues 12/27 of the direct-convolution
multiply-adds: the TFLOP/s above are ALGORITHMIC (direct-convolution) rates; the matrix pipe itself
runs at 4/9 of them (MFMA busy ~ that / peak).  The fetch side reads, per 4x8x8 brick and 8-channel chunk,
the raw halo image once (6x10x10 voxels = 2.34x the brick; the 4x4 patches are cut out of LDS): the halo
overlap between neighbouring bricks is served by L2 / the Infinity Cache (the counter sits on the L2's fabric
side and includes Infinity-Cache hits), and at
{traffic / 1e9 / (avg_ms / 1e3) / 1e3:.2f} TB/s the kernel is far from the HBM roof.
""")
    print("wrote profiles for", TAG, "avg_ms", avg_ms, "traffic GB", traffic / 1e9)


if __name__ == "__main__":
    main()
