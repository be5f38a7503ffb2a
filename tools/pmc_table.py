#!/usr/bin/env python
"""Per-kernel counter table of one bench.py run from three SEPARATE rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass; the SQ / GRBM counters take a third), as MI355X_MICROARCH.md 'HBM' and 'rocprofv3 PMC slots' prescribe:

    traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE   (KB units; gfx950 tallies 128-B read requests at 64 B, hence the factor 2;
                                                        counts what leaves the L2 — Infinity-Cache hits included)
    achieved TB/s      = traffic / launch duration     (duration from the kernel trace of the SQ pass: profiled, kernels serialised)
    MFMA busy          = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles of the launch), shader cycles = GRBM_GUI_ACTIVE / 8 XCDs
    clock              = GRBM_GUI_ACTIVE / 8 / duration (reads high below ~0.3 ms, see the guide's DVFS note)

usage: pmc_table.py <fetch_dir> <write_dir> <sq_dir> <out.md> "<title>" [n_kernels] [name substring filter]
"""
import collections
import csv
import glob
import sys


def load(d):
    cc = []
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        cc += list(csv.DictReader(open(f)))
    kt = {}
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            kt[r["Dispatch_Id"]] = r
    return cc, kt


def per_kernel(d):
    """name -> {counter: sum, 'n': launches, 'us': total duration}"""
    cc, kt = load(d)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in cc:
        name = r["Kernel_Name"]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (name, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            agg[name]["n"] += 1
            k = kt.get(r["Dispatch_Id"])
            if k is not None:
                agg[name]["us"] += (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e3
    return agg


def short(name, n=86):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    if cut > 0:
        name = name[:cut]
    return name if len(name) <= n else name[: n - 1] + "…"


def main():
    fetch, write, sq = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), per_kernel(sys.argv[3])
    out, title = sys.argv[4], sys.argv[5]
    top = int(sys.argv[6]) if len(sys.argv) > 6 else 12
    flt = sys.argv[7] if len(sys.argv) > 7 else ""
    names = [n for n in sorted(sq, key=lambda n: -sq[n]["us"]) if flt in n][:top]
    lines = [f"# {title}", "",
             "Three separate `rocprofv3 --kernel-trace --pmc` passes of the same command (FETCH_SIZE | WRITE_SIZE | SQ + GRBM counters); per-launch averages.",
             "`traffic` = 2 x FETCH_SIZE + WRITE_SIZE (bytes leaving the L2, Infinity-Cache hits included; gfx950 correction applied); durations are the",
             "profiled ones of the SQ pass (kernels serialised, ~2-3 % slower than in the un-profiled step).  `MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES /",
             "(1024 SIMDs x GRBM_GUI_ACTIVE / 8); `VALU/wave-cycle` = SQ_INSTS_VALU / (4 x SQ_WAVE_CYCLES) (SQ_WAVE_CYCLES counts quad-cycles);",
             "`LDS stall` = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES; `bank conflict` = SQ_LDS_BANK_CONFLICT / SQ_BUSY_CYCLES.", "",
             "| kernel | launches | avg µs | fetch MB | write MB | traffic MB | TB/s | MFMA busy | clock GHz | VALU / wave-cycle | LDS stall | bank conflict |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for n in names:
        s = sq[n]
        cnt = max(s["n"], 1.0)
        us = s["us"] / cnt
        f = fetch.get(n, {}).get("FETCH_SIZE", 0.0) / max(fetch.get(n, {}).get("n", 1.0), 1.0) * 1024.0 * 2.0
        w = write.get(n, {}).get("WRITE_SIZE", 0.0) / max(write.get(n, {}).get("n", 1.0), 1.0) * 1024.0
        gui = s.get("GRBM_GUI_ACTIVE", 0.0) / cnt / 8.0
        mfma = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / cnt
        wave = s.get("SQ_WAVE_CYCLES", 0.0) / cnt
        busy = s.get("SQ_BUSY_CYCLES", 0.0) / cnt
        row = [f"`{short(n)}`", f"{int(cnt)}", f"{us:.1f}", f"{f / 1e6:.1f}", f"{w / 1e6:.1f}", f"{(f + w) / 1e6:.1f}",
               f"{(f + w) / (us * 1e-6) / 1e12:.2f}" if us else "-",
               f"{100.0 * mfma / (1024.0 * gui):.1f} %" if gui else "-", f"{gui / (us * 1e3):.2f}" if us else "-",
               f"{s.get('SQ_INSTS_VALU', 0.0) / cnt / (4.0 * wave):.3f}" if wave else "-",
               f"{100.0 * s.get('SQ_WAIT_INST_LDS', 0.0) / cnt / wave:.1f} %" if wave else "-",
               f"{100.0 * s.get('SQ_LDS_BANK_CONFLICT', 0.0) / cnt / busy:.2f} %" if busy else "-"]
        lines.append("| " + " | ".join(row) + " |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
