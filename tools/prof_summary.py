#!/usr/bin/env python
"""Turn a rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv) into a short markdown table under profiles/."""
import csv
import sys


def main(src: str, dst: str, title: str, steps: int) -> None:
    rows = list(csv.DictReader(open(src)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(dst, "w") as f:
        f.write(f"# {title}\n\nsource: `rocprofv3 --kernel-trace --stats` ({src.split('/')[-1]}); {steps} steps profiled; "
                f"total kernel time {tot / 1e6:.1f} ms = {tot / 1e6 / steps:.1f} ms/step\n\n")
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for r in rows:
            if float(r["Percentage"]) < 0.005:
                continue
            f.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]))
