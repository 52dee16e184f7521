"""Kernel durations and launch-to-launch gaps of the step kernels from a rocprofv3 --kernel-trace csv (gpurun_out/prof_<tag>/stats/*kernel_trace.csv)."""
import csv, glob, sys
import numpy as np
for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "usim_step" in r["Kernel_Name"] and "ELi1EEE" not in r["Kernel_Name"][-40:]]
    rows = [r for r in csv.DictReader(open(f)) if "usim_step" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    st = np.array([int(r["Start_Timestamp"]) for r in rows]); en = np.array([int(r["End_Timestamp"]) for r in rows])
    dur, gap = en - st, st[1:] - en[:-1]
    ok = gap < 50000          # drop host-side pauses between blocks
    print(f"{f}: {len(rows)} step kernels; duration mean {dur.mean() / 1e3:.2f} us median {np.median(dur) / 1e3:.2f} us; "
          f"gap to next launch mean {gap[ok].mean() / 1e3:.2f} us median {np.median(gap[ok]) / 1e3:.2f} us; start-to-start median {np.median(np.diff(st)[ok]) / 1e3:.2f} us")
