"""Folds a rocprofv3 kernel_trace.csv of tools/solo_trace.py into the timeline of the LAST clone:
per kernel: start offset, duration, gap to the previous kernel (all us)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a clone starts with its mask stage -- k_mask_bbox / k_mask_erode3 (the group forms too) -- or, launched on a predicted box since late
# round 4, straight with the pre-process launch (which carries the scan and erodes the mask in its tiles)
ismask = lambda r: "k_mask_bbox" in r["Kernel_Name"] or "k_mask_erode" in r["Kernel_Name"]
ispre = lambda r: "k_preprocess" in r["Kernel_Name"]
starts = [i for i, r in enumerate(rows) if (ismask(r) or ispre(r)) and (i == 0 or not ismask(rows[i - 1]))]
a = starts[-1]
b = len(rows)
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
tot_busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void sc::", "").replace("sc::", "")
    g = r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?")
    print("%8.1f  dur %7.1f  gap %6.1f  %-46s grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name[:46], "x".join(g)))
    tot_busy += e - s; prev_end = e
print("span %.1f us, busy %.1f us, kernels %d" % ((prev_end - t0) / 1e3, tot_busy / 1e3, b - a))
