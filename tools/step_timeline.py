"""Print the kernel timeline of one training step out of a rocprofv3 kernel trace (steps are delimited by adam_kernel)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(("adam_kernel", "adam_rows_range_kernel"))]  # the last launch of a step
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]["End_Timestamp"])
busy_end = t0
idle = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - busy_end
    if gap > 0:
        idle += gap
    busy_end = max(busy_end, e)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {'  gap %6.1f' % (gap / 1e3) if gap > 3000 else '            '} s{r['Stream_Id']:>2s} {r['Kernel_Name'][:70]}")
print("step %.1f us, GPU idle %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3, idle / 1e3))
