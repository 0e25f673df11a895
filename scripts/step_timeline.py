"""Print the kernel timeline of the last simulated rank step from a rocprofv3 kernel trace of scripts/overlap_sim.py."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "scan_b16x" in r["Kernel_Name"]]
last = idx[-4]
j = last
while j > 0 and "select_staged" not in rows[j]["Kernel_Name"]:
    j -= 1
t0 = int(rows[j]["End_Timestamp"])
end = [i for i, r in enumerate(rows) if "select_staged" in r["Kernel_Name"]][-1]
prev = t0
for r in rows[j + 1:end + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:60]))
    prev = e
