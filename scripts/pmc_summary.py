"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv.  python scripts/pmc_summary.py file.csv [name filter]"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
flt = sys.argv[2] if len(sys.argv) > 2 else "mmf"
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    kn = r["Kernel_Name"].split("(")[0][-44:] + f" grid {int(r['Grid_Size']) // max(1, int(r['Workgroup_Size']))}"
    if flt not in kn:
        continue
    acc[kn][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (kn, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); calls[kn] += 1
for kn, d in sorted(acc.items()):
    print(kn, "calls", calls[kn], " ".join(f"{c}={v / calls[kn]:.4g}" for c, v in sorted(d.items())))
