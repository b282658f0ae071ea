"""Summarise a rocprofv3 --pmc counter_collection.csv: mean per dispatch per kernel."""
import collections, csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k, {"VGPR": rows[0]["VGPR_Count"], "grid": rows[0]["Grid_Size"]})
        for c, v in sorted(d.items()):
            print(f"   {c:24s} {sum(v) / len(v):16.1f}  (n={len(v)})")
