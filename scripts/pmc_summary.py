"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, the mean of every counter per
dispatch and the kernel's OWN launch metadata (registers, LDS, grid, workgroup) as the CSV reports it
for that kernel's dispatches (distinct values are listed when dispatches differ)."""
import collections, csv, glob, sys

META = ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")
for f in sorted(glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = collections.defaultdict(lambda: collections.defaultdict(set))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for m in META:
            if m in r:
                meta[k][m].add(r[m])
    for k, d in agg.items():
        desc = {m: (sorted(v, key=lambda s: float(s)) if len(v) > 1 else next(iter(v))) for m, v in meta[k].items()}
        for m in ("Grid_Size", "Workgroup_Size"):  # long lists of grids: first few
            if isinstance(desc.get(m), list) and len(desc[m]) > 6:
                desc[m] = desc[m][:3] + ["..."] + desc[m][-2:]
        print(k, desc)
        for c, v in sorted(d.items()):
            print(f"   {c:28s} {sum(v) / len(v):16.1f}  (n={len(v)})")
