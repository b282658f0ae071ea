"""Cut a rocprofv3 kernel trace's composite_kernel dispatches (time order) into the cases scripts/prof_single5.py ran.
    python scripts/trace_cases.py <kernel_trace.csv> <cases.json> [label]
Per case: launches, grid (workgroups x threads), mean / min duration over the last 3/4 of its launches, algorithmic
bytes / mean / 8 TB/s."""
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "composite_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = json.load(open(sys.argv[2]))
label = sys.argv[3] if len(sys.argv) > 3 else ""
assert len(rows) == sum(c["launches"] for c in seq), (len(rows), sum(c["launches"] for c in seq))
i = 0
for c in seq:
    part = rows[i:i + c["launches"]]
    i += c["launches"]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in part][len(part) // 4:]
    grids = sorted({(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_X"])) for r in part})
    mean = sum(d) / len(d) / 1e3
    print(f"{label:8s} {c['case']:12s} {len(d):3d} launches  grid {grids}  mean {mean:6.2f} us  min {min(d) / 1e3:6.2f} us  "
          f"{c['bytes'] / 1e6:7.1f} MB  {c['bytes'] / (mean * 1e-6) / 8e12:.3f} of 8 TB/s")
