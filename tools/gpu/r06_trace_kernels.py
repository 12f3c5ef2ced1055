"""Distinct (kernel, grid, workgroup, LDS, VGPR, SGPR, scratch) rows of a rocprofv3 kernel-trace CSV whose name matches a pattern."""
import csv, sys, collections
path, pat = sys.argv[1], sys.argv[2].split(",")
rows = collections.OrderedDict()
with open(path) as f:
    for r in csv.DictReader(f):
        n = r.get("Kernel_Name", "")
        if not any(p in n for p in pat):
            continue
        key = (n[:60],) + tuple(r.get(c) for c in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X", "LDS_Block_Size", "Scratch_Size",
                                                    "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        rows.setdefault(key, []).append(d)
print("kernel grid_x grid_y grid_z wg lds scratch vgpr agpr sgpr : calls avg_us")
for k, v in rows.items():
    print(*k, ":", len(v), round(sum(v) / len(v), 1))
