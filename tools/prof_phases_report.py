"""Reads the p_counter_collection.csv of a tools/prof_phases.py run: per-wave counter values per stop and their increments."""
import csv, sys, collections
names = ["1 raw stats", "7 tail/pz", "2 SG", "3 T", "4 sweeps", "5 runs+cross", "6 estimators", "11 y ready", "12 Dp+flat", "13 causal", "14 anti", "15 parabola", "0 finish"]
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "icpc_kernel" not in r["Kernel_Name"]: continue
    rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = rows.get(int(r["Dispatch_Id"]), {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
disp = list(rows.values())
ctrs = [c for c in disp[0] if c != "SQ_WAVES"]
prev = {c: 0.0 for c in ctrs}
print("stop".ljust(16) + "".join(c.rjust(22) for c in ctrs))
for nm, d in zip(names, disp):
    w = d.get("SQ_WAVES", 1.0)
    print(nm.ljust(16) + "".join(f"{d[c]/w:12.1f} (+{d[c]/w-prev[c]:7.1f})" for c in ctrs))
    prev = {c: d[c] / w for c in ctrs}
