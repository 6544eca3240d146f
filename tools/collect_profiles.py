"""Copies the judged summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/: the bench line (headline +
secondary results), rocprofv3 kernel statistics of the bench command and of the "next"-row timers, and the HBM-traffic records
built from the FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE x2 on gfx950 for 16 B/lane streaming loads, MI355X_MICROARCH.md HBM
section) — one record per workload of the bench command (bench.py looks them up by kernel name, batch size and trace length).
Usage: python tools/collect_profiles.py r02a"""
import csv, json, os, re, shutil, sys, collections
tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(R, "gpurun_out", tag), os.path.join(R, "profiles")
with open(os.path.join(src, "bench.json")) as f:
    line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
with open(os.path.join(dst, f"{tag}_bench.json"), "w") as f:
    f.write(line + "\n")
bench = json.loads(line)
shutil.copy(os.path.join(src, "stats", "p_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
# the same run's kernel trace without each kernel's FIRST launch (cold instruction cache / clocks): what the bench's timed steps see;
# and the launches in the STEADY state apart: the GPU clock ramps for ~50 ms after an idle gap (profiles/r04_pz_trap_launch_spread.txt),
# so a launch counts as steady when the library's kernels have been running back to back (gaps < 2 ms) for at least 60 ms before it
kt = os.path.join(src, "stats", "p_kernel_trace.csv")
if os.path.exists(kt):
    per = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(kt)) if r["Kernel_Name"].startswith("void ldsp::")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    busy_since, prev_end = None, None
    for r in rows:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if prev_end is None or st - prev_end > 2_000_000:
            busy_since = st
        prev_end = en
        key = (r["Kernel_Name"].replace("void ", "").split("(")[0], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
        per[key].append((st, en - st, st - busy_since >= 60_000_000))
    with open(os.path.join(dst, f"{tag}_bench_kernel_stats_warm.csv"), "w") as g:
        g.write("kernel,traces,launches,first_launch_ns,warm_launches,warm_avg_ns,warm_min_ns,warm_max_ns,steady_launches,steady_avg_ns,steady_min_ns,steady_max_ns\n")
        for (k, nd), v in sorted(per.items()):
            v.sort()
            d = [x[1] for x in v]
            w = d[1:] if len(d) > 1 else d
            sd = [x[1] for x in v if x[2]]
            steady = f"{len(sd)},{sum(sd) / len(sd):.0f},{min(sd)},{max(sd)}" if sd else "0,,,"
            g.write(f'"{k}",{nd},{len(d)},{d[0]},{len(w)},{sum(w) / len(w):.0f},{min(w)},{max(w)},{steady}\n')
for t in ("gpu_time_grid", "gpu_time_compressed", "gpu_time_multi_intersect"):
    ks = os.path.join(src, f"stats_{t}", "p_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(dst, f"{tag}_{t[9:]}_kernel_stats.csv"))
    lg = os.path.join(src, f"{t}.log")
    if os.path.exists(lg):
        with open(lg) as f, open(os.path.join(dst, f"{tag}_{t[9:]}_timing.txt"), "w") as g:
            g.write("".join(l for l in f if "amdgpu.ids" not in l and not l.startswith("[rocprofv3]") and "rocprofiler" not in l
                            and not re.match(r"[WEI]\d{8} ", l)))   # (glog lines of the profiler itself)
# workloads of the bench command: which (n_traces, L) a kernel of the PMC passes ran on.  A dispatch is identified by its kernel name
# AND its number of workgroups (= traces): icpc_lean3_kernel<NT, M, SEP, FULL> is the headline <.., false, true>, and the secondary
# dsp_icpc lines run <.., true, true> (CUSP / ZAC optimised separately), <.., false, false> twice (a trace shorter than the tile; a length
# that is no multiple of four samples — told apart by their batch sizes) and the generic icpc_kernel (a parameter set the lean kernel
# does not take)
lines = [("headline", bench["config"]["traces_per_gpu"], bench["config"]["samples"])]
for sec in bench.get("secondary", []):
    n_, L_ = sec["config"]["traces_per_gpu"], sec["config"]["samples"]
    wl = sec["config"]["workload"]
    if "pole-zero" in sec["metric"]: lines.append(("pz_trap", n_, L_))
    elif "dsp_sipm" in sec["metric"]: lines.append(("k_sipm", n_, L_))
    elif "separately" in wl: lines.append(("sep", n_, L_))
    elif "fallback" in wl: lines.append(("icpc_kernel<", n_, L_))
    else: lines.append(("short", n_, L_))


def shape_of(k, n):
    for kind, n_, L_ in lines:
        if n_ != n:
            continue
        if "icpc_lean3_kernel<" in k:
            targs = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
            sep, full = targs[2] == "true", (len(targs) < 4 or targs[3] == "true")
            if kind == ("sep" if sep else "headline" if full else "short"):
                return (n_, L_)
        elif kind not in ("headline", "sep", "short") and kind in k:
            return (n_, L_)
    return None


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))):
        if r["Counter_Name"] == ctr and r["Kernel_Name"].startswith("void ldsp::"):
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = (r["Kernel_Name"].replace("void ", "").split("(")[0], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
    for k, v in per_dispatch.items():
        acc[names[k]][ctr].append(v)
recs = []
for (k, n_disp), v in acc.items():
    shape = shape_of(k, n_disp)
    if shape is None or not v["FETCH_SIZE"] or not v["WRITE_SIZE"]:
        continue
    fa, wa = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
    recs.append({"command": "rocprofv3 --kernel-trace --pmc <COUNTER> -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0   (one pass per counter; tools/profile_round.sh)",
                 "n_traces": shape[0], "L": shape[1],
                 "units": "FETCH_SIZE / WRITE_SIZE in KB (1024 B) as reported; gfx950 correction: FETCH_SIZE x2 for 16 B/lane streaming loads (MI355X_MICROARCH.md, HBM section)",
                 "kernels": {k: {"FETCH_SIZE_KB_avg": fa, "launches": len(v["FETCH_SIZE"]), "WRITE_SIZE_KB_avg": wa,
                                 "hbm_bytes_per_launch_corrected": (2.0 * fa + wa) * 1024.0,
                                 "hbm_bytes_per_trace_corrected": (2.0 * fa + wa) * 1024.0 / shape[0]}}})
json.dump(recs, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
for r in recs:
    for k, v in r["kernels"].items():
        print(k, r["n_traces"], r["L"], f"{v['hbm_bytes_per_trace_corrected']:.0f} B/trace")
