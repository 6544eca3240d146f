"""Copies the judged summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/: the bench line (headline +
secondary results), rocprofv3 kernel statistics of the bench command and of the "next"-row timers, and the HBM-traffic records
built from the FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE x2 on gfx950 for 16 B/lane streaming loads, MI355X_MICROARCH.md HBM
section) — one record per workload of the bench command (bench.py looks them up by kernel name, batch size and trace length).
Usage: python tools/collect_profiles.py r02a"""
import csv, json, os, shutil, sys, collections
tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(R, "gpurun_out", tag), os.path.join(R, "profiles")
with open(os.path.join(src, "bench.json")) as f:
    line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
with open(os.path.join(dst, f"{tag}_bench.json"), "w") as f:
    f.write(line + "\n")
bench = json.loads(line)
shutil.copy(os.path.join(src, "stats", "p_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
# the same run's kernel trace without each kernel's FIRST launch (cold instruction cache / clocks): what the bench's timed steps see
kt = os.path.join(src, "stats", "p_kernel_trace.csv")
if os.path.exists(kt):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        if r["Kernel_Name"].startswith("void ldsp::"):
            per[r["Kernel_Name"].replace("void ", "").split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    with open(os.path.join(dst, f"{tag}_bench_kernel_stats_warm.csv"), "w") as g:
        g.write("kernel,launches,first_launch_ns,warm_launches,warm_avg_ns,warm_min_ns,warm_max_ns\n")
        for k, v in sorted(per.items()):
            v.sort()
            d = [x[1] for x in v]
            w = d[1:] if len(d) > 1 else d
            g.write(f'"{k}",{len(d)},{d[0]},{len(w)},{sum(w) / len(w):.0f},{min(w)},{max(w)}\n')
for t in ("gpu_time_grid", "gpu_time_compressed", "gpu_time_multi_intersect"):
    ks = os.path.join(src, f"stats_{t}", "p_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(dst, f"{tag}_{t[9:]}_kernel_stats.csv"))
    lg = os.path.join(src, f"{t}.log")
    if os.path.exists(lg):
        with open(lg) as f, open(os.path.join(dst, f"{tag}_{t[9:]}_timing.txt"), "w") as g:
            g.write("".join(l for l in f if "amdgpu.ids" not in l and not l.startswith("[rocprofv3]") and "rocprofiler" not in l))
# workloads of the bench command: which (n_traces, L) a kernel of the PMC passes ran on.  icpc_lean3_kernel<NT, M, SEP, FULL>: the
# headline is <.., false, true>; the secondary dsp_icpc lines run <.., true, true> (CUSP / ZAC optimised separately), <.., false, false>
# (a trace shorter than the tile) and the generic icpc_kernel (a length the lean kernel does not take)
shapes = {"headline": (bench["config"]["traces_per_gpu"], bench["config"]["samples"])}
for sec in bench.get("secondary", []):
    shp = (sec["config"]["traces_per_gpu"], sec["config"]["samples"])
    if "pole-zero" in sec["metric"]: shapes["pz_trap"] = shp
    elif "dsp_sipm" in sec["metric"]: shapes["k_sipm"] = shp
    elif "separately" in sec["config"]["workload"]: shapes["sep"] = shp
    elif "fallback" in sec["config"]["workload"]: shapes["icpc_kernel<"] = shp
    elif "shorter than the tile" in sec["config"]["workload"]: shapes["short"] = shp


def shape_of(k):
    if "icpc_lean3_kernel<" in k:
        targs = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
        sep, full = targs[2] == "true", (len(targs) < 4 or targs[3] == "true")
        return shapes.get("sep") if sep else shapes.get("headline") if full else shapes.get("short")
    return next((sh for frag, sh in shapes.items() if frag in k and frag not in ("headline", "sep", "short")), None)


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))):
        if r["Counter_Name"] == ctr and r["Kernel_Name"].startswith("void ldsp::"):
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for k, v in per_dispatch.items():
        acc[names[k].replace("void ", "").split("(")[0]][ctr].append(v)
recs = []
for k, v in acc.items():
    shape = shape_of(k)
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
