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
for t in ("gpu_time_grid", "gpu_time_compressed", "gpu_time_multi_intersect"):
    ks = os.path.join(src, f"stats_{t}", "p_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(dst, f"{tag}_{t[9:]}_kernel_stats.csv"))
    lg = os.path.join(src, f"{t}.log")
    if os.path.exists(lg):
        with open(lg) as f, open(os.path.join(dst, f"{tag}_{t[9:]}_timing.txt"), "w") as g:
            g.write("".join(l for l in f if "amdgpu.ids" not in l and not l.startswith("[rocprofv3]") and "rocprofiler" not in l))
# workloads of the bench command: kernel-name fragment -> (n_traces, L)
shapes = {"icpc": (bench["config"]["traces_per_gpu"], bench["config"]["samples"])}
for sec in bench.get("secondary", []):
    key = "pz_trap" if "pole-zero" in sec["metric"] else "k_sipm"
    shapes[key] = (sec["config"]["traces_per_gpu"], sec["config"]["samples"])
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
    shape = next((s for frag, s in shapes.items() if frag in k), None)
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
