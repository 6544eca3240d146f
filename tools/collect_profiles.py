"""Copies the judged summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/:
bench lines, rocprofv3 kernel statistics, and the HBM-traffic record built from the FETCH_SIZE / WRITE_SIZE passes
(FETCH_SIZE x2 on gfx950 for 16 B/lane streaming loads, MI355X_MICROARCH.md HBM section).
Usage: python tools/collect_profiles.py r01f"""
import csv, json, os, shutil, sys, collections
tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(R, "gpurun_out", tag), os.path.join(R, "profiles")
for a, b in (("bench.json", "bench.json"), ("bench_sipm.json", "bench_sipm.json"), ("bench_pz_trap.json", "bench_pz_trap.json")):
    with open(os.path.join(src, a)) as f:
        line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
    with open(os.path.join(dst, f"{tag}_{b}"), "w") as f:
        f.write(line + "\n")
shutil.copy(os.path.join(src, "stats", "p_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(src, "stats_sipm", "p_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_sipm_kernel_stats.csv"))
bench = json.loads(open(os.path.join(dst, f"{tag}_bench.json")).read())
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))):
        if r["Counter_Name"] == ctr and r["Kernel_Name"].startswith("void ldsp::"):
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for k, v in per_dispatch.items():
        acc[names[k].replace("void ", "").split("(")[0]][ctr].append(v)
rec = {"command": "rocprofv3 --kernel-trace --pmc <COUNTER> -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0   (one pass per counter; tools/profile_round.sh)",
       "n_traces": bench["config"]["traces_per_gpu"], "L": bench["config"]["samples"],
       "units": "FETCH_SIZE / WRITE_SIZE in KB (1024 B) as reported; gfx950 correction: FETCH_SIZE x2 for 16 B/lane streaming loads (MI355X_MICROARCH.md, HBM section)",
       "kernels": {}}
for k, v in acc.items():
    fa, wa = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
    rec["kernels"][k] = {"FETCH_SIZE_KB_avg": fa, "launches": len(v["FETCH_SIZE"]), "WRITE_SIZE_KB_avg": wa,
                         "hbm_bytes_per_launch_corrected": (2.0 * fa + wa) * 1024.0}
json.dump(rec, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(rec["kernels"], indent=1))
