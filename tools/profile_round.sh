#!/bin/bash
# Usage (on the GPU box): bash tools/profile_round.sh r02a
# The default bench line (headline + secondary: config 2, dsp_sipm), rocprofv3 kernel statistics and the HBM traffic counters of the
# same bench command, then kernel statistics of the "next" rows (grid scans, qc features, compressed routines, MultiIntersect)
# -> gpurun_out/<tag>/ ; tools/collect_profiles.py <tag> copies the judged summaries into profiles/.
set -eo pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-300
rocprofv3 --kernel-trace --stats -d $O/stats -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --cpu-sample 0 > $O/stats.log 2>&1
echo "stats done"
# HBM traffic: one pass per counter (MI355X_MICROARCH.md, HBM / rocprofv3 section), nothing else traced
# (the synthetic input is generated in a few large passes on the GPU, legenddsp_jl_amd/synth.py: the counters serialise every kernel)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/pmc_write.log 2>&1
echo "pmc done"
for t in gpu_time_grid gpu_time_compressed gpu_time_multi_intersect; do
  rocprofv3 --kernel-trace --stats -d $O/stats_$t -o p --output-format csv -- python3 $R/tools/$t.py > $O/$t.log 2>&1 || echo "$t failed"
  grep -v amdgpu $O/$t.log | tail -12
done
rm -f $O/*/p_agent_info.csv
# keep the rows of the library's own kernels only (the traces also list every torch kernel of the input generation)
for f in $O/*/p_kernel_trace.csv $O/*/p_counter_collection.csv; do
  [ -f "$f" ] && { head -1 "$f" > "$f.tmp"; grep "ldsp::" "$f" >> "$f.tmp" || true; mv "$f.tmp" "$f"; }
done
du -sh $O
ls $O
