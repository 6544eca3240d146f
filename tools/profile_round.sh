#!/bin/bash
# Usage (on the GPU box): bash tools/profile_round.sh r01d
# bench line + rocprofv3 kernel stats + HBM traffic counters of the same bench command -> gpurun_out/<tag>/
set -eo pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json
rocprofv3 --kernel-trace --stats -d $O/stats -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/pmc_write.log 2>&1
# the second routine of the path: dsp_sipm, BASELINE config 5 shape (625 k x 16384 per GPU)
python3 $R/bench.py --workload sipm > $O/bench_sipm.json 2> $O/bench_sipm.err
tail -1 $O/bench_sipm.json
rocprofv3 --kernel-trace --stats -d $O/stats_sipm -o p --output-format csv -- python3 $R/bench.py --workload sipm --steps 3 --warmup 1 > $O/stats_sipm.log 2>&1
python3 $R/bench.py --workload pz_trap > $O/bench_pz_trap.json 2> $O/bench_pz_trap.err
tail -1 $O/bench_pz_trap.json
rm -f $O/*/p_agent_info.csv
ls $O
