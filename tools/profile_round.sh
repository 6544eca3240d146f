#!/bin/bash
# Usage (on the GPU box): bash tools/profile_round.sh r01b
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
rm -f $O/*/p_agent_info.csv
ls -la $O $O/stats
