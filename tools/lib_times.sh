set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "$@"; do
  LDSP_HIP_LIB=$(readlink -f $lib) timeout -k 10 120 python3 $R/tools/lean_time.py 262144 6 2>&1 | tail -1
done
