set -o pipefail
# (A/B of builds made at different times: the staleness check of legenddsp_jl_amd._lib is waived explicitly)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "$@"; do
  LDSP_ALLOW_STALE=1 LDSP_HIP_LIB=$(readlink -f $lib) timeout -k 10 120 python3 $R/tools/lean_time.py 262144 6 2>&1 | tail -1
done
