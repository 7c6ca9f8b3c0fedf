# Sourced by the tools/pmc_*.sh scripts: one rocprofv3 --pmc pass with a fast fail.
#   pmc_pass <out_dir> <name> <counter> [<counter> ...] -- <program> [args ...]
# Round 2 lost ~10 GPU-minutes to three passes that had aborted at the FIRST kernel launch ("error code 38: Request exceeds the
# capabilities of the hardware to collect" -> rocprofv3 caught signal 6) and then sat until `timeout 200` killed them: a --pmc list
# may hold at most what ONE hardware block offers per pass (gfx950: SQ 8, TCC 4 with FETCH_SIZE = 3 and WRITE_SIZE = 2, TCP / TA /
# TD 2 -- the lists that succeeded in profiles/r02_* had <= 2 of those).  So: refuse over-long lists before starting, and while the
# pass runs watch its log for the abort and kill the profiler (by PID) at once instead of waiting the timeout out.
PMC_LIMITS="SQ:8 TCC:4 TCP:2 TA:2 TD:2 GRBM:2"
pmc_check_list() {
  local -A n=()
  for c in "$@"; do
    local blk=${c%%_*} w=1
    [ "$c" = FETCH_SIZE ] && { blk=TCC; w=3; }
    [ "$c" = WRITE_SIZE ] && { blk=TCC; w=2; }
    n[$blk]=$(( ${n[$blk]:-0} + w ))
  done
  for lim in $PMC_LIMITS; do
    local blk=${lim%%:*} max=${lim##*:}
    if [ "${n[$blk]:-0}" -gt "$max" ]; then echo "pmc: $blk list needs ${n[$blk]} slots, one pass holds $max: split it ($*)" >&2; return 1; fi
  done
}
pmc_pass() {
  local out=$1 name=$2; shift 2
  local ctrs=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  pmc_check_list "${ctrs[@]}" || { echo "pass $name refused"; return 1; }
  mkdir -p "$out"
  local log=$out/$name.log
  # the program itself goes after `--` (no env / bash -c hop: the profiler's preloaded library has initialised the GPU by then)
  timeout -k 10 ${PMC_TIMEOUT:-200} rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d "$out/$name" -- "$@" > "$log" 2>&1 &
  local pid=$!
  while kill -0 $pid 2>/dev/null; do
    if grep -q -E "error code 38|caught signal 6|tool.cpp:1198" "$log" 2>/dev/null; then
      echo "pass $name: rocprofv3 aborted (see $log): killing it"; kill $pid 2>/dev/null; sleep 2; kill -9 $pid 2>/dev/null
      wait $pid 2>/dev/null; return 1
    fi
    sleep 2
  done
  wait $pid || { echo "pass $name failed ($(tail -1 "$log"))"; return 1; }
}
