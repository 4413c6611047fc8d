#!/bin/bash
# Runs every Python tool of this directory briefly against the current library (the tools/README.md "still run" list is
# what this script exercised in round 4).  On the GPU box, from the repo root: tools/check_tools.sh > gpurun_out/tools_check.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export PYTHONPATH=$PWD
ok=0; bad=0
run() {
  local name=$1; shift
  if timeout 600 "$@" > /tmp/tool_$$.log 2>&1; then echo "ok    $name: $(tail -1 /tmp/tool_$$.log | cut -c1-160)"; ok=$((ok+1));
  else echo "FAIL  $name (rc $?): $(tail -3 /tmp/tool_$$.log | tr '\n' ' ' | cut -c1-300)"; bad=$((bad+1)); fi
}
run fuzz_gpu        python tools/fuzz_gpu.py 150 4242
run fuzz_routes     python tools/fuzz_routes.py 12 7
run fuzz_gcn3       python tools/fuzz_gcn3.py 12 7
run dp_check        python tools/dp_check.py discrete 4
run rmat_check      python tools/rmat_check.py 14 64
run shard_step_time python tools/shard_step_time.py delta
run marks_ab        python tools/marks_ab.py
run stageb_lab      python tools/stageb_lab.py 16 128 512
run create_time     python tools/create_time.py
run dbg_full_sparse python tools/dbg_full_vs_sparse.py
run graph_try       python tools/graph_try.py
echo "tools: $ok ok, $bad failed"
rm -f /tmp/tool_$$.log
[ $bad -eq 0 ]
