#!/bin/bash
# The one GPU-side runner (sent to the MI355X box with `gpurun -- 'bash tools/gpu/run.sh <recipe> [args]'`); every recipe writes
# under gpurun_out/<tag>/ -- copy what should be judged into profiles/.
#
#   suite                      the whole `-m gpu` test suite
#   bench [bench.py args]      the driver-style bench line (default flags)
#   rocprof <name> [args]      rocprofv3 --kernel-trace --stats of `bench.py <args>` -> <name>_kernel_stats.csv
#   pmc <name> <counters> -- <python script + args>   one rocprofv3 --pmc pass (counters: space separated, own run, no tracing)
#   t3 [args]                  tests/t3_baseline_shapes.py (needs tools/stage_reference.sh first: the REAL reference as the oracle)
#   py <script> [args]         any probe under tools/ or tools/gpu/
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
TAG=${AR_TAG:-r04}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$O"
recipe=$1; shift
case "$recipe" in
  suite)
    timeout ${AR_TIMEOUT:-1500} python -m pytest tests -q -m gpu "$@" > "$O/gpu_suite.log" 2>&1; echo "suite rc=$?"; tail -5 "$O/gpu_suite.log" ;;
  bench)
    timeout ${AR_TIMEOUT:-1500} python bench.py "$@" > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"; tail -c 400 "$O/bench.err"; tail -c 3000 "$O/bench.json" ;;
  rocprof)
    name=$1; shift
    cd /tmp && rm -rf /tmp/prof_$name
    timeout ${AR_TIMEOUT:-900} rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- python "$GRAFT_REPO_ROOT/bench.py" "$@" > "$O/${name}_bench_under_rocprof.json" 2> "$O/${name}_rocprof.err"; echo "rocprof rc=$?"
    python "$GRAFT_REPO_ROOT/tools/rocprof_summary.py" "$(find /tmp/prof_$name -name '*.db' | head -1)" --stats "$O/${name}_kernel_stats.csv"
    head -12 "$O/${name}_kernel_stats.csv" | cut -c1-170 ;;
  pmc)
    name=$1; shift; counters=$1; shift; [ "$1" = "--" ] && shift
    cd /tmp && rm -rf /tmp/pmc_$name
    timeout ${AR_TIMEOUT:-600} rocprofv3 --pmc $counters -d /tmp/pmc_$name --output-format csv -- python "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$O/${name}_pmc.log" 2>&1; echo "pmc rc=$?"
    find /tmp/pmc_$name -name '*counter_collection.csv' -exec cp {} "$O/${name}_counter_collection.csv" \; ; ls -la "$O" | tail -3 ;;
  t3)
    timeout ${AR_TIMEOUT:-1500} python tests/t3_baseline_shapes.py "$@" > "$O/t3.log" 2>&1; echo "t3 rc=$?"; grep -v "amdgpu.ids\|layer_idx" "$O/t3.log" | tail -40 ;;
  py)
    script=$1; shift
    timeout ${AR_TIMEOUT:-900} python "$script" "$@" > "$O/$(basename "$script" .py).log" 2>&1; echo "rc=$?"; grep -v "amdgpu.ids\|layer_idx" "$O/$(basename "$script" .py).log" | tail -40 ;;
  *) echo "unknown recipe $recipe"; exit 2 ;;
esac
