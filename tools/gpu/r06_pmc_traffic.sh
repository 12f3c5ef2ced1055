#!/bin/bash
# HBM traffic of K1 / K2 from the PMC counters, as MI355X_MICROARCH.md prescribes: two SEPARATE passes (FETCH_SIZE, WRITE_SIZE), kernel
# trace + counters only, over tools/pmc_traffic_probe.py; merged into gpurun_out/r06/pmc_traffic.json (-> profiles/r06_pmc_traffic.json).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p "$O"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -- python "$GRAFT_REPO_ROOT/tools/pmc_traffic_probe.py" > "$O/pmc_$c.log" 2>&1; echo "pmc $c rc=$?"
  python "$GRAFT_REPO_ROOT/tools/rocprof_summary.py" "$(find /tmp/pmc_$c -name '*.db' | head -1)" --pmc-rows "$O/pmc_rows_$c.csv"
done
python "$GRAFT_REPO_ROOT/tools/pmc_traffic_merge.py" "$O/pmc_rows_FETCH_SIZE.csv" "$O/pmc_rows_WRITE_SIZE.csv" 5 > "$O/pmc_traffic.json"; echo "merge rc=$?"; head -c 1500 "$O/pmc_traffic.json"
