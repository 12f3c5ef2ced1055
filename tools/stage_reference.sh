#!/bin/bash
# Stage the reference tree for builder-side `gpurun` calls (T3: the REAL reference running on the MI355X as the oracle).
#
#   tools/stage_reference.sh [/root/reference]
#
# Copies the reference's two Python packages and the `cpuinfo` import shim into oracle/_ref/, which is git-ignored (the
# reference's sources never enter this repository's history) but NOT gpurun-ignored, so the copy travels to the GPU box
# with the snapshot.  tests/ref_tree.py finds /root/reference (build container) or oracle/_ref (GPU box); the T3 tests
# (`tests/test_gpu_t3_reference.py`) skip when neither exists -- e.g. in the driver's round-end run.
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${1:-/root/reference}
DST="$ROOT/oracle/_ref"
[ -d "$SRC/auto_round" ] || { echo "no reference tree at $SRC" >&2; exit 1; }
rm -rf "$DST/auto_round" "$DST/auto_round_extension"
mkdir -p "$DST"
cp -r "$SRC/auto_round" "$SRC/auto_round_extension" "$DST/"
cp "$ROOT/oracle/ref_shim/cpuinfo.py" "$DST/"
find "$DST" -name __pycache__ -type d -prune -exec rm -rf {} +
du -sh "$DST" | sed 's/^/[stage_reference] /'
