#!/bin/bash
# tools/pmc_sets.sh <tag> <filter> <"counter set 1" "counter set 2" ...> -- <command ...>
# One `rocprofv3 --pmc` pass per counter set (no trace domains) of an arbitrary command; per-kernel means -> gpurun_out/<tag>.json.
# A set with a counter this rocprofv3 does not know fails on its own and is skipped (its log stays in gpurun_out/<tag>_logs/).
set -u
TAG=$1; FILTER=$2; shift 2
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift
REPO=$(pwd)
OUT=/tmp/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" "$REPO/gpurun_out/${TAG}_logs"
export TMPDIR=/tmp
cd /tmp
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$i" -- "$@" > "$REPO/gpurun_out/${TAG}_logs/pmc_$i.log" 2>&1 || echo "set $i ($set) failed" >> "$REPO/gpurun_out/${TAG}_logs/failed.txt"
done
cd "$REPO"
python tools/pmc_table.py "$OUT" "$FILTER" > "$REPO/gpurun_out/$TAG.json"
rm -rf "$OUT"
