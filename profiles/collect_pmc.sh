#!/bin/bash
# Per-kernel HBM traffic from rocprofv3 PMC counters, on the GPU box:  bash profiles/collect_pmc.sh <tag> [iters]
# FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC counter budget), so each gets its own run of the same
# workload (bench.py at 10M cells, <iters> composite iterations, no CPU leg).  Counter runs use no trace domains.
set -u
TAG=${1:-r01}
ITERS=${2:-6}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd "$ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o pmc -- \
        python bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 0 --iters "$ITERS" > "$OUT/$C.log" 2>&1
    echo "$C rc=$?"
done
python profiles/summarize_pmc.py "$OUT" "$ITERS" > "$OUT/summary.json" && cp "$OUT/summary.json" "$ROOT/gpurun_out/pmc_${TAG}_summary.json"
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
