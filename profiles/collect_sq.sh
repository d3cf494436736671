#!/bin/bash
# Per-kernel SQ / TCC counters from rocprofv3 (separate passes: 8 SQ slots, 4 TCC slots per pass), on the GPU box:
#   bash profiles/collect_sq.sh <tag> [iters]
# Workload: bench.py at 10M cells, <iters> composite iterations, no CPU leg, no HIP-event profiling.  No trace domains.
set -u
TAG=${1:-r02}
ITERS=${2:-6}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p "$OUT"
cd "$ROOT"
run() {  # name, counters...
    local NAME=$1; shift
    timeout 900 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$NAME" -o pmc -- \
        python bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 0 --iters "$ITERS" > "$OUT/$NAME.log" 2>&1
    echo "$NAME rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run sq2 SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python profiles/summarize_counters.py "$OUT" "$ITERS" > "$ROOT/gpurun_out/sq_${TAG}_summary.json"
find "$OUT" -name "*.csv" -delete; find "$OUT" -name "*.db" -delete
