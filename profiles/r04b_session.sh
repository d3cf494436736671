#!/bin/bash
# Round 4, session b: SQ / TA / TCP / TCC counters of the per-iteration kernels (one planet, 10 M cells, 20 iterations), one
# rocprofv3 --pmc pass per counter group, no trace domains (name the limiter of the streaming kernels).
cd /root/repo; OUT=/root/repo/gpurun_out/r04b; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counter_list.txt 2>&1
grep -o "TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TD_[A-Z0-9_]*" $OUT/counter_list.txt | sort -u > $OUT/ta_tcp_names.txt
pass() {  # name counters...
  local NAME=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$NAME -o pmc -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 0 --iters 20 > $OUT/$NAME.log 2>&1
  echo "$NAME rc=$?"
  python /root/repo/profiles/summarize_sq.py /tmp/pmc_$NAME > $OUT/$NAME.json
  rm -rf /tmp/pmc_$NAME
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum
pass tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum
pass ta1 TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass tcc2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_ATOMIC_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
# kernel trace of the same command for durations
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 0 --iters 20 > $OUT/kt.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_20iters.csv
ls -la $OUT
