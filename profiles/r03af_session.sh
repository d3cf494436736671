#!/bin/bash
# (1) polling variants of the one-launch carve, kernel-trace timings; (2) SQ counters of the per-iteration kernels
cd /root/repo; mkdir -p gpurun_out/r03af
export TMPDIR=/tmp
cd /tmp
for V in "1 0" "1 1" "0 0" "0 1"; do
  set -- $V
  WO_CARVE_FLOW_WATCH=$1 WO_CARVE_FLOW_SLEEP=$2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w$1s$2 -o t -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 0 --iters 40 > /root/repo/gpurun_out/r03af/w$1s$2.log 2>&1
  cp $(find /tmp/prof_w$1s$2 -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r03af/w$1s$2_kernel_stats.csv
  grep -E "k_carve_flow|k_ice_climb" /root/repo/gpurun_out/r03af/w$1s$2_kernel_stats.csv | cut -c1-120
  tail -1 /root/repo/gpurun_out/r03af/w$1s$2.log | grep -o '"parity": {[^}]*}'
done
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_sq1 -o pmc -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 0 --iters 20 > /root/repo/gpurun_out/r03af/pmc1.log 2>&1
python /root/repo/profiles/summarize_sq.py /tmp/pmc_sq1 > /root/repo/gpurun_out/r03af/sq_pass1.json
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM --output-format csv -d /tmp/pmc_sq2 -o pmc -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 0 --iters 20 > /root/repo/gpurun_out/r03af/pmc2.log 2>&1
python /root/repo/profiles/summarize_sq.py /tmp/pmc_sq2 > /root/repo/gpurun_out/r03af/sq_pass2.json
ls -la /root/repo/gpurun_out/r03af/
