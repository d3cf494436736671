#!/bin/bash
# TA / TCP counters of the per-iteration kernels (two passes)
cd /root/repo; mkdir -p gpurun_out/r03as
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --pmc TA_TA_BUSY TA_TOTAL_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_READ_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_ta -o pmc -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 0 --iters 20 > /root/repo/gpurun_out/r03as/pmc_ta.log 2>&1
python /root/repo/profiles/summarize_sq.py /tmp/pmc_ta > /root/repo/gpurun_out/r03as/ta.json
timeout 900 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_UTCL1_TRANSLATION_MISS --output-format csv -d /tmp/pmc_tcp -o pmc -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 0 --iters 20 > /root/repo/gpurun_out/r03as/pmc_tcp.log 2>&1
python /root/repo/profiles/summarize_sq.py /tmp/pmc_tcp > /root/repo/gpurun_out/r03as/tcp.json
tail -2 /root/repo/gpurun_out/r03as/pmc_tcp.log | cut -c1-200
