#!/bin/bash
# Round 5, session au: where the GPU sits idle inside a step (kernel trace of the timed region, last step): every idle stretch >= 50 us with the kernels on either side.
cd /root/repo; OUT=/root/repo/gpurun_out/r05au; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp; rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 3 --warmup 2 > $OUT/bench_under_trace.json 2> $OUT/kt.err
cd /root/repo
python profiles/step_idle_gaps.py /tmp/kt 50 > $OUT/step_idle_gaps.txt 2>&1
cat $OUT/step_idle_gaps.txt | cut -c1-200 | head -60
