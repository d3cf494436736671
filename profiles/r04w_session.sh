#!/bin/bash
# Round 4, final measurements of the default bench command: rocprofv3 kernel stats, PMC FETCH/WRITE per kernel (200 iterations), bench line
cd /root/repo; OUT=/root/repo/gpurun_out/r04w; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_default_bench_command.csv
cd /root/repo
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o pmc -- python bench.py --no-cpu --no-profile --no-relaxed --in-flight 0 --steps 1 --warmup 0 --iters 200 > "$OUT/$C.log" 2>&1
    echo "$C rc=$?"
done
python profiles/summarize_pmc.py "$OUT" 200 > "$OUT/pmc_summary.json"
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
ls -la $OUT; head -c 600 $OUT/pmc_summary.json
