#!/bin/bash
# Round 6, session s: planets in flight on one GPU when the host flood dominates (config 5's usual case): 6 other terrains of the bench mesh, sequential against 3 and 6 in flight.
cd /root/repo; OUT=/root/repo/gpurun_out/r06s; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python profiles/in_flight_flood_heavy_probe.py 6 > $OUT/in_flight_flood_heavy.txt 2> $OUT/in_flight.err; echo "rc=$?"; cat $OUT/in_flight_flood_heavy.txt; tail -2 $OUT/in_flight.err
