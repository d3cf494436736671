#!/bin/bash
# Round 6, session q: the flood's static tables with the landmass lists built in parallel (no zero-filled temporaries): new-terrain probe (set-up laps), then the tests that lean on those tables.
cd /root/repo; OUT=/root/repo/gpurun_out/r06q; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python profiles/new_terrain_probe.py > $OUT/new_terrain_probe.txt 2> $OUT/new_terrain_probe.err; echo "probe rc=$?"
cat $OUT/new_terrain_probe.txt | cut -c1-220; grep "flood static" $OUT/new_terrain_probe.err | head -20
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flood or config3_checksum or config4_size or decomposed or every_eighth or land_count or edge_cases" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_subset.log
