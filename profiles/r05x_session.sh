#!/bin/bash
# Round 5, session x: (1) EXPERIMENT — one planet as two landmass shares on one GPU, the largest landmass apart, no flood exchange: does the rest of
# the planet iterating during the largest landmass's walk hide the host flood?  (2) does RCCL take two ranks on one device?
cd /root/repo; OUT=/root/repo/gpurun_out/r05x; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python research/ab/r05_two_shares_flood_overlap.py 10000000 3 > $OUT/two_shares_flood_overlap.txt 2>&1; echo "overlap rc=$?"
tail -6 $OUT/two_shares_flood_overlap.txt
NCCL_DEBUG=WARN timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 research/ab/r05_rccl_two_ranks_one_device.py > $OUT/rccl_two_ranks_one_device.txt 2>&1; echo "rccl rc=$?"
grep -v "^$" $OUT/rccl_two_ranks_one_device.txt | tail -12
