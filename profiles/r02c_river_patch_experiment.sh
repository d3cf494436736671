# River-aligned solve patches (csrc/river.hip): launches and time vs the refresh period and the polling cap of a visit.
O=gpurun_out/r02c; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 1 > $O/$name.log 2>&1; python - $O/$name.log $name <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); st=d['stage_ms_last_step']; es=d['erode_stats']; fam=d['roofline']['families']
        print(sys.argv[2], 'ms/step %.0f'%d['ms_per_step'], 'crc', d['parity']['parity_crc_ok'], 'solve %.0f'%st['solve'], 'launches', es['solve_patch_launches_total'], 'patch_ms %.0f'%fam['solve_patch']['ms'], 'river_ms %.1f (%d)'%(fam.get('river_order',{}).get('ms',0), fam.get('river_order',{}).get('launches',0)), 'setup %.0f final %.0f'%(fam['solve_setup']['ms'], fam['solve_final']['ms']))
P
}
run morton_s16 WO_RIVER_PATCHES=0
run river4_s16 WO_RIVER_PATCHES=4
run river4_s64 WO_RIVER_PATCHES=4 WO_SOLVE_SPINS=64
run river4_s256 WO_RIVER_PATCHES=4 WO_SOLVE_SPINS=256
run river4_s4096 WO_RIVER_PATCHES=4 WO_SOLVE_SPINS=4096
run river1_s256 WO_RIVER_PATCHES=1 WO_SOLVE_SPINS=256
run river2_s256 WO_RIVER_PATCHES=2 WO_SOLVE_SPINS=256
run river8_s256 WO_RIVER_PATCHES=8 WO_SOLVE_SPINS=256
run river16_s256 WO_RIVER_PATCHES=16 WO_SOLVE_SPINS=256
