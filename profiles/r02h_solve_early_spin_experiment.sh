O=gpurun_out/r02h; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 1 > $O/$name.log 2>&1; python - $O/$name.log $name <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); st=d['stage_ms_last_step']; es=d['erode_stats']; fam=d['roofline']['families']
        print(sys.argv[2], 'ms/step %.0f'%d['ms_per_step'], 'crc', d['parity']['parity_crc_ok'], 'solve %.0f'%st['solve'], 'launches', es['solve_patch_launches_total'], 'patch_ms %.0f'%fam['solve_patch']['ms'])
P
}
run base WO_X=1
run early4_32 WO_SOLVE_EARLY=4 WO_SOLVE_EARLY_SPINS=32
run early4_64 WO_SOLVE_EARLY=4 WO_SOLVE_EARLY_SPINS=64
run early8_32 WO_SOLVE_EARLY=8 WO_SOLVE_EARLY_SPINS=32
run early8_64 WO_SOLVE_EARLY=8 WO_SOLVE_EARLY_SPINS=64
run early2_128 WO_SOLVE_EARLY=2 WO_SOLVE_EARLY_SPINS=128
run early12_24 WO_SOLVE_EARLY=12 WO_SOLVE_EARLY_SPINS=24
