# Dense visits (k_solve_patch_dense) for the first launches of a solve pass: how many launches should use them?
O=gpurun_out/r02h; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 1 > $O/$name.log 2>&1; python - $O/$name.log $name <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); st=d['stage_ms_last_step']; es=d['erode_stats']; fam=d['roofline']['families']
        print(sys.argv[2], 'ms/step %.0f'%d['ms_per_step'], 'crc_ok', d['parity']['parity_crc_ok'], 'solve_stage %.0f'%st['solve'], 'launches', int(es['solve_patch_launches_total']), 'solve_patch_ms %.0f'%fam['solve_patch']['ms'])
P
}
run dense0 WO_SOLVE_DENSE=0
run dense4 WO_SOLVE_DENSE=4
run dense10 WO_SOLVE_DENSE=10
run dense20 WO_SOLVE_DENSE=20
run dense1000 WO_SOLVE_DENSE=1000
run dense10_s32 WO_SOLVE_DENSE=10 WO_SOLVE_SPINS=32
