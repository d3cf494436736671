#!/bin/bash
# Round 6, session h: the thermal step's one-launch form (k_thermal_apply_reg<.., true>, chosen on the device per iteration): parity subset, then A/B of the timed region —
# base (the tree before it), default (adaptive, 6 waves for the fused kernel), the form pinned to 0 / 1, fused kernel at 4 / 5 waves.
cd /root/repo; OUT=/root/repo/gpurun_out/r06h; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "thermal_forms or golden or config3_checksum" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.log
for rep in 1 2; do
  for v in base default form1 tf6 tf4; do
    unset WO_LIBWOROGEN WO_TEST_HOOKS
    case $v in
      base) export WO_LIBWOROGEN=/root/repo/research/ab/variants/libworogen_base.so;;
      form0) export WO_TEST_HOOKS=thermal_form=0;;
      form1) export WO_TEST_HOOKS=thermal_form=1;;
      tf6|tf4) export WO_LIBWOROGEN=/root/repo/research/ab/variants/libworogen_$v.so;;
    esac
    python bench.py --timed-only --steps 8 --warmup 2 > $OUT/ab_${v}_$rep.json 2> $OUT/ab_${v}_$rep.err
  done
done
unset WO_LIBWOROGEN WO_TEST_HOOKS
python - <<'PY'
import json,glob
for v in ("base","default","form1","tf6","tf4"):
    rows=[]
    for f in sorted(glob.glob(f"/root/repo/gpurun_out/r06h/ab_{v}_*.json")):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); rows.append((round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"], round(d["stage_ms_last_step"]["thermal"],1), d["erode_stats"].get("thermal_fused_iterations")))
        except Exception as e:
            rows.append(("failed", str(e)[:80]))
    print(v, rows)
PY
