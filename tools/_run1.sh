set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for v in 1 0; do
  SRX_WGRAD_WALK=$v timeout -k 10 400 python bench.py --no-cpu-baseline --no-profile --steps 30 2>/dev/null > gpurun_out/bench_walk$v.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_walk$v.json").read().strip().splitlines()[-1])
print("walk=$v", d["ms_per_step"], {k:(v["ms_per_step"], v["pct_bf16_mfma_peak"]) for k,v in d.get("extras",{}).items()})
PY
done
