# OBB bench lines at HEAD (both phases): bash tools/obb_now.sh
for w in step1 step2; do timeout -k 10 300 python bench.py --variant obb --workload $w --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/now_obb_$w.json 2>/dev/null; done
python - <<PY
import json
for w in ("step1", "step2"):
    d = json.load(open("gpurun_out/now_obb_%s.json" % w))
    print(w, d["ms_per_step"], d["value"], {k: v for k, v in d["custom_kernels_ms_per_step"].items() if v > 0.4})
PY
