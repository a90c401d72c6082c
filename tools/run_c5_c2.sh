python bench.py --workload c5 --reads 100000000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_c5_100m.json; python bench.py --workload c2 --steps 5 --warmup 2 2>/dev/null > gpurun_out/bench_c2.json; python - <<'PY'
import json
for f in ("bench_c5_100m","bench_c2"):
    d=json.load(open("gpurun_out/%s.json"%f))
    ks={k:(round(v["ms_per_step"],1)) for k,v in d["kernels"].items()}
    print(f, "value=%.3g ms/step=%.1f edges=%d nodes=%d span=%s"%(d["value"],d["ms_per_step"],d["distinct_edges"],d["nodes"],d["config"]["tile_span"]), ks, d.get("cpu_baseline",{}).get("value"))
PY
