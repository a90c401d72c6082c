set -o pipefail
mkdir -p gpurun_out
for m in 1 0 2; do
KATOME_SORTED_TILES=$m timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 3 > gpurun_out/st_$m.json 2> gpurun_out/st_$m.err; echo "mode $m rc=$?"
done
python - <<'PY'
import json
for n in ("1","0","2"):
    try:
        d=json.loads(open("gpurun_out/st_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], {k:round(v["ms_per_step"],1) for k,v in d["kernels"].items() if not k.startswith("k:")} if "kernels" in d else "")
        print({k:(round(v["ms_per_step"],1), v["launches_per_step"], round(v.get("frac_of_hbm_peak",0),2)) for k,v in d.get("kernel_launches",{}).items()})
        print(d["roofline"]["kernel"], d["roofline"]["frac"])
    except Exception as e:
        print(n, "ERR", e); print(open("gpurun_out/st_%s.err"%n).read()[-1500:])
PY
timeout -k 10 900 python -m pytest tests/test_gpu_build.py -x -q -m gpu > gpurun_out/st_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/st_tests.log
