import json,sys
for c in (2,4,5,6):
    d=json.loads(open(f"gpurun_out/bench_c{c}.json").read().strip().splitlines()[-1])
    print(c, d["ms_per_step"], d["value"], d.get("step_device_ms"), d["steps"], d["warmup"])
