import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "n_gpus", d["n_gpus"])
r = d["roofline"]
print("dominant", r["kernel"], r["achieved"], "TF; whole path", r["whole_path_tflops_per_gpu"], "TF")
for k, v in r["kernels"].items():
    print(f"  {k:24s} n={v['launches_per_step']:3d} avg_ms={v['avg_ms']:.4f} tot_ms={v['avg_ms']*v['launches_per_step']:.3f} tflops={v['tflops']:7.2f} share={v['share']:.3f}")
if d.get("cpu_baseline"): print(d["cpu_baseline"])
