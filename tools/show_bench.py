"""Readable summary of a bench.py JSON line: python tools/show_bench.py gpurun_out/bench_TAG.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"].get("baseline_config"), "value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "median", d.get("median_ms_per_step"),
      "n_gpus", d["n_gpus"], "pipelined", d["config"].get("pipelined"))
r = d["roofline"]
if r:
    print("dominant", r["kernel"], r["achieved"], r["unit"], "frac", r["frac"], "avg launch ms", r["avg_launch_ms"], "| whole path",
          r["whole_path_tflops_per_gpu"], "TF | kernel time per step", r["kernel_time_ms_per_step"], "ms | traffic", r["traffic"])
    for k, v in r["classes"].items():
        extra = f"tflops={v['tflops']:7.2f} frac={v['frac']:.3f}" if "tflops" in v else (f"GB/s={v['gbs']:8.1f} frac={v['frac']:.4f}" if "gbs" in v else "")
        print(f"  {k:28s} n={v['launches_per_step']:7.2f} ms/step={v['ms_per_step']:.4f} {extra}")
if d.get("cpu_baseline"):
    print(d["cpu_baseline"])
