"""Prints the single-model frame-batch leg of a bench.py JSON line (default gpurun_out/bench_pb.json)."""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_pb.json"))
print("headline", d["value"], "GN it/s", d["ms_per_step"], "ms/step")
p = d["tracking_batch"]
print("pose batch", p["value"], "GN it/s", p["ms_per_step"], "ms/step", p["frames_per_s"], "frames/s")
r = p["roofline"]
print("dominant", r["kernel"], r["frac"], r["avg_launch_us"])
print(r["per_step_us_events"]); print(r["launches_per_step"])
for k, v in r["per_kernel"].items():
    print("%-26s frac %.4f  %8.2f us/launch  %8.1f GB/s  rocprof %s" % (k, v["frac"], v["avg_launch_us"], v["achieved"], v["rocprof"] and v["rocprof"]["median_us"]))
print(r["whole_step"])
