"""Pretty-print a bench.py JSON line: python tools/show_bench.py file.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
steps = d.get("kernels_steps", d["steps"])
print(f"{d['value']}x RT   {d['ms_per_step']} ms/step   whole path {d.get('whole_path_tflops')} TFLOP/s   roofline {d['roofline']}")
tot = 0.0
for k in d["kernels"]:
    ms = k["ms"] / steps
    tot += ms
    print(f"{k['name']:40s} {k['launches'] // steps:5d} {ms:8.2f} ms/step {k['tflops']:7.1f} TF {k['gbps']:8.1f} GB/s")
print(f"instrumented {tot:.1f} ms/step of {d['ms_per_step']}")
for key in ("pcie_inclusive_value", "cpu_baseline"):
    if key in d:
        print(key, d[key])
