"""one line per bench JSON: tokens/s, ms/step, GEMM rate and the per-family ms/step."""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print(f, round(d["value"]), "tok/s", round(d["ms_per_step"], 2), "ms/step", "gemm", round(d["roofline"]["achieved"]), "TF/s",
          {k: round(v["ms_per_step"], 2) for k, v in d["kernels"].items()})
