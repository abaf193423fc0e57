#!/bin/bash
# GPU box: parity suite, then the bench line and its roofline objects in short form
set -o pipefail
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
timeout -k 10 300 python bench.py --cpu-seconds 0 > gpurun_out/bench_q.json || exit 1
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_q.json"))
print(d["value"], d["single_clone_stages_ms"])
for k in d:
    if k.startswith("roofline"):
        print(" ", k, d[k]["us_per_launch"], d[k]["frac"])
PY
