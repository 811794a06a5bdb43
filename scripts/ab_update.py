#!/usr/bin/env python
"""A/B of the update-kernel variants over (degree, beam, batch) on the 1M-row HBM working set."""
import json
import subprocess
import sys

rows = []
for deg in (10, 32):
    for beam in (1, 4):
        for batch in (1024, 8192):
            for v in (3, 4):
                r = subprocess.run([sys.executable, "scripts/kernel_bench.py", "--variant", str(v), "--beam", str(beam), "--deg", str(deg),
                                    "--batch", str(batch), "--reps", "2"], capture_output=True, text=True)
                try:
                    d = json.loads(r.stdout.strip().splitlines()[-1])["search"]
                    rows.append({"deg": deg, **{k: d[k] for k in ("beam", "batch", "variant", "update_GBps_algorithmic", "update_ms", "expand_ms", "evals_per_launch", "launches")}})
                    print(rows[-1], flush=True)
                except Exception as e:  # noqa: BLE001
                    print("ERR", deg, beam, batch, v, r.stderr[-300:], e, flush=True)
