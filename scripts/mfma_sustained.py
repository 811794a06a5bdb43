#!/usr/bin/env python
"""scripts/mfma_sustained.cpp, one setting per process, with bench.py's clock / power sampler alongside: the matrix-pipe rate this MI355X SUSTAINS under its
power cap with nothing else in the way (register-only v_mfma_f32_32x32x16_f16 on every SIMD) -- the ceiling the cap leaves under the 2.5 PFLOP/s figure the
roofline rows are quoted against.  One JSON line per setting.   python scripts/mfma_sustained.py [seconds per setting]"""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench as B

exe = ROOT / "leann_amd" / "lib" / "bin" / "mfma_sustained"
if not exe.exists():
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", str(ROOT / "scripts" / "mfma_sustained.cpp"), "-o", str(exe)], check=True)
seconds = sys.argv[1] if len(sys.argv) > 1 else "4"
for i in range(12):
    smp = B.BoxSampler(0, period_s=0.2).start()
    r = subprocess.run([str(exe), seconds, str(i)], capture_output=True, text=True, timeout=120)
    box = smp.stop()
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            row = json.loads(line)
            row["sclk_mhz"], row["power_w"] = box.get("sclk_mhz"), box.get("power_w")
            if box.get("sclk_mhz"):  # the pipe's duty at the clock the sampler saw (2.5 PFLOP/s = every SIMD's pipe busy every cycle at 2.4 GHz)
                row["matrix_pipe_duty_at_the_sampled_clock"] = round(row["TFLOPs"] / (2500.0 * box["sclk_mhz"]["median"] / 2400.0), 3)
            if box.get("power_w"):
                row["W_per_TFLOPs"] = round(box["power_w"]["median"] / row["TFLOPs"], 3)
            print(json.dumps(row), flush=True)
    if r.returncode != 0:
        print(json.dumps({"setting_index": i, "rc": r.returncode, "stderr": r.stderr[-500:]}), flush=True)
