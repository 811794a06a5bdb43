"""Start-up selection of the encoder kernels (the untimed part of a run).

The default encoder path (hipBLASLt GEMMs + the first-generation hand-written kernels) is the one validated on
hardware.  The second-generation kernels (attention revision 2, fused feed-forward block, hidden-384 linear kernel,
16-lane LayerNorm, segmented mean pooling, fused embedding front end) are switched by ``LEANN_MI355X_*``
environment variables.  ``pick_encoder_switches`` decides which of them to turn on for THIS process's GPU the way a
GEMM library picks a solution: a child process (so that a device fault cannot take the caller down) encodes a
sample of synthetic chunks with the default path and with each candidate added in turn, keeps a candidate only if
its embeddings agree with the default path's (max |diff| <= tol on L2-normalised fp32 embeddings) AND it is
faster, and prints the accepted set after every decision.  Whatever it printed last before ending -- normally or
not -- is what the parent uses; nothing printed = the default path.  Both alternatives are GPU kernels; there is no
CPU fallback anywhere.

    python -m leann_amd.autotune --device 0          # prints one JSON line per decision
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time
from pathlib import Path

# (switches to add, label) in the order they are tried; later entries build on the accepted earlier ones
CANDIDATES = [
    ({"LEANN_MI355X_ATTN": "2"}, "attention revision 2"),
    ({"LEANN_MI355X_LN": "2"}, "LayerNorm, 16 lanes per row"),
    ({"LEANN_MI355X_POOL": "1"}, "segmented mean pooling"),
    ({"LEANN_MI355X_EMBED": "1"}, "fused embedding front end"),
    ({"LEANN_MI355X_PACK": "1"}, "packing front end in one kernel"),
    ({"LEANN_MI355X_LINEAR": "1"}, "hidden-384 linear kernel (QKV, out-projection + LayerNorm)"),
    ({"LEANN_MI355X_MLP": "1", "LEANN_MI355X_MLP_VARIANT": "2"}, "fused feed-forward block, cross-slab pipelined"),
    ({"LEANN_MI355X_MLP": "1", "LEANN_MI355X_MLP_VARIANT": "1"}, "fused feed-forward block"),
]
ALL_KEYS = sorted({k for c, _ in CANDIDATES for k in c})


def _child(device: int, model: str, n_chunks: int, tol: float, min_gain: float) -> None:
    import torch

    from .encoder import BertEncoder
    from .synth import CorpusSpec, SyntheticCorpus, pad_batch

    for k in ALL_KEYS:
        os.environ.pop(k, None)
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    enc = BertEncoder.load(model, allow_random=True)  # throughput + self-consistency only.to(dev, dtype=torch.float16).eval()
    ids, lens = pad_batch(*SyntheticCorpus(CorpusSpec(n_chunks=n_chunks, seed=99)).chunks(), 256)
    ti, tl = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)

    def run(env: dict):
        for k in ALL_KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        e = enc.encode_tokens_packed(ti, tl, 262144)  # warm-up (also packs weights for the fused kernels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            e = enc.encode_tokens_packed(ti, tl, 262144)
        torch.cuda.synchronize()
        return e, (time.perf_counter() - t0) / 3

    ref, t_best = run({})
    t_default = t_best
    accepted: dict = {}
    print(json.dumps({"accepted": accepted, "ms": round(t_best * 1e3, 2), "default_ms": round(t_default * 1e3, 2), "decision": "default path"}), flush=True)
    for add, label in CANDIDATES:
        if any(k in accepted for k in add):
            continue  # e.g. the second MLP variant when the first one was accepted
        trial = {**accepted, **add}
        print(json.dumps({"trying": label}), flush=True)  # if the process dies now, the parent knows which kernel did it
        e, t = run(trial)
        diff = float((e - ref).abs().max())
        ok = bool(torch.isfinite(e).all()) and diff <= tol
        keep = ok and t < t_best * (1.0 - min_gain)
        if keep:
            accepted, t_best = trial, t
        print(json.dumps({"accepted": accepted, "ms": round(t_best * 1e3, 2), "default_ms": round(t_default * 1e3, 2),
                          "decision": f"{label}: max|diff|={diff:.2e} {'ok' if ok else 'MISMATCH'}, {t * 1e3:.2f} ms -> {'kept' if keep else 'dropped'}"}),
              flush=True)


def pick_encoder_switches(device: int = 0, model: str = "sentence-transformers/all-MiniLM-L6-v2", n_chunks: int = 2048,
                          tol: float = 3e-3, min_gain: float = 0.02, timeout: float = 480.0) -> dict:
    """Returns {"switches": {...}, "log": [...]}; never raises (any failure = default path, reported in the log)."""
    log: list = []
    switches: dict = {}
    root = Path(__file__).resolve().parent.parent
    cmd = [sys.executable, "-m", "leann_amd.autotune", "--device", str(device), "--model", model, "--chunks", str(n_chunks),
           "--tol", str(tol), "--min-gain", str(min_gain)]
    env = {k: v for k, v in os.environ.items() if k not in ALL_KEYS}
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):  # the child is a plain single-GPU process
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, cwd=str(root), env=env, capture_output=True, text=True, timeout=timeout)
        out, rc = r.stdout, r.returncode
        if rc != 0:
            log.append({"child_exit": rc, "stderr_tail": r.stderr[-400:]})
    except subprocess.TimeoutExpired as ex:
        out = ex.stdout.decode() if isinstance(ex.stdout, bytes) else (ex.stdout or "")
        log.append({"child_exit": "timeout"})
    except Exception as ex:  # noqa: BLE001
        out = ""
        log.append({"child_exit": repr(ex)[:200]})
    for line in out.splitlines():
        try:
            d = json.loads(line)
        except Exception:  # noqa: BLE001
            continue
        log.append(d)
        if "accepted" in d:
            switches = d["accepted"]
    return {"switches": switches, "log": log}


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--model", default="sentence-transformers/all-MiniLM-L6-v2")
    ap.add_argument("--chunks", type=int, default=2048)
    ap.add_argument("--tol", type=float, default=3e-3)
    ap.add_argument("--min-gain", type=float, default=0.02)
    a = ap.parse_args()
    _child(a.device, a.model, a.chunks, a.tol, a.min_gain)
