#!/usr/bin/env python
"""Builds libleann_mi355x_emul.so: the COMPLETE product library (same sources, same C ABI) compiled for the host against
tests/hip_emul/full/hip/hip_runtime.h.  Two mechanical source substitutions are applied to copies of the files
(a block-scope `extern` array cannot be given a host definition, and LDS variables become function statics):
    extern __shared__ [__align__(16)] T name[];             ->   T* name = (T*)emul::dyn_smem();
    __shared__                                              ->   static
Test infrastructure only: nothing in leann_amd/ loads this library.
    python tests/hip_emul/build_emul_lib.py <out_dir> [--sanitize thread|address]"""
import re
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SRCS = ["lm_search.hip", "lm_tokens.hip", "lm_recompute.hip", "lm_encoder_ops.hip", "lm_attn_v2.hip", "lm_attn_v3.hip", "lm_encoder_ops2.hip", "lm_layer_tail_h384.hip", "lm_qkv_h384.hip", "lm_qkv_attn_h384.hip",
        "lm_gemm_ws_h384.hip", "lm_gemm_f16.hip", "lm_csr_reader.cpp", "lm_encoder_forward.cpp", "lm_timing.cpp"]


def build(out_dir: Path, sanitize: str | None = None) -> Path:
    out_dir.mkdir(parents=True, exist_ok=True)
    src = out_dir / "leann_amd" / "csrc"
    if src.exists():
        shutil.rmtree(src)
    src.mkdir(parents=True)
    (out_dir / "include").mkdir(exist_ok=True)
    shutil.copy(ROOT / "include" / "leann_mi355x.h", out_dir / "include" / "leann_mi355x.h")
    dyn = re.compile(r"extern\s+__shared__\s+(?:__align__\(\d+\)\s+)?([A-Za-z_][\w ]*?)\s+(\w+)\[\];")
    for f in (ROOT / "leann_amd" / "csrc").iterdir():
        if f.suffix not in (".hip", ".h", ".cpp"):
            continue
        t = f.read_text()
        t = dyn.sub(lambda m: f"{m.group(1)}* {m.group(2)} = ({m.group(1)}*)emul::dyn_smem();", t)
        t = re.sub(r"\b__shared__\b", "static", t)
        (src / f.name).write_text(t)
    lib = out_dir / ("libleann_mi355x_emul" + (f"_{sanitize}" if sanitize else "") + ".so")
    # plain build: -O0 (the host compile of lm_search.hip takes 6 s instead of 40; the test problems are tiny);
    # sanitizer builds: -O1 with line tables, one compiler process per source file
    flags = ["-std=c++20", "-fPIC", "-pthread", "-w", "-ffp-contract=off", f"-I{ROOT / 'tests' / 'hip_emul' / 'full'}"]
    flags += ["-O1", "-gline-tables-only", f"-fsanitize={sanitize}"] if sanitize else ["-O0"]
    objdir = out_dir / ("obj_" + (sanitize or "plain"))
    objdir.mkdir(exist_ok=True)

    def cc(name: str) -> Path:
        o = objdir / (name + ".o")
        r = subprocess.run([CLANG, *flags, "-x", "c++", "-c", str(src / name), "-o", str(o)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{name}:\n{r.stderr[-6000:]}")
        return o

    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(len(SRCS)) as ex:
        objs = list(ex.map(cc, SRCS))
    link = [CLANG, "-shared", "-pthread", *[str(o) for o in objs], "-o", str(lib)]
    if sanitize:
        link += [f"-fsanitize={sanitize}", "-shared-libsan"]
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-6000:])
    return lib


if __name__ == "__main__":
    san = sys.argv[sys.argv.index("--sanitize") + 1] if "--sanitize" in sys.argv else None
    print(build(Path(sys.argv[1]), san))
