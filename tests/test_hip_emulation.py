"""Runs the REAL sources of the attention kernel (leann_amd/csrc/lm_attn_v2.hip) and of the elementwise kernels (lm_encoder_ops.hip,
lm_encoder_ops2.hip) on the CPU: tests/hip_emul compiles them for x86 with stub HIP headers and executes one workgroup at a time with a
thread per lane, MFMA / shuffles / __syncthreads as barriers.  (The hidden-384 GEMM kernels run the same way inside the emulated LIBRARY:
tests/test_emulated_search.py.)"""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def emulator(tmp_path_factory):
    if not Path(CLANG).exists() and not shutil.which("clang++"):
        pytest.skip("needs a clang++ (ext_vector_type, _Float16)")
    exe = tmp_path_factory.mktemp("hip_emul") / "run_kernels"
    cmd = [CLANG if Path(CLANG).exists() else "clang++", "-std=c++20", "-O1", "-pthread", "-w", f"-I{ROOT / 'tests' / 'hip_emul' / 'full'}",
           f"-I{ROOT / 'include'}", str(ROOT / "tests" / "hip_emul" / "run_kernels.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.parametrize("what", ["attention", "elementwise"])
def test_kernel_sources_run_correctly_on_the_host(emulator, what):
    r = subprocess.run([str(emulator), what], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert "FAIL" not in r.stdout.replace("FAILED", "")


@pytest.mark.parametrize("sanitizer,marker", [("thread", "ThreadSanitizer"), ("address", "AddressSanitizer")])
def test_kernels_are_clean_under_sanitizers(tmp_path, sanitizer, marker):
    """The same harness built with -fsanitize=thread / address.
    thread: the emulation's barriers are the only synchronisation between lanes, so a missing or misplaced
    __syncthreads() around the staged K / V^T tiles is a data race.
    address: every global buffer is an exact-size heap array, so an out-of-range row / column / tail access is reported."""
    if not Path(CLANG).exists():
        pytest.skip("needs ROCm's clang++ with the sanitizer runtimes")
    exe = tmp_path / f"run_kernels_{sanitizer}"
    cmd = [CLANG, "-std=c++20", "-O1", "-g", "-pthread", "-w", f"-fsanitize={sanitizer}", f"-I{ROOT / 'tests' / 'hip_emul' / 'full'}",
           f"-I{ROOT / 'include'}", str(ROOT / "tests" / "hip_emul" / "run_kernels.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "san" in r.stderr.lower() and "cannot find" in r.stderr.lower():
        pytest.skip("sanitizer runtime not available")
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe), "all"], capture_output=True, text=True, timeout=1800,
                       env={"TSAN_OPTIONS": "halt_on_error=0", "ASAN_OPTIONS": "detect_leaks=0"})
    out = r.stdout + r.stderr
    assert marker not in out, out[-4000:]
    assert r.returncode == 0 and "ALL OK" in r.stdout
