"""The product library itself -- C ABI, host loop and every HIP search kernel -- executed on the CPU: tests/hip_emul/
build_emul_lib.py compiles the same sources for the host (HIP runtime stubbed by host memory, a launch = blocks run one
after another with one OS thread per lane, collectives and __syncthreads() as barriers) and tests/emulated_search_cases.py
drives it through the ABI against the oracle: bit-exact labels, distances and distance-evaluation counts in stored-table
(persistent / lock-step, L2 / IP, padded D, fp16), recompute (all update variants, memo), PQ traversal and two-level modes.
Under ThreadSanitizer the only inter-lane hand-overs not ordered by a barrier are the ones marked LM_WAVE_SYNC().
This is test infrastructure: nothing in leann_amd/ can load the emulated library."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CLANG = Path("/opt/rocm/lib/llvm/bin/clang++")
sys.path.insert(0, str(ROOT / "tests" / "hip_emul"))


@pytest.fixture(scope="module")
def emul_dir(tmp_path_factory, built_libs):
    if not CLANG.exists():
        pytest.skip("needs ROCm's clang++ as a host compiler")
    return tmp_path_factory.mktemp("emul_lib")


def test_product_library_on_the_cpu_matches_the_oracle(emul_dir):
    import build_emul_lib

    lib = build_emul_lib.build(emul_dir)
    r = subprocess.run([sys.executable, "-m", "tests.emulated_search_cases", str(lib)], cwd=str(ROOT), capture_output=True, text=True,
                       timeout=1800)
    if r.returncode != 0 or "ALL CASES OK" not in r.stdout:  # keep the whole transcript: the tail alone does not always name the case
        (emul_dir / "emulated_cases_failure.log").write_text(r.stdout + "\n==== stderr ====\n" + r.stderr)
    assert r.returncode == 0 and "ALL CASES OK" in r.stdout, f"(full log: {emul_dir / 'emulated_cases_failure.log'})\n" + r.stdout[-3000:] + r.stderr[-5000:]
    assert "MISMATCH" not in r.stdout


def test_real_leann_searcher_drives_the_product_library_on_the_cpu(emul_dir, tmp_path):
    """the reference's own leann.api.LeannSearcher (api.py:623-642,644-796) -> BACKEND_REGISTRY["mi355x"] -> our
    searcher -> C ABI -> the product's kernel sources (host build).  tests/real_caller_over_emulation.py holds the scenario; the same
    test against the real library on an MI355X is tests/test_gpu_plugin_callers.py::test_real_leann_searcher_on_top_of_the_backend,
    which needs leann-core on the GPU box."""
    import build_emul_lib

    ref = Path("/root/reference/packages/leann-core/src")
    if not (ref / "leann" / "api.py").exists():
        pytest.skip("leann-core (the reference's caller) is not on this box")
    lib = build_emul_lib.build(emul_dir)
    r = subprocess.run([sys.executable, "-m", "tests.real_caller_over_emulation", str(lib), str(ref), str(tmp_path)], cwd=str(ROOT),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "REAL CALLER OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_two_rank_gloo_over_the_emulated_library(emul_dir):
    """SURVEY 8(e) with the product's own kernels on the CPU box: broadcast_graph + PartitionedSearch + ShardedSearch (incl. the
    lm_topk_merge kernel) on two gloo ranks, each over the emulated library -- the scenario tests/test_distributed.py runs over RCCL
    where two GPUs are visible (tests/emulated_two_rank.py)."""
    import socket

    import build_emul_lib
    import torch.multiprocessing as mp

    from tests.emulated_two_rank import worker

    lib = build_emul_lib.build(emul_dir)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(worker, args=(2, port, str(lib), out), nprocs=2, join=True)
    assert dict(out) == {0: True, 1: True}


def test_plain_c_host_known_answer_against_the_emulated_library(emul_dir):
    """tests/abi/abi_host.c (C11, no Python) linked against the host build: with a "device" present it takes the same
    branch as on the GPU box and checks the hand-traced known-answer search through lm_index_search."""
    import build_emul_lib

    lib = build_emul_lib.build(emul_dir)
    link = emul_dir / "libleann_mi355x.so"  # the program links -lleann_mi355x
    if not link.exists():
        link.symlink_to(lib.name)
    exe = emul_dir / "abi_host"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", f"-I{ROOT / 'include'}", "-o", str(exe),
                    str(ROOT / "tests" / "abi" / "abi_host.c"), f"-L{emul_dir}", "-lleann_mi355x", "-lm", f"-Wl,-rpath,{emul_dir}"],
                   check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GPU-PATH-OK" in r.stdout


def test_search_kernels_have_no_unmarked_lane_races(emul_dir):
    import build_emul_lib

    rt = subprocess.run([str(CLANG), "-print-file-name=libclang_rt.tsan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not Path(rt).exists():
        pytest.skip("ThreadSanitizer runtime not available")
    lib = build_emul_lib.build(emul_dir, sanitize="thread")
    env = dict(os.environ, LD_PRELOAD=rt, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4", OMP_NUM_THREADS="1")
    # (the last case: the built-in recompute provider's kernels and the general-width one-call forward -- GEMM, head_dim-64 attention,
    # LayerNorm, CLS pooling)
    cases = ["table_mips", "recompute_memo", "pq_deferred", "two_level", "degenerate_graphs", "layer_tail_small", "attention_v3", "native_recompute_general_hd64_cls"]
    r = subprocess.run([sys.executable, "-m", "tests.emulated_search_cases", str(lib), *cases], cwd=str(ROOT), capture_output=True,
                       text=True, timeout=3000, env=env)
    out = r.stdout + r.stderr
    assert "ALL CASES OK" in r.stdout, out[-4000:]
    # only reports located in the library under test count: the oracle and torch run OpenMP regions on libgomp, whose own
    # barriers ThreadSanitizer cannot see (false positives inside liblm_oracle / libtorch_cpu)
    races = [ln for ln in out.splitlines() if ln.startswith("SUMMARY: ThreadSanitizer") and ("leann_amd/csrc" in ln or "hip_emul" in ln or "lm::" in ln)]
    assert not races, "\n".join(races[:10]) + out[-3000:]


def test_sort_and_counting_merge_agree(emul_dir):
    """A hop's new keys get their places in the candidate list either by a bitonic sort + rank merge or -- when they are few (round 6) -- by counting
    (csrc/lm_beam_common.h: rank_merge_unsorted; callers k_update, k_search_table, k_pq_traverse).  The emulated cases' small graphs always take the counting
    branch at the library's limits; LM_EMUL_COUNTING_MERGE_LIMIT (host emulation only) moves the limit so that the same oracle comparisons walk the sort
    (0: never count) and an in-between mix (12)."""
    import build_emul_lib

    lib = build_emul_lib.build(emul_dir)
    cases = ["table_mips", "table_l2_d100", "recompute_memo", "recompute_wave_variant", "pq_deferred", "pq_table", "two_level", "dynamic_batching"]
    for limit in ("0", "12"):
        env = dict(os.environ, LM_EMUL_COUNTING_MERGE_LIMIT=limit)
        r = subprocess.run([sys.executable, "-m", "tests.emulated_search_cases", str(lib), *cases], cwd=str(ROOT), capture_output=True, text=True, timeout=1500, env=env)
        assert r.returncode == 0 and "ALL CASES OK" in r.stdout and "MISMATCH" not in r.stdout, f"limit {limit}:\n" + r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.skipif(os.environ.get("LEANN_EMUL_ASAN") != "1", reason="opt-in (adds ~1.5 min): LEANN_EMUL_ASAN=1")
def test_search_kernels_stay_in_bounds_under_address_sanitizer(emul_dir):
    """Device allocations of the emulated runtime are exact-size heap blocks: any read or write past a graph / pool /
    bitmap / table array in the kernels or the host loop is an AddressSanitizer error.  (Clean on all cases, 2026-09.)"""
    import build_emul_lib

    rt = subprocess.run([str(CLANG), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not Path(rt).exists():
        pytest.skip("AddressSanitizer runtime not available")
    lib = build_emul_lib.build(emul_dir, sanitize="address")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=0", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "tests.emulated_search_cases", str(lib)], cwd=str(ROOT), capture_output=True, text=True,
                       timeout=3000, env=env)
    out = r.stdout + r.stderr
    assert "ALL CASES OK" in r.stdout, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out, out[-6000:]
