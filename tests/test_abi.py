"""The C-ABI library loads and exports every symbol include/leann_mi355x.h declares (no GPU needed)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    src = (ROOT / "include" / "leann_mi355x.h").read_text()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(lm_[a-z0-9_]+)\s*\(", src))
    names -= {"lm_provider_fn"}
    return sorted(names)


def test_header_symbols_are_exported(built_libs):
    from leann_amd import _lib

    lib = _lib.load()
    decl = _declared()
    assert len(decl) >= 20
    missing = [n for n in decl if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(set(_lib.EXPORTED_SYMBOLS)) == decl  # the Python binding list tracks the header


def test_no_gpu_fails_loudly(built_libs):
    """Without a HIP device every compute entry point reports LM_EHIP -- there is no CPU fallback."""
    import numpy as np

    from leann_amd import _lib
    from leann_amd.index import Mi355xIndex
    from leann_amd.hnsw_builder import build_hnsw

    lib = _lib.load()
    if lib.lm_device_count() > 0:
        pytest.skip("GPU present")
    g = build_hnsw(np.random.default_rng(0).standard_normal((50, 64)).astype(np.float32), "l2", M=4, ef_construction=10)
    with pytest.raises(_lib.LeannMi355xError, match="no HIP device"):
        Mi355xIndex.from_csr(g)
    with pytest.raises(_lib.LeannMi355xError):
        _lib.require_gpu()
    h = C.c_void_p()
    off = np.zeros(2, np.uint64)
    assert lib.lm_tokens_create(None, off.ctypes.data_as(C.c_void_p), 1, 0, C.byref(h)) == _lib.LM_EHIP


def test_argument_validation_precedes_device_use(built_libs):
    import numpy as np

    from leann_amd import _lib

    lib = _lib.load()
    h = C.c_void_p()
    z = np.zeros(4, np.uint64)
    zi = np.zeros(4, np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    # bad metric / dims
    assert lib.lm_index_create_from_csr(1, 0, 0, vp(z), vp(z), 2, vp(zi), 0, vp(zi), 0, 0, 0, C.byref(h)) == _lib.LM_EINVAL
    assert lib.lm_index_create_from_csr(1, 8, 7, vp(z), vp(z), 2, vp(zi), 0, vp(zi), 0, 0, 0, C.byref(h)) == _lib.LM_EINVAL
    # inconsistent CSR: node_offsets[N] != len(level_ptr)
    no = np.array([0, 2], np.uint64)
    lv = np.array([1], np.int32)
    assert lib.lm_index_create_from_csr(1, 8, 0, vp(no), vp(z), 3, vp(zi), 0, vp(lv), 0, 0, 0, C.byref(h)) == _lib.LM_EFORMAT
    assert b"node_offsets" in lib.lm_last_error()
    p = _lib.SearchParams()
    lib.lm_search_params_default(C.byref(p))
    assert (p.efSearch, p.beam_size, p.check_relative_distance, p.recompute) == (64, 1, 1, 1)
    assert lib.lm_index_search(None, 1, None, 1, None, None, C.byref(p)) == _lib.LM_EINVAL


def _build_c_host(tmp_path):
    import subprocess

    exe = tmp_path / "abi_host"
    lib_dir = ROOT / "leann_amd" / "lib"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", f"-I{ROOT / 'include'}", "-o", str(exe),
                    str(ROOT / "tests" / "abi" / "abi_host.c"), f"-L{lib_dir}", "-lleann_mi355x", "-lm", f"-Wl,-rpath,{lib_dir}"],
                   check=True, capture_output=True)
    return exe


def test_plain_c_host(built_libs, tmp_path):
    """A C11 program (no Python, no torch) compiles against include/leann_mi355x.h with -Werror, links the library and
    passes its checks: argument validation everywhere; without a GPU the loud LM_EHIP path."""
    import subprocess

    r = subprocess.run([str(_build_c_host(tmp_path))], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "PATH-OK" in r.stdout


@pytest.mark.gpu
def test_plain_c_host_known_answer_on_gpu(tmp_path):
    """Same program on the GPU box: the hand-traced line graph searched through lm_index_search from C."""
    import subprocess

    r = subprocess.run([str(_build_c_host(tmp_path))], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "GPU-PATH-OK" in r.stdout
