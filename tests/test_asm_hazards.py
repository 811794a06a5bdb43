"""Static guard (CPU box: hipcc cross-compiles gfx950) for one class of bug that tolerance tests do not see.

A VALU read of an MFMA's destination registers needs wait states after the MFMA (11 for the 8-pass 32x32x16 fp16 MFMA on gfx950).  The compiler's
hazard recogniser inserts the s_nop for instructions it knows -- and nothing for the contents of an inline-asm block.  Round 5 shipped (for one
build) an attention kernel whose row-maximum tree was `asm("v_max3_f32 ...")` on the score MFMA's result registers: the tree read registers the
matrix pipe had not written yet, the results stayed fp16-close (a stale maximum only moves the deferred-rescale reference) and stopped being
bit-reproducible; only the GPU suite's "same bits from both launch paths" tests noticed.

The rule checked here is deliberately coarse: in the device assembly of every kernel source, NO VALU instruction that comes from an inline-asm block
may read a VGPR that is (part of) the destination of an MFMA anywhere in the same kernel.  The asm blocks the kernels do use -- LDS-DMA pieces, counted
waits, the GELU stream's `v_max_f32 x, 0, x` on a COPY of an accumulator (the copy is a compiler-visible v_accvgpr_read) -- pass it."""
import re
import subprocess
from pathlib import Path

import pytest

CSRC = Path(__file__).resolve().parents[1] / "leann_amd" / "csrc"
HIPCC = "/opt/rocm/bin/hipcc"


def _sources_and_flags():
    mk = (CSRC / "Makefile").read_text()
    srcs = re.search(r"^SRCS = (.*)$", mk, re.M).group(1).split()
    base = re.search(r"^HIPFLAGS = (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").replace("$(EXTRA_DEFS)", "").split()
    extra = {m.group(1): m.group(2).split() for m in re.finditer(r"^\$\(OBJDIR\)/([\w/]+)\.o: EXTRA = (.*)$", mk, re.M)}
    return [(s, base + extra.get(s.rsplit(".", 1)[0], [])) for s in srcs if s.endswith(".hip")]


def _regs(tok):
    """VGPR numbers named by one operand token: v12, v[4:7], |v3|, -v3, v3.l ..."""
    m = re.search(r"\bv\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"\bv(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def violations(asm_text):
    out = []
    for kernel in asm_text.split("s_endpgm"):
        mfma_dst = set()
        for m in re.finditer(r"^\s*v_(?:mfma|smfmac)\w*\s+([^,]+),", kernel, re.M):
            mfma_dst |= _regs(m.group(1))  # a[..] destinations name no VGPR: empty
        if not mfma_dst:
            continue
        name = re.findall(r"^(_Z\w+):", kernel, re.M)
        inside = False
        for line in kernel.splitlines():
            if "#ASMSTART" in line:
                inside = True
            elif "#ASMEND" in line:
                inside = False
            elif inside:
                t = line.strip()
                if not t.startswith("v_"):
                    continue
                ops = t.split(None, 1)[1].split(",") if " " in t or "\t" in t else []
                src = set().union(*[_regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
                if src & mfma_dst:
                    out.append((name[-1] if name else "?", t))
    return out


def test_the_rule_flags_the_round_5_bug():
    bad = """
_ZN2lm1kEv:
\tv_mfma_f32_32x32x16_f16 v[32:47], v[64:67], v[56:59], v[16:31]
\tds_read_b64_tr_b16 v[68:69], v66 offset:4096
\t;;#ASMSTART
\tv_max3_f32 v95, v32, v33, v34
\t;;#ASMEND
\ts_endpgm
"""
    good = bad.replace("v_max3_f32 v95, v32, v33, v34", "v_max_f32 v95, 0, v96").replace("v[32:47], v[64:67]", "a[0:15], v[64:67]")
    assert violations(bad) and not violations(good)


@pytest.mark.parametrize("src,flags", _sources_and_flags(), ids=lambda v: v if isinstance(v, str) else "")
def test_no_inline_asm_valu_reads_an_mfma_destination(src, flags, tmp_path):
    if not Path(HIPCC).exists():
        pytest.skip("no hipcc")
    text = (CSRC / src).read_text()
    if "asm" not in text and "mfma" not in text:
        return  # nothing to look at (and nothing to compile)
    out = tmp_path / "k.s"
    r = subprocess.run([HIPCC, *flags, "-I", str(CSRC), "--cuda-device-only", "-S", "-o", str(out), str(CSRC / src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    v = violations(out.read_text())
    assert not v, f"{src}: inline-asm VALU instructions read MFMA destination registers (no hazard wait states are inserted for them): {v[:5]}"
