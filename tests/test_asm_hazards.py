"""Static guards on the device assembly (CPU box: hipcc cross-compiles gfx950): register budgets / no scratch for the occupancy each kernel is
designed for (`test_register_budgets_and_no_scratch`), and a class of bug that tolerance tests do not see: hardware hazards inside inline-asm
blocks.  Three rules: `violations` (VALU reads of MFMA destinations, below), `sgpr_violations` (VMEM reads of VALU-written SGPRs) and
`scratch_violations` (no scratch traffic in kernels whose asm blocks count their vector-memory operations).

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


def sgpr_violations(asm_text, need=5):
    """Second rule: a VMEM instruction inside an inline-asm block (the LDS-DMA pieces: global_load_lds_dwordx4 vOFF, s[BASE:BASE+1]) whose SGPR
    base was written by a VALU instruction (v_readfirstlane / v_readlane -- the latter is how the compiler reloads a spilled SGPR) fewer than
    `need` wait states earlier.  The compiler inserts `s_nop 4` for VMEM instructions it knows; the asm block has to carry its own (s_mov m0 +
    s_nop 3 in lm_dma16_sv).  Wait states are counted along the layout order inside a basic block (a label resets the window: conservative
    towards NOT reporting across a branch target, where the distance is unknown)."""
    out = []
    recent = []  # [(sgpr number, wait states since its VALU definition)]
    inside = False
    for line in asm_text.splitlines():
        t = line.strip()
        if "#ASMSTART" in t:
            inside = True
            continue
        if "#ASMEND" in t:
            inside = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", t):
                recent = []
            continue
        if re.match(r"^[\w$.]+:", t):
            recent = []
            continue
        ws = int(t.split()[1]) + 1 if t.startswith("s_nop") else 1
        if inside and t.startswith(("global_load", "buffer_load", "global_store", "buffer_store")):
            m = re.search(r"\bs\[(\d+):(\d+)\]", t)
            if m:
                used = set(range(int(m.group(1)), int(m.group(2)) + 1))
                for sg, c in recent:
                    if sg in used and c < need:
                        out.append((t, f"s{sg} written by a VALU instruction {c} wait states earlier"))
        recent = [(sg, c + ws) for sg, c in recent if c + ws < 16]
        m = re.match(r"(?:v_readfirstlane_b32|v_readlane_b32)\s+s(\d+)\b", t)
        if m:
            recent.append((int(m.group(1)), 0))
    return out


def scratch_violations(asm_text):
    """Third rule: a kernel whose asm blocks carry COUNTED vector-memory waits (`s_waitcnt vmcnt(n)`, n > 0: "my older requests have landed, the n
    younger ones may be in flight") must not touch scratch memory -- a spill store or reload is one more vector-memory operation in the wave's in-order
    counter, which the count in the source knows nothing about.  (Spills to AGPRs or to VGPR lanes are register traffic and fine.)"""
    counted = set()
    name, inside = None, False
    for line in asm_text.splitlines():
        t = line.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            name = m.group(1)
        if "#ASMSTART" in t:
            inside = True
        elif "#ASMEND" in t:
            inside = False
        elif inside and name:
            m = re.match(r"s_waitcnt\s+vmcnt\((\d+)\)", t)
            if m and int(m.group(1)) > 0:
                counted.add(name)
    out = []
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", asm_text, re.S):
        b = m.group(0)
        kn = re.search(r"\.name:\s+(\S+)", b).group(1)
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1))
        if kn in counted and scratch:
            out.append((kn, f"{scratch} B of scratch per lane in a kernel with counted vmcnt waits"))
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
    spill = """
\tv_readlane_b32 s12, v247, 3
\tv_readlane_b32 s13, v247, 4
\t;;#ASMSTART
\ts_mov_b32 m0, s71
\ts_nop 0
\tglobal_load_lds_dwordx4 v184, s[12:13]
\t;;#ASMEND
"""
    assert sgpr_violations(spill) and not sgpr_violations(spill.replace("s_nop 0", "s_nop 3"))


def undeclared_m0_writes(source_text):
    """Fourth rule, on the SOURCE (a clobber list leaves no trace in the assembly): an asm statement that writes M0 names "m0" among its clobbers.
    The compiler re-materialises M0 in front of its own uses but may merge equal initialisations over a region; an asm block that changes M0 behind
    its back must say so."""
    out = []
    for m in re.finditer(r"asm\s+volatile\s*\((.*?)\)\s*;", source_text, re.S):
        stmt = m.group(1)
        if re.search(r"s_mov_b32\s+m0\b|s_add_u32\s+m0\b|s_movk_i32\s+m0\b", stmt) and not re.search(r':[^:]*"m0"[^:]*$', stmt):
            out.append(" ".join(stmt.split())[:160])
    return out


def test_m0_writes_are_declared():
    assert undeclared_m0_writes('asm volatile("s_mov_b32 m0, %1\\n\\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(m) : "memory");')
    assert not undeclared_m0_writes('asm volatile("s_mov_b32 m0, %1\\n\\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(m) : "memory", "m0");')
    bad = {}
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + list(CSRC.glob("diag/*.hip"))):
        v = undeclared_m0_writes(f.read_text())
        if v:
            bad[f.name] = v
    assert not bad, bad


_ASM_CACHE = {}


def _assembly(src, flags, tmp_path_factory):
    """gfx950 assembly of one kernel source with the Makefile's flags, compiled once per test session"""
    if src not in _ASM_CACHE:
        out = tmp_path_factory.mktemp("asm") / (src + ".s")
        r = subprocess.run([HIPCC, *flags, "-I", str(CSRC), "--cuda-device-only", "-S", "-o", str(out), str(CSRC / src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        _ASM_CACHE[src] = out.read_text()
    return _ASM_CACHE[src]


def kernel_resources(asm_text):
    """{mangled kernel name: {"vgpr": total VGPRs (architectural + accumulator), "agpr", "sgpr", "scratch": bytes per lane}} from the metadata"""
    res = {}
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", asm_text, re.S):
        b = m.group(0)
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)  # noqa: E731
        res[g("name")] = {"vgpr": int(g("vgpr_count")), "agpr": int(g("agpr_count")), "sgpr": int(g("sgpr_count")), "scratch": int(g("private_segment_fixed_size"))}
    return res


# The occupancy each kernel is DESIGNED for: waves per SIMD = 512 / VGPRs.  A source or compiler change that pushes a kernel over its
# line costs a wave per SIMD (or, with scratch, far more) without failing any numerics test -- round 5 measured exactly that once (attention with two
# score tuples squeezed into 128 registers: 76 B of scratch, 365 instead of 271 us).
REGISTER_BUDGETS = [
    ("lm_attn_v3.hip", r"k_attn_varlen_hd32_v3ILi\d+ELi0E", 128),       # four workgroups of four waves per CU
    ("lm_attn_v2.hip", r"k_attn_varlen_hd32_v2ILi\d+ELi32E", 128),
    ("lm_attn_v2.hip", r"k_attn_varlen_hd32_v2ILi\d+ELi64E", 168),      # head_dim 64: three waves per SIMD
    ("lm_qkv_h384.hip", r"k_qkv_h384", 256),                             # two waves per SIMD
    ("lm_gemm_f16.hip", r"k_gemm_f16", 256),
    ("lm_gemm_ws_h384.hip", r"k_gemm_ws_h384", 256),
    ("lm_layer_tail_h384.hip", r"k_layer_tail_h384", 512),               # one wave per SIMD: the whole register file
]


@pytest.mark.parametrize("src,pattern,budget", REGISTER_BUDGETS, ids=[f"{a}:{b[:28]}" for a, b, _ in REGISTER_BUDGETS])
def test_register_budgets_and_no_scratch(src, pattern, budget, tmp_path_factory):
    if not Path(HIPCC).exists():
        pytest.skip("no hipcc")
    flags = dict(_sources_and_flags())[src]
    res = {k: v for k, v in kernel_resources(_assembly(src, flags, tmp_path_factory)).items() if re.search(pattern, k)}
    assert res, f"no kernel of {src} matches {pattern}"
    over = {k: v for k, v in res.items() if v["vgpr"] > budget or v["scratch"]}
    assert not over, over


@pytest.mark.parametrize("src,flags", _sources_and_flags(), ids=lambda v: v if isinstance(v, str) else "")
def test_no_hazards_inside_inline_asm_blocks(src, flags, tmp_path_factory):
    if not Path(HIPCC).exists():
        pytest.skip("no hipcc")
    text = (CSRC / src).read_text()
    if "asm" not in text and "mfma" not in text:
        return  # nothing to look at (and nothing to compile)
    asm = _assembly(src, flags, tmp_path_factory)
    v = violations(asm) + sgpr_violations(asm) + scratch_violations(asm)
    assert not v, f"{src}: hazards inside inline-asm blocks, which the compiler's hazard recogniser does not look into: {v[:5]}"
