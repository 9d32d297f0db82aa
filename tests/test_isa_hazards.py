"""Build-time ISA check of the hand-written DPP multiply-adds.  resample.hip emits `v_fmac_f32_dpp ... row_newbcast` as inline assembly;
LLVM's hazard recogniser does not look inside inline assembly, and gfx9 needs 2 wait states between a VALU write of a VGPR and a DPP read of it
(5 after a VALU write of EXEC).  Today the tap registers come straight from loads; a compiler upgrade that copies one through a v_mov right in front
of the multiply-add would silently read stale data.  This test compiles the file to assembly and checks every such instruction."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fluidaudio_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _regs(tok):
    """v12 -> {12}; v[4:7] -> {4, 5, 6, 7}; anything else -> empty."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_dpp_multiply_adds_of_the_resampler_respect_the_valu_to_dpp_hazard(tmp_path):
    out = tmp_path / "resample.s"
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only", os.path.join(CSRC, "resample.hip"),
                        "-o", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    insts = []
    for line in out.read_text().splitlines():
        t = line.split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            continue
        insts.append(t)
    checked = 0
    for i, t in enumerate(insts):
        if not t.startswith("v_fmac_f32_dpp"):
            continue
        ops = [o.strip() for o in t[len("v_fmac_f32_dpp"):].split(",")]
        src = _regs(ops[1].split()[0])                      # the DPP operand: the register of 16 taps
        assert src, t
        waits = 0
        for back in range(1, 6):                            # wait states seen walking back: every instruction is one, s_nop N is N + 1
            if i - back < 0:
                break
            p = insts[i - back]
            m = re.match(r"s_nop\s+(\d+)", p)
            if p.startswith("v_") and not p.startswith("v_fmac_f32_dpp"):
                dst = _regs(p.split(None, 1)[1].split(",")[0].strip()) if " " in p else set()
                assert not (dst & src and waits < 2), f"VALU write of the DPP source {sorted(dst & src)} {waits} wait state(s) before: {p!r} -> {t!r}"
                assert not (p.startswith("v_cmpx") and waits < 5), f"EXEC written {waits} wait state(s) before a DPP read: {p!r} -> {t!r}"
            waits += int(m.group(1)) + 1 if m else 1
        checked += 1
    assert checked > 100, checked                           # the kernels are there (down = 6 alone has 169 x 8 of them)
