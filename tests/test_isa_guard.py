"""Compile-time guard of the sweep kernel's steady-state phase loop (no GPU needed: hipcc cross-compiles).

hipcc's waitcnt pass once guarded the fragment registers with an `s_waitcnt vmcnt(0)` INSIDE that loop -- a
drain of the LDS-DMA ring in every phase -- after an unrelated register-allocation change.
The loop must stay: 12 ds_read_b128, 4 LDS-DMA pieces, one counted vmcnt(8), 16 MFMAs, 2 barriers, no
spill reloads and no scratch traffic."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vllm_amd", "csrc")
SRCS = [os.path.join(CSRC, f"sweep_inst_{n}.hip") for n in ("bf16", "i8", "f32")]   # one translation unit per element type


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    tmp = tmp_path_factory.mktemp("isa")
    procs = []
    for src in SRCS:                                                   # the three compile side by side
        out = tmp / (os.path.basename(src) + ".s")
        procs.append((out, subprocess.Popen(
            ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-value",
             "-mllvm", "-pragma-unroll-threshold=200000",              # as vllm_amd/csrc/Makefile
             "-S", "--cuda-device-only", src, "-o", str(out)], cwd=CSRC,
            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    text = ""
    for out, pr in procs:
        assert pr.wait() == 0, out
        text += out.read_text()
    return text


def _kernel(isa_text, dt, count, four_wave=False):
    name = (f"_Z18topk_sweep4_kernelILi{dt}ELb{count}EEv11SweepParams" if four_wave
            else f"_Z17topk_sweep_kernelILi{dt}ELb{count}EEv11SweepParams")
    m = re.search(rf"^{name}:[^\n]*\n(.*?)s_endpgm", isa_text, re.S | re.M)
    assert m, name
    return m.group(1)


def _first_inner_loop(body):
    """The first depth-2 loop of the kernel is the straight-line steady-state phase loop."""
    lines = body.split("\n")
    start = next(i for i, l in enumerate(lines) if "This Inner Loop Header" in l)
    # the loop ends at the first backward conditional branch after its second s_barrier
    barriers = 0
    for j in range(start, len(lines)):
        if re.match(r"\s*s_barrier", lines[j]):
            barriers += 1
        if barriers >= 2 and re.match(r"\s*s_cbranch_scc", lines[j]):
            return lines[start:j + 1]
    raise AssertionError("loop end not found")


@pytest.mark.parametrize("dt,mfma", [(0, "v_mfma_f32_32x32x16_bf16"), (1, "v_mfma_i32_32x32x32_i8"), (2, "v_mfma_f32_32x32x2_f32")])
def test_steady_state_phase_loop_is_clean(isa, dt, mfma):
    loop = _first_inner_loop(_kernel(isa, dt, 0))
    text = "\n".join(loop)
    assert text.count("ds_read_b128") == 12
    assert len(re.findall(r"buffer_load_dwordx4 .* lds", text)) == 4
    assert text.count(mfma) == (16 if dt != 2 else 64)
    assert len(re.findall(r"^\s*s_barrier", text, re.M)) == 2
    waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", text)
    assert waits == ["8"], waits                      # the one counted wait; never a drain
    assert "v_readlane" not in text and "scratch_" not in text and "s_cbranch_vcc" not in text


def test_main_variants_do_not_spill_vectors(isa):
    for dt in (0, 1, 2):
        body = _kernel(isa, dt, 0)
        assert "scratch_" not in body, f"dtype {dt}: vector spills in the mining/search variant"


@pytest.mark.parametrize("dt,mfma,per_phase", [(0, "v_mfma_f32_32x32x16_bf16", 64), (1, "v_mfma_i32_32x32x32_i8", 64),
                                               (2, "v_mfma_f32_32x32x2_f32", 256)])
def test_four_wave_phase_loop_is_clean(isa, dt, mfma, per_phase):
    """topk_sweep4_kernel (one wave per SIMD): the phase loop is two basic blocks (steps 1-3 with the barrier, then
    step 0).  Per 128-byte phase: 32 fragment reads, 16 LDS-DMA pieces, ONE vmcnt wait (the explicit vmcnt(0) before
    the barrier) -- hipcc once added vmcnt(0)/(1) in step 0, a wait for the pieces issued one step earlier, when the
    epilogue's loads were not declared complete -- no scratch traffic, accumulators in AGPRs."""
    body = _kernel(isa, dt, 0, four_wave=True)
    assert "scratch_" not in body, "the accumulators (or spills) went to scratch: is the unroll threshold flag set?"
    blocks = re.split(r"^\.LBB\d+_\d+:", body, flags=re.M)
    loop = [b for b in blocks if mfma in b]
    assert len(loop) == 2, [b.count(mfma) for b in loop]
    text = "".join(loop)
    assert text.count(mfma) == per_phase
    assert text.count("ds_read_b128") == 32
    assert len(re.findall(r"buffer_load_dwordx4 .* lds", text)) == 16
    assert len(re.findall(r"^\s*s_barrier", text, re.M)) == 1
    assert re.findall(r"s_waitcnt vmcnt\((\d+)\)", text) == ["0"]
    assert re.search(mfma + r" a\[", text), "accumulators not in AGPRs"
