"""Compile-time guard of the sweep kernel's phase loop (no GPU needed: hipcc cross-compiles).

hipcc's waitcnt pass once guarded the fragment registers with an `s_waitcnt vmcnt(0)` INSIDE such a loop -- a
drain of the LDS-DMA pipeline in every phase -- after an unrelated register-allocation change, and
its register allocator unties MFMA accumulators written through the builtin.  The loop must stay what the source says."""
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


def _kernel(isa_text, dt, count):
    name = f"_Z19topk_sweep16_kernelILi{dt}ELb{count}EEv11SweepParams"
    m = re.search(rf"^{name}:[^\n]*\n(.*?)s_endpgm", isa_text, re.S | re.M)
    assert m, name
    return m.group(1)


@pytest.mark.parametrize("dt,mfma,per_phase", [(0, "v_mfma_f32_16x16x32_bf16", 128), (1, "v_mfma_i32_16x16x64_i8", 128),
                                               (2, "v_mfma_f32_16x16x4_f32", 512)])
def test_phase_loop_is_clean(isa, dt, mfma, per_phase):
    """topk_sweep16_kernel (one wave per SIMD, 16 x 16 MFMA tiles, asm MFMAs tied in place): the phase loop is two basic
    blocks (H0-H2 with the barrier, then H3).  Per 128-byte phase: 128 products (x4 MFMAs for f32), 40 fragment reads, 16
    LDS-DMA pieces, ONE vmcnt wait (the explicit vmcnt(0) before the barrier) -- hipcc once added vmcnt waits inside such
    a loop when the epilogue's loads were not declared complete --, no accumulator shuffling (v_accvgpr_*) inside the
    loop, no scratch traffic, accumulators in AGPRs."""
    body = _kernel(isa, dt, 0)
    assert "scratch_" not in body, "the accumulators (or spills) went to scratch: is the unroll threshold flag set?"
    blocks = re.split(r"^\.LBB\d+_\d+:", body, flags=re.M)
    loop = [b for b in blocks if b.count(mfma) >= 32]
    # the first phase of a tile is peeled (its products start from C = 0 instead of zeroed accumulators): that copy comes
    # first in the code, the loop proper is the last two blocks
    assert len(loop) == 4, [b.count(mfma) for b in blocks if mfma in b]
    loop = loop[2:]
    text = "".join(loop)
    assert text.count(mfma) == per_phase
    assert text.count("ds_read_b128") == 40
    assert len(re.findall(r"buffer_load_dwordx4 .* lds", text)) == 16
    assert len(re.findall(r"^\s*s_barrier", text, re.M)) == 1
    assert re.findall(r"s_waitcnt vmcnt\((\d+)\)", text) == ["0"]
    assert "v_accvgpr" not in text, "hipcc moves accumulators inside the phase loop"
    assert re.search(mfma + r" a\[", text), "accumulators not in AGPRs"


def test_no_variant_spills_vectors(isa):
    """Neither the search/mining kernels nor the rank-counting (COUNT) ones touch scratch.  The COUNT variants spilled
    35 VGPRs in round 1 and 93-141 midway through round 2: hipcc hoisted the 16 column indices of every block's
    own-positive test out of the tile loop; the test now runs only in the one tile that holds such a column."""
    for dt in (0, 1, 2):
        for count in (0, 1):
            body = _kernel(isa, dt, count)
            assert "scratch_" not in body, f"dtype {dt} count {count}: vector spills"
