"""Compile-time guards on the gfx950 code objects (hipcc cross-compiles without a GPU): register / LDS budgets the kernels'
designs rely on, no scratch traffic, and the one compiler hazard this repo works around.

* The Winograd conv kernels own a whole CU (one wave per SIMD, 512 registers, 138 240 B of LDS): a spill turns the loop's
  in-order memory queue into a scratch queue (the persistent-workgroup experiment of round 4: 255 spilled registers, 30 % slower).
* v_cvt_scalef32_2xpk16_fp6_f32 reads its 32 source registers while it writes its 6 destination registers; hipcc (ROCm 7.2) lets
  them overlap when the builtin is used directly (tools/probes/f6_probe.hip: two values of a block came out as +-7.5).
  md_cvt_2xpk16_fp6 wraps the instruction with an early-clobber destination: every occurrence must have disjoint ranges.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "meshdiffusion_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")


def _asm(src, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", f"-I{ROOT}/include", f"-I{CSRC}",
                    os.path.join(CSRC, src), "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    return out.read_text()


def _kernels(text):
    """{mangled name: {vgpr_count, vgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size}} from the metadata."""
    res = {}
    for blk in text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))      # noqa: E731
        res[name] = {k: get(k) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                                        "group_segment_fixed_size")}
    return res


def _ranges(operand):
    m = re.match(r"v\[(\d+):(\d+)\]", operand)
    return (int(m.group(1)), int(m.group(2))) if m else None


def test_winograd_conv_kernels_fit_the_register_file_without_scratch(tmp_path):
    text = _asm("conv3_wino.hip", tmp_path)
    ks = {n: v for n, v in _kernels(text).items() if "md_conv3_wino_kernel" in n}
    assert len(ks) == 6, list(ks)                                   # bf16x3, f16f8, f16f6, each with and without a residual operand
    for name, k in ks.items():
        assert k["vgpr_count"] <= 512 and k["vgpr_spill_count"] == 0 and k["sgpr_spill_count"] <= 4, (name, k)
        assert k["private_segment_fixed_size"] == 0, (name, k)
        assert k["group_segment_fixed_size"] == 138240, (name, k)   # halo buffers + offset tables; the exchange area aliases them
    assert "scratch_" not in text
    # the residual forms park one SGPR pair (the residual pointer) in a VGPR's lanes across the main loop: allowed OUTSIDE the MFMA
    # loop only (a v_readlane / v_writelane between the first and the last MFMA of a kernel would sit in the issue stream)
    for name in ks:
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")].split("\n")
        mf = [i for i, ln in enumerate(body) if "v_mfma" in ln]
        lanes = [i for i, ln in enumerate(body) if "v_readlane" in ln or "v_writelane" in ln]
        assert all(i < mf[0] or i > mf[-1] for i in lanes), name
    # the weight packer of the f16f6 fragments uses the e2m3 conversion
    assert "v_cvt_scalef32_2xpk16_fp6_f32" in text


@pytest.mark.parametrize("src", ["conv3_wino.hip", "wino_prep2.hip"])
def test_fp6_conversion_destination_never_overlaps_its_sources(src, tmp_path):
    text = _asm(src, tmp_path)
    n = 0
    for ln in text.splitlines():
        if "v_cvt_scalef32_2xpk16_fp6_f32" not in ln:
            continue
        ops = [o.strip() for o in ln.split("v_cvt_scalef32_2xpk16_fp6_f32")[1].split(",")]
        dst, s0, s1 = _ranges(ops[0]), _ranges(ops[1]), _ranges(ops[2])
        assert dst and s0 and s1 and dst[1] - dst[0] == 5 and s0[1] - s0[0] == 15 and s1[1] - s1[0] == 15, ln
        for s in (s0, s1):
            assert dst[1] < s[0] or dst[0] > s[1], ln
        n += 1
    assert n >= 1


def test_attention_score_mfmas_carry_their_hazard_nops(tmp_path):
    """md_attn_fwd's S^T MFMAs are inline asm (hipcc's hazard recogniser does not see them): the MFMA -> VALU wait states must be part of
    the same asm statement as each score accumulator's last MFMA, i.e. the `s_nop 15 / s_nop 3` pair directly follows a v_mfma twice per
    tile, and no VALU instruction sits between them."""
    body = _asm("attention.hip", tmp_path).split("\n")
    hits = [i for i, ln in enumerate(body) if ln.strip() == "s_nop 15"]
    assert len(hits) >= 2
    for i in hits:
        assert body[i - 1].strip().startswith("v_mfma_f32_32x32x16_bf16") and " a[" in body[i - 1], body[i - 1]
        assert body[i + 1].strip() == "s_nop 3", body[i + 1]


def test_operand_pass_and_attention_budgets(tmp_path):
    prep = _kernels(_asm("wino_prep2.hip", tmp_path))
    for name, k in prep.items():
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (name, k)
        assert k["vgpr_count"] <= 128, (name, k)                    # >= 4 waves per SIMD: the pass hides HBM latency by occupancy
    att = _kernels(_asm("attention.hip", tmp_path))
    (name, k), = att.items()
    assert k["vgpr_count"] <= 512 and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (name, k)
