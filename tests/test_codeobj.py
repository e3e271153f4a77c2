"""The headline kernel must not spill: VERDICT round 2 found 124 spilled SGPRs (v_writelane / v_readlane are VALU
instructions in a VALU-bound kernel), 4 spilled VGPRs and 20 bytes of scratch behind a DESIGN.md that claimed none.
Reads the metadata note of the gfx950 code object inside libdsim_hip.so (tools/codeobj_meta.py; no GPU needed)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import codeobj_meta  # noqa: E402


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(codeobj_meta.DEFAULT_LIB):
        pytest.skip("libdsim_hip.so not built")
    ks = codeobj_meta.kernels()
    names = codeobj_meta.demangle([k["name"] for k in ks])
    return {n: k for n, k in zip(names, ks)}


def _find(kernels, prefix):
    hits = [k for n, k in kernels.items() if n.replace("void ", "").startswith(prefix)]
    assert hits, "no kernel %r in the code object" % prefix
    return hits


def test_cooperative_mesh_kernels_do_not_spill(kernels):
    for prefix in ("walk_kernel_coop4<1, 1>", "walk_kernel_coop4<2, 1>", "walk_kernel_coop4<4, 1>", "walk_kernel_coop4_traj"):
        for k in _find(kernels, prefix):
            assert k.get("private_segment_fixed_size", 0) == 0, (prefix, k)
            assert k.get("vgpr_spill_count", 0) == 0, (prefix, k)
            assert k.get("sgpr_spill_count", 0) <= 16, (prefix, k)
            assert k["vgpr_count"] <= 128, (prefix, k)  # four wavefronts per SIMD


def test_free_and_analytic_geometry_kernels_use_no_scratch(kernels):
    for geom in range(4):
        for k in _find(kernels, "walk_kernel<%d," % geom):
            assert k.get("private_segment_fixed_size", 0) == 0 and k.get("vgpr_spill_count", 0) == 0, k


def test_no_kernel_of_the_library_spills(kernels):
    """the fall-back kernels (per-lane mesh traversal, the 32-accumulator recording kernel) kept
    ~100 dwords of launch parameters and eight levels of saved execution masks in SGPRs and spilled 107-187 of them.
    Every kernel of the code object - not only the headline one - now stays within 16 spilled SGPRs, spills no VGPR and
    uses no scratch memory."""
    assert len(kernels) > 50
    bad = {n: (k.get("sgpr_spill_count", 0), k.get("vgpr_spill_count", 0), k.get("private_segment_fixed_size", 0))
           for n, k in kernels.items()
           if k.get("sgpr_spill_count", 0) > 16 or k.get("vgpr_spill_count", 0) > 0 or k.get("private_segment_fixed_size", 0) > 0}
    assert not bad, bad


def test_no_vector_write_right_before_a_dpp_read():
    """row_bcast / row_fmac issue `v_mov_b64_dpp` / `v_fmac_f64_dpp row_newbcast` through inline assembly,
    which the compiler's hazard recogniser does not see; the two wait states the hardware needs between a VALU write of
    the broadcast register and its DPP read hold only while no copy of that register is scheduled between dpp_fence
    (s_nop 1) and the instruction.  Checked on the shipped code object: every *_dpp instruction of the library."""
    if not os.path.exists(codeobj_meta.DEFAULT_LIB):
        pytest.skip("libdsim_hip.so not built")
    dis = codeobj_meta.disassembly()
    assert dis.count("_dpp") > 100  # the phase kernels' row broadcasts and the prefix sums are there
    assert codeobj_meta.dpp_hazards() == []
