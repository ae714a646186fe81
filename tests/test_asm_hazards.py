"""CPU-side fence around the hand-counted wait states (runs in the build container, no GPU): the gfx950 code objects of the
built library are disassembled (llvm-objdump) and every hazard that lives inside an opaque asm statement - where hipcc's
hazard recognizer and waitcnt insertion cannot see - is checked in the instruction stream that will actually run:

  * SGPR-base VMEM (store16_uniform's global_store_dwordx4, Pipe::dma_phase's global_load_lds_dwordx4): >= 5 wait states
    after the v_readfirstlane_b32 that produced the base (round 1 aborted on the GPU box before those were spelled out,
    HISTORY.md section 4.6);
  * M0 written one instruction + nop before the LDS-DMA that reads it;
  * no VALU write to a > 8-byte store's data registers in the following wait state (the round-1 data corruption);
  * the explicit LDS read pipeline of mlp_forward48_kernel / mlp_backward48_kernel: nothing names a fragment register between its ds_read and the
    counted s_waitcnt that retires it (the compiler sees the fragment as an ordinary value from the read-asm on; a phi copy at
    a control-flow merge once read them early - caught on the GPU by the geometry test, now caught here first).
"""
import os
import sys
import tempfile

import pytest

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "scripts"))
import codeobj  # noqa: E402

LIB = codeobj.DEFAULT_LIB


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        pytest.skip("libdexnerf_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for co in codeobj.extract_code_objects(LIB, d):
            out.update(codeobj.disassemble(co))
    assert any("mlp_forward48_kernel" in k for k in out) and any("mlp_backward_kernel" in k for k in out)
    return out


def mlp_kernels(kernels):
    return {k: v for k, v in kernels.items() if any(t in k for t in ("mlp_forward", "mlp_backward", "weight_grad"))}


def test_sgpr_base_vmem_has_five_wait_states(kernels):
    n_checked = 0
    for name, ins in mlp_kernels(kernels).items():
        bad = codeobj.sgpr_base_vmem_violations(ins, need=5)
        assert not bad, (name, bad[:3])
        n_checked += sum(1 for line in ins if line.startswith(("global_store_dwordx4", "global_load_lds_dwordx4")))
    assert n_checked > 1000   # the streams are there (this is not vacuous)


def test_m0_is_settled_before_lds_dma(kernels):
    for name, ins in mlp_kernels(kernels).items():
        bad = codeobj.m0_lds_dma_violations(ins)
        assert not bad, (name, bad[:3])


def test_wide_store_data_registers_are_not_rewritten_at_once(kernels):
    for name, ins in mlp_kernels(kernels).items():
        bad = codeobj.wide_store_data_violations(ins)
        assert not bad, (name, bad[:3])


def test_explicit_lds_read_pipeline_is_not_touched_in_flight(kernels):
    seen = 0
    for name, ins in kernels.items():
        if "mlp_forward48_kernel" not in name and "mlp_backward48_kernel" not in name:
            continue
        seen += 1
        bad = codeobj.lds_read_violations(ins)
        assert not bad, (name, bad[:4])
        # the pipeline is really the explicit one: counted waits dominate, the hot loop has no lgkmcnt(0) after a fresh read
        counted = sum(1 for line in ins if line.startswith("s_waitcnt lgkmcnt(1)") or line.startswith("s_waitcnt lgkmcnt(2)"))
        assert counted > 150, (name, counted)
    assert seen == 28   # forward: 2 widths x {fixed, run-time shape} x {bf16, fp16, bf16 training} + the overlapped-encoding instances of the as-shipped and the paper nets x {bf16, fp16} + the self-compositing fixed-shape instances (2 widths x {bf16, fp16}) + the two-point-group training instances (2 widths); backward: 2 widths x {fixed, run-time, fixed on two point groups}


def test_mfma_results_are_read_after_their_wait_states(kernels):
    """Every MFMA kernel, compiler-scheduled or not: no vector instruction reads an MFMA's destination inside the 7 wait states a
    4-pass result needs.  Load-bearing for the explicit-schedule render instances, whose MFMAs the hazard recognizer cannot see."""
    xs = 0
    for name, ins in mlp_kernels(kernels).items():
        bad = codeobj.mfma_result_read_violations(ins)
        assert not bad, (name, bad[:3])
        if any(f"mlp_forward48_kernel<256, {f}, 8, 16u, 1, 0, {o}, 0>" in name for f in (1, 2) for o in (0, 2)):
            xs += 1
            # the explicit schedule really is in place: no hazard padding in front of a conversion, at most one vector instruction
            # between two MFMAs of the trunk
            conv = [i for i, line in enumerate(ins) if line.startswith(("v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32"))]
            behind_mfma = sum(1 for i in conv if ins[i - 1].startswith("v_mfma"))
            assert behind_mfma > 0.8 * len(conv) and len(conv) > 900, (name, behind_mfma, len(conv))
    assert xs == 4      # plain and in-stage-encoding instance, bf16 and fp16


def test_the_checkers_catch_planted_hazards():
    """The checks above are only worth something if they fire: plant each hazard in a tiny stream."""
    assert codeobj.sgpr_base_vmem_violations(["v_readfirstlane_b32 s4, v1", "s_nop 2", "global_store_dwordx4 v0, v[4:7], s[4:5]"])
    assert not codeobj.sgpr_base_vmem_violations(["v_readfirstlane_b32 s4, v1", "s_nop 4", "global_store_dwordx4 v0, v[4:7], s[4:5]"])
    assert codeobj.m0_lds_dma_violations(["s_mov_b32 m0, s3", "global_load_lds_dwordx4 v1, s[0:1]"])
    assert not codeobj.m0_lds_dma_violations(["s_mov_b32 m0, s3", "s_nop 1", "global_load_lds_dwordx4 v1, s[0:1]"])
    assert codeobj.wide_store_data_violations(["global_store_dwordx4 v0, v[4:7], s[4:5] nt", "v_mov_b32_e32 v5, 0"])
    assert not codeobj.wide_store_data_violations(["global_store_dwordx4 v0, v[4:7], s[4:5] nt", "s_nop 1", "v_mov_b32_e32 v5, 0"])
    stream = ["ds_read_b128 v[8:11], v1", "ds_read_b128 v[12:15], v1 offset:1024", "s_waitcnt lgkmcnt(1)",
              "v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[20:23], v[0:3]"]
    assert not codeobj.lds_read_violations(stream)
    assert codeobj.lds_read_violations(stream[:2] + ["v_mov_b64_e32 v[30:31], v[12:13]"] + stream[2:])   # the phi copy
    assert codeobj.lds_read_violations(stream[:2] + ["s_waitcnt lgkmcnt(2)"] + stream[3:])               # wait too loose
    assert not codeobj.lds_read_violations(stream[:2] + ["s_branch 12", "v_mov_b64_e32 v[30:31], v[12:13]"])   # an out-of-line block behind a jump
    m = "v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[20:23], v[0:3]"
    assert codeobj.mfma_result_read_violations([m, "s_nop 3", "v_cvt_pk_bf16_f32 v40, v0, v1"])
    assert not codeobj.mfma_result_read_violations([m, "s_nop 6", "v_cvt_pk_bf16_f32 v40, v0, v1"])
    assert not codeobj.mfma_result_read_violations([m, m.replace("v[0:3]", "v[4:7]"), "ds_read_b128 v[8:11], v90", "s_waitcnt lgkmcnt(1)", m.replace("v[0:3]", "v[12:15]"),
                                                    "v_cvt_pk_bf16_f32 v40, v0, v1"])
