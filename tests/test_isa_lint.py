"""tools/isa_lint.py (run by arreau_amd/build.py on every source with inline asm): each rule of its wait-state table on a
minimal ISA snippet -- the violation is reported, the padded form is not -- plus the control-flow search, the landed
markers of asm loads and the M0 / unmodelled-instruction checks.  CPU only: the lint reads text."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lint():
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "arreau_amd", "_isa_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run(lint, tmp_path, body, all_pairs=False):
    text = "\t.text\n_Z6kernelv:\n" + body + "\ts_endpgm\n.Lfunc_end0:\n"
    p = tmp_path / "k-hip-amdgcn-amd-amdhsa-gfx950.s"
    p.write_text(text)
    return sorted({v[1] for v in lint.lint_file(str(p), all_pairs)})


ASM = "\t;;#ASMSTART\n%s\n\t;;#ASMEND\n"


def test_wide_store_data_hazard(lint, tmp_path):
    """The bug of round 2 (edge_f16.hip K-tile stores): data registers of an asm dwordx3/x4 store rewritten too soon."""
    store = ASM % "\tglobal_store_dwordx3 v35, v[0:2], s[6:7] offset:0"
    assert run(lint, tmp_path, store + "\tv_add_u32_e32 v0, 0x80, v76\n") == ["wide-store-data"]
    assert run(lint, tmp_path, store + "\ts_mov_b32 s1, 0\n\tv_add_u32_e32 v1, 0x80, v76\n") == ["wide-store-data"]
    assert run(lint, tmp_path, store + "\ts_mov_b32 s1, 0\n\ts_mov_b32 s2, 0\n\tv_add_u32_e32 v1, 0x80, v76\n") == []
    padded = ASM % "\tglobal_store_dwordx4 v35, v[0:3], s[6:7] offset:0\n\ts_nop 3"
    assert run(lint, tmp_path, padded + "\tv_add_u32_e32 v0, 0x80, v76\n") == []
    # a 32-bit store has no such hazard; a rewrite of the ADDRESS register is not one either
    assert run(lint, tmp_path, ASM % "\tglobal_store_dword v35, v0, s[6:7]" + "\tv_mov_b32_e32 v0, 0\n\tv_mov_b32_e32 v35, 0\n") == []


def test_hazard_is_found_along_a_branch(lint, tmp_path):
    body = (ASM % "\tglobal_store_dwordx4 v35, v[0:3], s[6:7] offset:0" +
            "\ts_cbranch_scc1 .LBB0_2\n\ts_nop 7\n\ts_branch .LBB0_3\n.LBB0_2:\n\tv_mov_b32_e32 v2, 0\n.LBB0_3:\n")
    assert run(lint, tmp_path, body) == ["wide-store-data"]  # only the branch instruction lies between on the taken path
    assert run(lint, tmp_path, body.replace("\ts_cbranch_scc1", "\ts_nop 0\n\ts_cbranch_scc1")) == []


def test_valu_written_sgpr_read_by_asm_vmem(lint, tmp_path):
    wait = "\ts_waitcnt vmcnt(0)\n"  # (every copy is waited for: the LDS-DMA rule of round 4 is not what these snippets test)
    body = "\tv_readfirstlane_b32 s20, v0\n" + ASM % "\ts_add_u32 m0, s22, 0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v221, s[20:21]" + wait
    assert run(lint, tmp_path, body) == ["valu-sgpr-vmem"]  # 2 states, needs 5
    ok = "\tv_readfirstlane_b32 s20, v0\n\ts_nop 2\n" + ASM % "\ts_add_u32 m0, s22, 0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v221, s[20:21]" + wait
    assert run(lint, tmp_path, ok) == []
    # an SALU-written base needs nothing
    assert run(lint, tmp_path, "\ts_add_u32 s20, s20, 0x2000\n" + ASM % "\ts_add_u32 m0, s22, 0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v221, s[20:21]" + wait) == []


def test_m0_write_before_lds_dma(lint, tmp_path):
    wait = "\ts_waitcnt vmcnt(0)\n"
    assert run(lint, tmp_path, ASM % "\ts_mov_b32 m0, s12\n\tglobal_load_lds_dwordx4 v[6:7], off" + wait) == ["salu-m0-ldsdma"]
    assert run(lint, tmp_path, ASM % "\ts_mov_b32 m0, s12\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[6:7], off" + wait) == []


def test_compiler_use_of_m0_next_to_asm_that_overwrites_it(lint, tmp_path):
    body = ASM % "\ts_mov_b32 m0, s12\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[6:7], off" + "\ts_mov_b32 m0, -1\n\tds_read_b32 v1, v2\n"
    assert "m0-compiler-use" in run(lint, tmp_path, body)


def test_asm_load_destination_must_stay_untouched_until_its_wait(lint, tmp_path):
    load = ASM % "\tglobal_load_dwordx4 v[0:3], v[8:9], off"
    assert run(lint, tmp_path, load + "\tv_mov_b32_e32 v20, v1\n") == ["asm-load-dest"]          # copied before any wait
    assert run(lint, tmp_path, load + ASM % "\ts_waitcnt vmcnt(0)" + "\tv_mov_b32_e32 v20, v1\n") == []
    # counted wait: one younger VMEM operation in flight, vmcnt(1) covers the load, vmcnt(2) does not
    younger = ASM % "\tglobal_store_dword v35, v30, s[6:7]"
    assert run(lint, tmp_path, load + younger + ASM % "\ts_waitcnt vmcnt(1)" + "\tv_mov_b32_e32 v20, v1\n") == []
    assert run(lint, tmp_path, load + younger + ASM % "\ts_waitcnt vmcnt(2)" + "\tv_mov_b32_e32 v20, v1\n") == ["asm-load-dest"]
    # the source's landed marker ends the search (its waits are conditional and hand-counted: checked by the debug-wait twin)
    marker = ASM % "\t; landed v[0:3]"
    assert run(lint, tmp_path, load + "\ts_cbranch_scc1 .LBB0_9\n" + ASM % "\ts_waitcnt vmcnt(4)" + ".LBB0_9:\n" + marker +
               "\tv_mov_b32_e32 v20, v1\n") == []
    assert run(lint, tmp_path, load + "\tv_mov_b32_e32 v20, v1\n" + marker) == ["asm-load-dest"]  # touched in front of the marker


def test_trans_and_half_register_forwarding_into_asm(lint, tmp_path):
    assert run(lint, tmp_path, "\tv_exp_f32_e32 v5, v4\n" + ASM % "\tv_fma_mixlo_f16 v7, v6, s2, v5 op_sel_hi:[1,0,0]") == ["trans-valu"]
    assert run(lint, tmp_path, "\tv_exp_f32_e32 v5, v4\n\ts_nop 0\n" + ASM % "\tv_fma_mixlo_f16 v7, v6, s2, v5 op_sel_hi:[1,0,0]") == []
    mix = ASM % "\tv_fma_mixhi_f16 v7, v6, s2, v5 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
    assert run(lint, tmp_path, mix + "\tv_mfma_f32_16x16x32_f16 v[10:13], v[20:23], v[4:7], v[10:13]\n") == ["dstsel-forward"]
    assert run(lint, tmp_path, mix + "\ts_nop 0\n\tv_mfma_f32_16x16x32_f16 v[10:13], v[20:23], v[4:7], v[10:13]\n") == []


def test_mfma_result_read_by_asm(lint, tmp_path):
    mfma = "\tv_mfma_f32_16x16x32_f16 v[10:13], v[20:23], v[4:7], v[10:13]\n"
    store = ASM % "\tglobal_store_dwordx4 v35, v[10:13], s[6:7] offset:0\n\ts_nop 3"
    assert run(lint, tmp_path, mfma + "\ts_nop 3\n" + store) == ["mfma-result"]   # 4 states, needs 4 passes + 4
    assert run(lint, tmp_path, mfma + "\ts_nop 7\n" + store) == []


def test_spills_must_not_share_a_path_with_counted_waits(lint, tmp_path):
    spill = "\tscratch_store_dword off, v0, off\n"
    wait = ASM % "\ts_waitcnt vmcnt(4)"
    assert run(lint, tmp_path, spill + wait) == ["scratch-before-counted-wait"]
    assert run(lint, tmp_path, wait + spill) == ["scratch-after-counted-wait"]
    assert run(lint, tmp_path, spill + ASM % "\ts_waitcnt vmcnt(0)") == []  # a full wait is not a count
    # two roles that never meet: one spills, the other counts
    roles = ("\ts_cbranch_scc1 .LBB0_5\n" + spill + "\ts_endpgm\n.LBB0_5:\n" + wait)
    assert run(lint, tmp_path, roles) == []


def test_unmodelled_instruction_inside_asm_is_reported(lint, tmp_path):
    assert run(lint, tmp_path, ASM % "\tv_permlane32_swap_b32 v1, v2") == ["unmodelled-asm"]


def test_product_sources_are_lint_clean_and_the_build_runs_the_lint():
    """build.py compiles every asm source with -save-temps and fails on a violation (the check itself is exercised by
    __graft_entry__.build(); here: the wiring exists and names every source that contains inline asm)."""
    import re
    from arreau_amd import build as b
    csrc = b.CSRC
    with_asm = set()
    for src in b.SOURCES:
        text = open(os.path.join(csrc, src)).read()
        for hdr in re.findall(r'#include "([\w.]+)"', text):
            if hdr in ("f16x3.h",):  # the LDS-DMA / mixed-precision asm lives in this header
                text += open(os.path.join(csrc, hdr)).read()
        if re.search(r'\basm\s*(volatile)?\s*\(\s*"[^"]', text):
            with_asm.add(src)
    assert with_asm <= set(b.ASM_LINT), with_asm - set(b.ASM_LINT)
    assert callable(b._isa_lint().lint_file)


def test_lds_dma_copy_needs_a_covering_wait(lint, tmp_path):
    """Round 4: an LDS-DMA copy must meet a covering vmcnt wait on every path -- before the wave ends; a counted wait that leaves the copy among the N youngest does not cover it."""
    dma = ASM % "\ts_mov_b32 m0, s4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v1, s[2:3]"
    # waited, then published by the barrier: clean
    assert run(lint, tmp_path, dma + "\ts_waitcnt vmcnt(0)\n\ts_barrier\n") == []
    # the wave ends without ever waiting for its copy
    assert "ldsdma-unwaited-exit" in run(lint, tmp_path, dma + "\ts_barrier\n")
    # a counted wait: vmcnt(1) with ONE younger VMEM operation covers the copy, with none it does not
    younger = "\tglobal_load_dword v9, v[10:11], off\n"
    assert run(lint, tmp_path, dma + younger + (ASM % "\ts_waitcnt vmcnt(1)") + "\ts_barrier\n\ts_waitcnt vmcnt(0)\n") == []
    assert "ldsdma-unwaited-exit" in run(lint, tmp_path, dma + (ASM % "\ts_waitcnt vmcnt(1)") + "\ts_barrier\n")
    # a path around the wait (conditional branch) is found
    body = dma + "\ts_cbranch_scc1 .LBB0_2\n\ts_waitcnt vmcnt(0)\n.LBB0_2:\n\ts_barrier\n"
    assert "ldsdma-unwaited-exit" in run(lint, tmp_path, body)
    # a wait many barriers later is fine (deep rings; the depth is reported by --lds, not judged)
    assert run(lint, tmp_path, dma + "\ts_barrier\n" * 9 + "\ts_waitcnt vmcnt(0)\n") == []
    rows = lint.lds_protocol_summary(str(tmp_path / "k-hip-amdgcn-amd-amdhsa-gfx950.s"))
    assert rows and rows[0][3] == 1  # one LDS-DMA copy in the last snippet
