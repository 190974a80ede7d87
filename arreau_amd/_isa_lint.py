#!/usr/bin/env python3
"""Hazard lint for gfx950 device ISA (`hipcc -save-temps` output, *-hip-amdgcn-amd-amdhsa-gfx950.s).

hipcc treats an inline-asm statement as ONE opaque instruction: it neither pads the software-inserted wait states the
CDNA3/4 ISA asks for around the instructions inside the string, nor counts their memory operations.  Round 2 shipped
(and then found by luck) exactly such a bug: the hand-written K-tile stores of edge_f16.hip are VMEM stores of more
than 64 bits, which read their data registers after issue, and the compiler re-used those registers in the next
instruction.  This lint makes that class of bug a BUILD failure (arreau_amd/build.py runs it on every source that
contains inline asm):

  * every instruction pair of the wait-state table below in which the producer OR the consumer comes from inline asm
    (between `;;#ASMSTART` and `;;#ASMEND`) must be separated by the required number of wait states, along EVERY
    control-flow path (backward search over the function's CFG: labels, s_branch / s_cbranch_*);
  * an inline-asm VMEM load with a VGPR destination: along every path no instruction may read or write the destination
    registers until a `s_waitcnt vmcnt(N)` that covers the load (vmcnt retires in issue order: N <= VMEM operations
    issued after it), or until a LANDED MARKER naming them -- `asm volatile("; landed %0" : "+v"(x))`, which the source
    places right behind its (possibly conditional, hand-counted) waits.  The marker is the author's claim that the waits
    in front of it cover the load; the claim itself is checked on the GPU by the debug-wait twin library (every counted
    wait -> vmcnt(0), outputs bit-identical).  What the lint proves is the part no test can: that the COMPILER did not
    copy, spill or re-use the destination registers between the load and that point;
  * register spills (scratch_*) must not share a control-flow path with an inline-asm counted wait `vmcnt(N > 0)` (they
    are VMEM operations in the same in-order queue); a role of waves that never counts may spill;
  * M0: in a function whose asm writes M0, every compiler instruction naming m0 is a violation (the asm does not
    restore it);
  * LDS-DMA copies (round 4, check_lds_dma): along every path a copy meets a covering `s_waitcnt vmcnt` before the wave
    ends -- a copy nobody waits for is published by no barrier (`--lds` prints the
    per-kernel LDS protocol summary: reads, writes, copies, barriers, counted waits, how many barriers ahead the rings run);
  * `--all` applies the table to compiler-only pairs too: it must report nothing (self-check of the table and of the
    parser against what LLVM's own hazard recognizer pads).

Wait states (gfx940 family = gfx950; LLVM GCNHazardRecognizer, CDNA3 ISA guide 4.5 "manually inserted wait states"):
    wide-store-data   VMEM/FLAT store of > 64 bits            -> VALU write of its data VGPRs                2
    valu-sgpr-vmem    VALU writes SGPR / VCC                  -> VMEM (incl. LDS-DMA) reads that SGPR         5
    salu-m0-ldsdma    SALU writes M0                          -> LDS-DMA / GDS / s_sendmsg                    1
    valu-vgpr-rdlane  VALU writes VGPR                        -> v_readlane / v_readfirstlane reads it        1
    valu-sgpr-lanesel VALU writes SGPR                        -> v_readlane / v_writelane lane select         4
    valu-sgpr-valu    VALU writes SGPR / VCC                  -> VALU reads it (constant, carry, mask)        2
    valu-vgpr-dpp     VALU writes VGPR                        -> DPP instruction reads it                     2
    valu-exec-dpp     VALU writes EXEC                        -> DPP instruction                              5
    trans-valu        v_exp/log/rcp/rsq/sqrt/sin/cos          -> non-transcendental VALU reads the result     1
    dstsel-forward    VALU writing half a register (mixlo/hi) -> VALU / MFMA reads the register               1
    mfma-result       MFMA writes D                           -> non-MFMA instruction reads / writes D        passes + 2 (+ 2, K >= 16)
Instruction kinds inside inline asm that the table does not model (anything but the ones _ASM_MODELLED lists) are
reported as `unmodelled-asm`: extend the lint before shipping a new kind of hand-written instruction.
A wait state = one issued instruction (s_nop N = N + 1).

    python tools/isa_lint.py file.s [...] [--all] [--quiet]
"""
import re
import sys
from collections import defaultdict

# ---------------------------------------------------------------------------------------------------------------------
# parsing
_REG_RANGE = re.compile(r"^([vsa])\[(\d+):(\d+)\]$")
_REG_ONE = re.compile(r"^([vsa])(\d+)$")
_SPECIAL = {"vcc": ("vcc_lo", "vcc_hi"), "vcc_lo": ("vcc_lo",), "vcc_hi": ("vcc_hi",), "exec": ("exec_lo", "exec_hi"),
            "exec_lo": ("exec_lo",), "exec_hi": ("exec_hi",), "m0": ("m0",), "scc": ("scc",)}
_TRANS = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f32|f16|legacy_f32)")
_WIDE_STORE = re.compile(r"^(global|flat|buffer|scratch)_store_(dwordx3|dwordx4|format_xyzw?|format_d16_xyzw)|_atomic_cmpswap_x2")
_VMEM = re.compile(r"^(global|flat|buffer|scratch|tbuffer|image)_")
_BRANCH = re.compile(r"^s_(branch|cbranch_\w+)$")


def regs_of(tok):
    """Register names (one per 32-bit register) an operand token stands for; () for literals / modifiers."""
    tok = tok.strip()
    if tok.startswith("-") or tok.startswith("|"):
        tok = tok.strip("-|")
    m = re.match(r"^(neg|abs)\((.*)\)$", tok)
    if m:
        tok = m.group(2)
    m = _REG_RANGE.match(tok)
    if m:
        return tuple(f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1))
    m = _REG_ONE.match(tok)
    if m:
        return (tok,)
    return _SPECIAL.get(tok, ())


class Ins:
    __slots__ = ("idx", "line", "text", "mn", "ops", "in_asm", "defs", "uses", "is_valu", "is_salu", "is_vmem", "is_mfma",
                 "is_trans", "is_dpp", "is_ldsdma", "is_lds", "wide_store_data", "mfma_passes", "target", "falls", "states",
                 "vmcnt", "vmem_dest", "lanesel", "partial_dst", "is_rdlane")

    def __repr__(self):
        return f"{self.line}: {self.text}"


def _split_ops(rest):
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def _mfma_passes(mn):
    # v_mfma_f32_MxNxK_type: passes of 4 cycles each (MI355X_MICROARCH.md: 32x32x16 16-bit = 32 cycles, 16x16x32 = 16,
    # 32x32x2 f32 = 64, 16x16x4 f32 = 32, 4x4 = 8)
    m = re.match(r"^v_s?mfma_\w+?_(\d+)x(\d+)x(\d+)", mn)
    if not m:
        return 16
    a, _, k = int(m.group(1)), int(m.group(2)), int(m.group(3))
    if "f8f6f4" in mn:  # block-scaled K = 128 / 64 forms: fp8 operands take twice the passes of fp6 / fp4 (measured: 16x16x128 e4m3
        return 16 if a == 32 else 8  # = 36 clocks against 18 for 16x16x32 f16, tools/exp/fp8_mfma_check.hip); the larger count
    if a == 32:
        return 16 if k <= 2 else 8
    if a == 16:
        return 8 if k <= 4 else 4
    return 2


def parse_ins(idx, lineno, text, in_asm):
    ins = Ins()
    ins.idx, ins.line, ins.text, ins.in_asm = idx, lineno, text, in_asm
    parts = text.split(None, 1)
    mn = ins.mn = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    # trailing modifiers (offset:, op_sel:, nt, sc0, row_shr:, ...) are separated by spaces, not commas
    toks = _split_ops(rest)
    ops = []
    mods = ""
    for t in toks:
        first = t.split()[0] if t.split() else ""
        ops.append(first)
        mods += " " + " ".join(t.split()[1:])
    ins.ops = ops
    ins.is_mfma = mn.startswith(("v_mfma", "v_smfmac"))
    ins.mfma_passes = _mfma_passes(mn) if ins.is_mfma else 0
    ins.is_valu = mn.startswith("v_") and not mn.startswith("v_nop")
    ins.is_salu = mn.startswith("s_") and not mn.startswith(("s_load", "s_buffer_load", "s_store", "s_waitcnt", "s_nop", "s_barrier",
                                                            "s_endpgm", "s_branch", "s_cbranch", "s_sleep", "s_sendmsg",
                                                            "s_dcache", "s_icache", "s_setprio", "s_trap", "s_code_end"))
    ins.is_vmem = bool(_VMEM.match(mn))
    ins.is_lds = mn.startswith("ds_")
    ins.is_trans = bool(_TRANS.match(mn))
    ins.is_dpp = "_dpp" in mn or any(k in rest for k in ("quad_perm:", "row_shl:", "row_shr:", "row_ror:", "row_bcast:", "row_mirror",
                                                        "row_half_mirror", "wave_shl:", "wave_shr:", "wave_ror:", "wave_rol:",
                                                        "row_newbcast:"))
    ins.is_ldsdma = ins.is_vmem and ("_load_lds_" in mn or re.search(r"(^|\s)lds(\s|$)", rest) is not None)
    ins.is_rdlane = mn.startswith(("v_readlane", "v_readfirstlane"))
    ins.partial_dst = mn.startswith(("v_fma_mixlo", "v_fma_mixhi", "v_mad_mixlo", "v_mad_mixhi")) or "dst_sel:WORD" in rest \
        or "dst_sel:BYTE" in rest
    ins.states = 1
    ins.vmcnt = None
    ins.target = None
    ins.falls = True
    ins.wide_store_data = ()
    ins.vmem_dest = ()
    ins.lanesel = ()
    defs, uses = set(), set()
    R = [regs_of(o) for o in ops]

    if mn == "s_nop":
        ins.states = int(ops[0], 0) + 1 if ops else 1
    elif mn == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", rest)
        if m:
            ins.vmcnt = int(m.group(1))
        elif re.match(r"^\s*(0x[0-9a-f]+|\d+)\s*$", rest):  # raw immediate: gfx9 vmcnt = bits 3:0 | bits 15:14 << 4
            v = int(rest.strip(), 0)
            ins.vmcnt = (v & 0xF) | ((v >> 14) & 0x3) << 4
    elif _BRANCH.match(mn):
        ins.target = ops[0] if ops else None
        ins.falls = mn != "s_branch"
        if "vccz" in mn or "vccnz" in mn:
            uses.update(_SPECIAL["vcc"])
        if "execz" in mn or "execnz" in mn:
            uses.update(_SPECIAL["exec"])
        if "scc" in mn:
            uses.add("scc")
    elif mn in ("s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap"):
        ins.falls = mn == "s_trap"
        for r in R:
            uses.update(r)
    elif ins.is_vmem or ins.is_lds:
        is_store = "_store" in mn or mn.startswith(("ds_write", "ds_gws"))
        is_atomic = "atomic" in mn or mn.startswith(("ds_add", "ds_sub", "ds_min", "ds_max", "ds_and", "ds_or", "ds_xor",
                                                     "ds_cmpst", "ds_wrxchg", "ds_inc", "ds_dec", "ds_append", "ds_consume"))
        returns = (not is_store and not ins.is_ldsdma) and (not is_atomic or "_rtn" in mn or " sc0" in (" " + rest) or " glc" in (" " + rest))
        start = 0
        if returns and R and R[0] and R[0][0][0] in "va":
            defs.update(R[0])
            if ins.is_vmem:
                ins.vmem_dest = R[0]
            start = 1
        for r in R[start:]:
            uses.update(r)
        if ins.is_ldsdma or mn.startswith(("ds_gws", "ds_read_addtid", "ds_write_addtid")) or " gds" in (" " + rest):
            uses.add("m0")
        if _WIDE_STORE.search(mn):
            # global_store  vaddr, vdata, saddr | flat_store vaddr, vdata | buffer_store vdata, vaddr, srsrc, soffset
            data = R[0] if mn.startswith(("buffer_", "tbuffer_")) else (R[1] if len(R) > 1 else ())
            ins.wide_store_data = tuple(r for r in data if r[0] in "va")
    elif mn.startswith(("s_load", "s_buffer_load", "s_scratch_load")):
        defs.update(R[0] if R else ())
        for r in R[1:]:
            uses.update(r)
    elif mn.startswith("s_"):
        if mn.startswith(("s_cmp", "s_bitcmp")):
            defs.add("scc")
            for r in R:
                uses.update(r)
        elif mn.startswith(("s_waitcnt", "s_barrier", "s_sleep", "s_setprio", "s_sendmsg", "s_code_end", "s_dcache", "s_icache",
                            "s_set_gpr_idx", "s_setvskip", "s_sethalt", "s_incperflevel", "s_decperflevel", "s_ttrace")):
            if mn.startswith("s_sendmsg"):
                uses.add("m0")
        else:
            if R:
                defs.update(R[0])
            for r in R[1:]:
                uses.update(r)
            if mn.startswith(("s_cselect", "s_cmov", "s_addc", "s_subb")):
                uses.add("scc")
            if mn.startswith(("s_add", "s_sub", "s_and", "s_or", "s_xor", "s_nand", "s_nor", "s_xnor", "s_andn2", "s_orn2", "s_lshl", "s_lshr",
                              "s_ashr", "s_min", "s_max", "s_abs", "s_bfe", "s_not", "s_wqm", "s_quadmask", "s_bcnt", "s_absdiff",
                              "s_mul_i32")) and not mn.startswith("s_mul"):
                defs.add("scc")
            if "saveexec" in mn:
                defs.update(_SPECIAL["exec"])
                uses.update(_SPECIAL["exec"])
                defs.add("scc")
    elif ins.is_valu:
        nd = 1
        if mn.startswith("v_cmpx"):
            defs.update(_SPECIAL["exec"])
            if not mn.endswith("_e32") and R and R[0] and R[0][0][0] == "s":
                defs.update(R[0])
            elif R and ops[0] in ("vcc", "exec"):
                pass
            for r in R[(1 if ops and (ops[0] in ("vcc", "exec") or (R[0] and R[0][0][0] == "s")) else 0):]:
                uses.update(r)
            nd = None
        elif mn.startswith("v_cmp"):
            if R:
                defs.update(R[0])
            for r in R[1:]:
                uses.update(r)
            nd = None
        elif mn.startswith(("v_readlane", "v_readfirstlane")):
            defs.update(R[0])
            uses.update(R[1] if len(R) > 1 else ())
            if len(R) > 2:
                uses.update(R[2])
                ins.lanesel = R[2]
            nd = None
        elif mn.startswith("v_writelane"):
            defs.update(R[0])
            uses.update(R[0])
            for r in R[1:]:
                uses.update(r)
            if len(R) > 2:
                ins.lanesel = R[2]
            nd = None
        elif mn.startswith("v_swap"):
            for r in R:
                defs.update(r)
                uses.update(r)
            nd = None
        elif re.match(r"^v_(add|sub|subrev)_co_|^v_(addc|subb|subbrev)_co_|^v_div_scale|^v_mad_[ui]64_[ui]32", mn):
            nd = 2
        if nd is not None:
            for r in R[:nd]:
                defs.update(r)
            for r in R[nd:]:
                uses.update(r)
            if ins.is_mfma or mn.startswith(("v_fmac", "v_mac", "v_dot2c", "v_dot4c", "v_dot8c", "v_pk_fmac")):
                uses.update(R[0] if R else ())
            if mn.startswith("v_div_fmas"):
                uses.update(_SPECIAL["vcc"])
        # every VALU executes under EXEC; not modelled as a use (EXEC hazards are checked by rule valu-exec-dpp only)
    ins.defs, ins.uses = frozenset(defs), frozenset(uses)
    return ins


def parse_functions(path):
    """[(name, [Ins], {label: idx})] -- one entry per function of the text section."""
    funcs, cur, labels, name = [], None, None, None
    in_asm = False
    with open(path) as fh:
        for lineno, raw in enumerate(fh, 1):
            line = raw.split(";", 1)[0] if not raw.lstrip().startswith(";;#") else raw
            s = line.strip()
            if raw.lstrip().startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if raw.lstrip().startswith(";;#ASMEND"):
                in_asm = False
                continue
            if in_asm and cur is not None and raw.strip().startswith("; landed"):
                mk = parse_ins(len(cur), lineno, "s_nop 0", True)
                mk.mn, mk.text, mk.states = "landed", raw.strip(), 0
                mk.ops = _split_ops(raw.strip()[len("; landed"):])
                cur.append(mk)
                continue
            if not s or s.startswith("//"):
                continue
            m = re.match(r"^([A-Za-z_.$][\w.$]*):", s)
            if m:
                lab = m.group(1)
                if not lab.startswith(".L"):
                    if cur:
                        funcs.append((name, cur, labels))
                    name, cur, labels = lab, [], {}
                elif cur is not None:
                    labels[lab] = len(cur)
                continue
            if s.startswith("."):
                if s.startswith((".section", ".rodata", ".data", ".amdhsa_kernel", ".amdgpu_metadata")) and cur:
                    funcs.append((name, cur, labels))
                    name, cur, labels = None, None, None
                continue
            if cur is None:
                continue
            if not re.match(r"^[a-z]", s):
                continue
            cur.append(parse_ins(len(cur), lineno, s, in_asm))
    if cur:
        funcs.append((name, cur, labels))
    return [(n, c, l) for n, c, l in funcs if n and c]


# ---------------------------------------------------------------------------------------------------------------------
# control flow
def build_preds(code, labels):
    preds = defaultdict(list)
    for i, ins in enumerate(code):
        if ins.falls and i + 1 < len(code):
            preds[i + 1].append(i)
        if ins.target is not None and ins.target in labels:
            t = labels[ins.target]
            if t < len(code):
                preds[t].append(i)
    return preds


def build_succs(code, labels):
    succs = defaultdict(list)
    for i, ins in enumerate(code):
        if ins.falls and i + 1 < len(code):
            succs[i].append(i + 1)
        if ins.target is not None and ins.target in labels and labels[ins.target] < len(code):
            succs[i].append(labels[ins.target])
    return succs


def producers_within(code, preds, c_idx, limit):
    """(producer index, wait states between producer and code[c_idx]) for every instruction reachable backwards with
    fewer than `limit` wait states in between (minimum over paths)."""
    best = {}
    stack = [(p, 0) for p in preds[c_idx]]
    while stack:
        i, between = stack.pop()
        if between >= limit:
            continue
        if i in best and best[i] <= between:
            continue
        best[i] = between
        nxt = between + code[i].states
        for p in preds[i]:
            stack.append((p, nxt))
    return best.items()


# ---------------------------------------------------------------------------------------------------------------------
# rules
def _is_sgpr(r):
    return r[0] == "s" and r != "scc" or r.startswith("vcc")


def _is_vgpr(r):
    return r[0] in "va" and not r.startswith("vcc")


MAX_STATES = 24


def rule_hits(P, C):
    """[(rule, required states, registers)] for producer P followed by consumer C."""
    hits = []
    if P.wide_store_data and C.is_valu:  # (a load's write-back cannot arrive inside the window: LLVM checks VALU only)
        regs = set(P.wide_store_data) & C.defs
        if regs:
            hits.append(("wide-store-data", 2, regs))
    if P.is_valu:
        sg = {r for r in P.defs if _is_sgpr(r)}
        vg = {r for r in P.defs if _is_vgpr(r)}
        if sg:
            if C.is_vmem and sg & C.uses:
                hits.append(("valu-sgpr-vmem", 5, sg & C.uses))
            if C.lanesel and sg & set(C.lanesel):
                hits.append(("valu-sgpr-lanesel", 4, sg & set(C.lanesel)))
            if C.is_valu and sg & C.uses:
                hits.append(("valu-sgpr-valu", 2, sg & C.uses))
        if vg:
            if C.is_rdlane and vg & C.uses:
                hits.append(("valu-vgpr-rdlane", 1, vg & C.uses))
            if C.is_dpp and vg & C.uses:
                hits.append(("valu-vgpr-dpp", 2, vg & C.uses))
            if P.is_trans and C.is_valu and not C.is_trans and vg & C.uses:
                hits.append(("trans-valu", 1, vg & C.uses))
            if P.partial_dst and C.is_valu and vg & C.uses:
                hits.append(("dstsel-forward", 1, vg & C.uses))
        if C.is_dpp and P.defs & set(_SPECIAL["exec"]):
            hits.append(("valu-exec-dpp", 5, P.defs & set(_SPECIAL["exec"])))
    if P.is_salu and "m0" in P.defs and (C.is_ldsdma or ("m0" in C.uses and not C.is_salu and not C.is_valu)):
        hits.append(("salu-m0-ldsdma", 1, {"m0"}))
    if P.is_mfma and not C.is_mfma:
        # (MFMA -> MFMA dependencies have their own, shorter table; no MFMA is written in inline asm here -- one that
        # is gets reported as `unmodelled-asm` below -- so those pairs are always the compiler's own)
        touched = set(P.defs) & (C.uses | C.defs)
        if touched:
            # LLVM: passes + 2 (+ 1 on gfx950); the gfx950 double-rate shapes (K >= 16) are padded one more by hipcc 7.2
            m = re.search(r"x(\d+)_", P.mn)
            need = P.mfma_passes + (4 if m and int(m.group(1)) >= 16 else 2)
            hits.append(("mfma-result", need, touched))
    return hits


def check_waitstates(name, code, preds, all_pairs):
    out = []
    for C in code:
        if not (C.uses or C.defs):
            continue
        for p_idx, between in producers_within(code, preds, C.idx, MAX_STATES):
            P = code[p_idx]
            if not (all_pairs or P.in_asm or C.in_asm):
                continue
            if not P.defs and not P.wide_store_data:
                continue
            for rule, need, regs in rule_hits(P, C):
                if between < need:
                    # an intervening redefinition does not cure a hazard, so none is looked for
                    out.append((name, rule, need, between, P, C, sorted(regs)))
    return out


def check_asm_loads(name, code, succs):
    """Inline-asm VMEM loads with VGPR destinations: untouched until a covering s_waitcnt vmcnt."""
    out = []
    CAP = 64
    for L in code:
        if not (L.in_asm and L.vmem_dest):
            continue
        dest = set(L.vmem_dest)
        seen = set()
        stack = [(s, 0) for s in succs[L.idx]]
        steps = 0
        while stack and steps < 200000:
            i, younger = stack.pop()
            key = (i, min(younger, CAP))
            if key in seen:
                continue
            seen.add(key)
            steps += 1
            ins = code[i]
            if ins.vmcnt is not None and ins.vmcnt <= younger:
                continue  # the load has retired on this path
            if ins.mn == "landed":
                named = set()
                for o in ins.ops:
                    named.update(regs_of(o))
                if dest <= named:
                    continue  # the source's waits in front of this marker cover the load (debug-wait twin checks that)
            if ins.mn == "s_endpgm":
                continue
            touched = dest & (ins.uses | ins.defs)
            if touched and ins is not L:
                out.append((name, "asm-load-dest", 0, younger, L, ins, sorted(touched)))
                continue
            if ins is L:
                continue  # back at the load itself (loop): its previous instance was covered on this path or is flagged
            y = younger + (1 if (ins.is_vmem) else 0)
            for s in succs[i]:
                stack.append((s, y))
    return out


def check_lds_dma(name, code, succs, report=None):
    """Cross-wave LDS rule (round 4), the part of it a static pass can decide.  An LDS-DMA copy (global_load_lds_*) lands in
    LDS some time after issue; OTHER waves read the landing zone, so the protocol is: the issuing wave waits for the copy
    (`s_waitcnt vmcnt(N)`, N <= VMEM operations it issued after the copy: vmcnt retires in order) and only then meets the
    `s_barrier` that publishes it.  Along EVERY control-flow path from every copy this checks that such a covering wait is
    reached before the wave ends (`ldsdma-unwaited-exit`): the hardware's implicit wait at s_endpgm would save the memory
    safety, but a copy nobody waited for was published by no barrier, and a landing zone must not outlive its workgroup's
    LDS allocation on the strength of an implicit rule -- every kernel that issues copies ends in an explicit vmcnt(0).
    NOT checked, because a path search over the CFG cannot: "at most RING barriers between a copy and its covering wait".
    The ring kernels branch on loop-invariant conditions (`has_next`: counted wait with copies following, or vmcnt(0) and
    no more copies); a path that takes one arm at the wait and the other at the copy is infeasible, but the CFG holds it,
    and along it the counted waits never cover anything.  (The search simply stops after RING_MAX barriers.)  Whether the
    COUNT of a wait is right is checked on the GPU by the debug-wait twin library (every counted wait -> vmcnt(0), outputs
    bit-identical); that each landing zone's READERS sit behind the publishing barrier needs address reasoning a lint over
    register names cannot do and stays an argument in the source (ring protocols of edge_f16.hip / conv_proj.hip).
    `report`, if given, receives (function, line of the copy, max barriers passed before its covering wait) per copy."""
    out = []
    RING_MAX, CAP = 16, 96
    for D in code:
        if not D.is_ldsdma:
            continue
        worst = 0
        seen = set()
        stack = [(s, 0, 0) for s in succs[D.idx]]
        steps = 0
        flagged = False
        while stack and steps < 400000 and not flagged:
            i, younger, bars = stack.pop()
            key = (i, min(younger, CAP), bars)
            if key in seen:
                continue
            seen.add(key)
            steps += 1
            ins = code[i]
            if ins.vmcnt is not None and ins.vmcnt <= younger:
                worst = max(worst, bars)
                continue  # covered on this path
            if ins.mn == "s_endpgm":
                out.append((name, "ldsdma-unwaited-exit", 0, bars, D, ins, []))
                flagged = True
                continue
            if ins.mn == "s_barrier":
                bars += 1
                if bars > RING_MAX:
                    continue  # (stop exploring: see the docstring -- not a finding)
            y = younger + (1 if ins.is_vmem else 0)
            for nx in succs[i]:
                stack.append((nx, y, bars))
        if report is not None:
            report.append((name, D.line, worst))
    return out


def lds_protocol_summary(path):
    """Per kernel: how many LDS reads / writes / LDS-DMA copies / barriers / counted waits the ISA holds, and for the copies the
    largest number of barriers between issue and covering wait (= how far ahead the ring runs).  Printed by `--lds`."""
    rows = []
    for name, code, labels in parse_functions(path):
        succs = build_succs(code, labels)
        rep = []
        bad = check_lds_dma(name, code, succs, rep)
        n_rd = sum(1 for i in code if i.is_lds and i.mn.startswith("ds_read"))
        n_wr = sum(1 for i in code if i.is_lds and i.mn.startswith("ds_write"))
        n_dma = sum(1 for i in code if i.is_ldsdma)
        n_bar = sum(1 for i in code if i.mn == "s_barrier")
        n_cnt = sum(1 for i in code if i.vmcnt is not None and i.vmcnt > 0)
        if n_rd or n_wr or n_dma:
            rows.append((name, n_rd, n_wr, n_dma, n_bar, n_cnt, max([w for _, _, w in rep], default=0), len(bad)))
    return rows


# what the rules above know how to reason about when it appears INSIDE an asm string
_ASM_MODELLED = re.compile(r"^(s_mov_b32|s_add_u32|s_nop|s_waitcnt|landed|global_load_lds_dwordx4|global_load_dwordx4|"
                           r"global_store_dword|global_store_dwordx2|global_store_dwordx3|global_store_dwordx4|"
                           r"v_fma_mixlo_f16|v_fma_mixhi_f16|v_fma_mix_f32|v_mov_b32)$")


def check_modelled(name, code):
    return [(name, "unmodelled-asm", 0, 0, i, i, []) for i in code if i.in_asm and not _ASM_MODELLED.match(i.mn)]


def check_scratch(name, code, succs):
    """Register spills (scratch_load / scratch_store) are VMEM operations: in a wave that hand-counts `s_waitcnt vmcnt(N)`
    they would sit in the same in-order queue and silently shift the count.  A kernel may spill in one role (a branch of
    waves that never executes a counted wait -- hipcc counts its own scratch traffic correctly there) but not on any path
    that also meets an inline-asm counted wait: no asm `vmcnt(N > 0)` may be reachable from a scratch instruction, nor a
    scratch instruction from such a wait."""
    scratch = [i for i in code if i.mn.startswith("scratch_")]
    waits = [i for i in code if i.in_asm and i.vmcnt is not None and i.vmcnt > 0]
    if not scratch or not waits:
        return []

    def reach(starts):
        seen, stack = set(), [x.idx for x in starts]
        while stack:
            i = stack.pop()
            for nxt in succs[i]:
                if nxt not in seen:
                    seen.add(nxt)
                    stack.append(nxt)
        return seen

    from_scratch, from_waits = reach(scratch), reach(waits)
    out = []
    for w in waits:
        if w.idx in from_scratch:
            out.append((name, "scratch-before-counted-wait", 0, 0, scratch[0], w, []))
            break
    for x in scratch:
        if x.idx in from_waits:
            out.append((name, "scratch-after-counted-wait", 0, 0, waits[0], x, []))
            break
    return out


_M0_OK = (re.compile(r"^s_mov_b32 m0, s\d+$"), re.compile(r"^s_add_u32 m0, s\d+, (0x[0-9a-f]+|\d+|s\d+)$"),
          re.compile(r"^s_mov_b32 s\d+, m0$"))


def check_m0(name, code):
    if not any(i.in_asm and "m0" in i.defs for i in code):
        return []
    out = []
    for i in code:
        if not i.in_asm and ("m0" in i.defs or "m0" in i.uses or "m0" in i.ops):
            out.append((name, "m0-compiler-use", 0, 0, i, i, ["m0"]))
    return out


def lint_file(path, all_pairs=False):
    """List of violations (function, rule, states required, states found, producer Ins, consumer Ins, registers)."""
    out = []
    for name, code, labels in parse_functions(path):
        preds = build_preds(code, labels)
        succs = build_succs(code, labels)
        out += check_waitstates(name, code, preds, all_pairs)
        out += check_asm_loads(name, code, succs)
        out += check_m0(name, code)
        out += check_modelled(name, code)
        out += check_scratch(name, code, succs)
        out += check_lds_dma(name, code, succs)
    return out


def format_violation(v):
    name, rule, need, got, P, C, regs = v
    where = "asm" if P.in_asm else "compiler"
    wherec = "asm" if C.in_asm else "compiler"
    return (f"{rule}: needs {need} wait states, has {got}  [{','.join(regs[:6])}]\n"
            f"    in {name}\n    producer ({where}) line {P.line}: {P.text}\n    consumer ({wherec}) line {C.line}: {C.text}")


def summarize(path):
    funcs = parse_functions(path)
    n_ins = sum(len(c) for _, c, _ in funcs)
    n_asm = sum(1 for _, c, _ in funcs for i in c if i.in_asm)
    return len(funcs), n_ins, n_asm


def main(argv):
    files = [a for a in argv if not a.startswith("--")]
    all_pairs = "--all" in argv
    quiet = "--quiet" in argv
    bad = 0
    if "--lds" in argv:
        for f in files:
            print(f"{f}: LDS protocol summary (kernels that touch LDS)")
            print("  %-90s %8s %8s %8s %8s %8s %14s %6s" % ("kernel", "ds_read", "ds_write", "LDS-DMA", "barriers", "counted", "max bars ahead", "viol."))
            for name, n_rd, n_wr, n_dma, n_bar, n_cnt, ahead, nb in lds_protocol_summary(f):
                print("  %-90s %8d %8d %8d %8d %8d %14d %6d" % (name[:90], n_rd, n_wr, n_dma, n_bar, n_cnt, ahead, nb))
        return 0
    for f in files:
        v = lint_file(f, all_pairs)
        nf, ni, na = summarize(f)
        seen = set()
        uniq = []
        for x in v:
            k = (x[1], x[4].line, x[5].line)
            if k not in seen:
                seen.add(k)
                uniq.append(x)
        print(f"{f}: {nf} functions, {ni} instructions ({na} from inline asm): {len(uniq)} violation(s)")
        if not quiet:
            for x in uniq[:200]:
                print("  " + format_violation(x))
        bad += len(uniq)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
