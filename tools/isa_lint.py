#!/usr/bin/env python3
"""Compile-time checks of the gfx950 code of the product kernels.  Used by the library build (ffcnn_amd/csrc/Makefile: a failing
lint fails the build) and by tests/test_isa_lint.py.      python tools/isa_lint.py <listing.s> [--census]

(1) No packed fp32 instruction may take ONE register pair in two source slots under op_sel / op_sel_hi modifiers, e.g.
        v_pk_fma_f32 v[6:7], v[2:3], v[0:1], v[0:1] op_sel:[0,0,1] op_sel_hi:[1,0,1]        (hipcc's form of acc * sc + bi with sc, bi in one pair).
    While a bf16 MFMA is in flight on the SIMD -- the wave's own or another kernel's -- that instruction sporadically dropped its addend on lanes 48-63
    (found in k_pw_x3, then in k_pw_mfma under the split-bf16 kernels of neighbouring chains: tools/x3s_exec_race.py).
(2) Every OTHER packed fp32 form with op_sel / op_sel_hi / neg modifiers that hipcc emits today is listed in KNOWN_FORMS: each ran by the million next to bf16
    MFMAs in the round-4 / round-5 soaks (tests/test_gpu_round4.py, test_gpu_round5.py) without a differing bit.  A compiler upgrade that emits a NEW form, or
    many more of a known one (CAP x today's count), fails here instead of in the field.
(3) Kernels that stage operands by LDS-DMA (global_load_lds_dwordx4 / buffer_load_dwordx4 ... lds): every s_barrier must be preceded by an
    s_waitcnt vmcnt(0) with no vector-memory instruction in between -- hipcc does not model that these loads write LDS.
(4) No buffer_store_dwordx3 / x4 with its soffset in an SGPR may be followed DIRECTLY by a vector instruction that writes one of its data registers: on gfx950
    that dword of the store is corrupted on some lanes (found in round 5 in a k_pw_x3t epilogue: `buffer_store_dwordx4 v[2:5], v158, s[28:31], s6 offen`
    then `v_fma_f32 v3, ...` -- element 1 of sporadic 16-byte stores wrong, profiles/r05_e_buffer_store_hazard.txt).  hipcc's hazard recognizer inserts the
    wait state only for an immediate soffset (and for global / flat stores, which is why the product uses those)."""
import re
import sys

CAP = 1.6       # a known form may grow to CAP x its recorded count (+ 16) before the lint asks for a fresh look
# signature -> count at the time the form was last soaked (tools/isa_lint.py <listing> --census prints today's table)
KNOWN_FORMS = {
    # (operand kinds: v = VGPR pair, s = SGPR pair, c = constant / literal; counts of the round-5 tree incl. the three-column k_front, whose u8 conversion
    #  brought the two `vs ... op_sel:[0,1]` forms: mean / norm out of the high half of an SGPR pair -- soaked by the u8 mode of tests/test_gpu_round5.py)
    'v_pk_add_f32 vc op_sel_hi:[1,0]': 3,
    'v_pk_add_f32 vs neg_hi:[0,1] neg_lo:[0,1]': 69,
    'v_pk_add_f32 vs neg_hi:[0,1] neg_lo:[0,1] op_sel_hi:[1,0]': 51,
    'v_pk_add_f32 vv neg_hi:[0,1] neg_lo:[0,1]': 43,      # round 6: the wave-level Inf test (x3_wave_has_inf) lets hipcc pack the subtractions of the fused blocks' split; soaked (tests/test_gpu_round5.py)
    'v_pk_fma_f32 vsc op_sel_hi:[1,1,0]': 40,
    'v_pk_fma_f32 vvc op_sel:[0,1,0] op_sel_hi:[1,1,0]': 8,
    'v_pk_fma_f32 vvc op_sel:[1,0,0] op_sel_hi:[1,1,0]': 72,
    'v_pk_fma_f32 vvc op_sel_hi:[0,1,0]': 48,
    'v_pk_fma_f32 vvc op_sel_hi:[1,0,0]': 302,
    'v_pk_fma_f32 vvc op_sel_hi:[1,1,0]': 790,
    'v_pk_fma_f32 vvv op_sel:[0,1,0]': 212,
    'v_pk_fma_f32 vvv op_sel:[1,0,0]': 1992,
    'v_pk_fma_f32 vvv op_sel_hi:[0,1,1]': 4781,
    'v_pk_fma_f32 vvv op_sel_hi:[1,0,1]': 2608,
    'v_pk_mul_f32 sv op_sel:[1,0]': 51,
    'v_pk_mul_f32 sv op_sel_hi:[0,1]': 349,
    'v_pk_mul_f32 vc op_sel_hi:[1,0]': 1,
    'v_pk_mul_f32 vs op_sel_hi:[1,0]': 143,
    'v_pk_mul_f32 vv op_sel:[1,0]': 216,
    'v_pk_mul_f32 vv op_sel_hi:[0,1]': 144,
}


def kernels(text):
    """name -> body of every kernel / device function of the listing.  A body runs to the function's `.Lfunc_endN:` label (or the next symbol / the end of
    the text), NOT to its first s_endpgm: a kernel with early returns has several, and everything behind the first one went unchecked."""
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n", text, re.M):
        rest = text[m.end():]
        e = re.search(r"^(?:\.Lfunc_end\d+:|_Z\w+:)", rest, re.M)
        out[m.group(1)] = rest[:e.start()] if e else rest
    return out


def pk_signature(line):
    """(opcode, operand kinds, same-pair flag, modifiers) of a packed fp32 instruction, or None"""
    m = re.match(r"\s+(v_pk_\w+_f32)\s+(\S+),\s*(\S+),\s*(\S+)(?:,\s*(\S+?))?((?:\s+\w+:\[[^\]]*\])*)\s*$", line)
    if not m:
        return None
    ops = [x.rstrip(",") for x in (m.group(3), m.group(4), m.group(5)) if x]
    kinds = "".join("v" if o.startswith("v[") else "s" if o.startswith("s[") else "c" for o in ops)
    regs = [o for o in ops if o[0] in "vs" and "[" in o]
    dup = len(regs) != len(set(regs))
    mods = " ".join(sorted(re.findall(r"\w+:\[[^\]]*\]", m.group(6) or "")))
    return m.group(1), kinds, dup, mods


def census(text):
    forms, bad, npk = {}, [], 0
    for line in text.split("\n"):
        sig = pk_signature(line)
        if not sig:
            continue
        npk += 1
        op, kinds, dup, mods = sig
        if not mods:
            continue
        if dup and "op_sel" in mods:
            bad.append(line.strip())
        key = "%s %s%s %s" % (op, kinds, " SAMEPAIR" if dup else "", mods)
        forms[key] = forms.get(key, 0) + 1
    return forms, bad, npk


def check_lds_dma(text):
    """every s_barrier of a kernel with LDS-DMA loads: the closest s_waitcnt before it (no vector-memory instruction in between) is vmcnt(0), or
    vmcnt(N) with the N youngest vector-memory instructions in front of it all plain loads (loads return in order: the LDS-DMA pieces, older, are then complete)"""
    errs, seen = [], 0
    vmem = ("global_load", "buffer_load", "global_store", "buffer_store", "global_atomic", "buffer_atomic", "flat_")
    for name, body in kernels(text).items():
        if "global_load_lds" not in body and not re.search(r"buffer_load_dword\w*\s.*\blds\b", body):
            continue
        seen += 1
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        for i, l in enumerate(lines):
            if not l.startswith("s_barrier"):
                continue
            j, good, why = i - 1, False, "no s_waitcnt vmcnt in front of it"
            while j >= 0:
                m = re.search(r"vmcnt\((\d+)\)", lines[j])
                if m:
                    n = int(m.group(1))
                    young = [x for x in reversed(lines[:j]) if x.startswith(vmem)][:n]
                    if n == 0 or (len(young) == n and all(x.startswith(("global_load", "buffer_load")) and "lds" not in x for x in young)):
                        good = True
                    else:
                        why = "vmcnt(%d) may leave an LDS-DMA piece or a store in flight" % n
                    break
                if lines[j].startswith(vmem):
                    why = "a vector-memory instruction between the wait and the barrier"
                    break
                j -= 1
            if not good:
                errs.append("%s: s_barrier: %s: %s" % (name, why, " | ".join(lines[max(0, i - 6):i + 1])))
    return errs, seen


def reg_range(op):
    """'v[2:5]' -> (2, 5), 'v3' -> (3, 3), else None"""
    m = re.match(r"v\[(\d+):(\d+)\]$", op)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.match(r"v(\d+)$", op)
    return (int(m.group(1)),) * 2 if m else None


def check_buffer_store(text):
    """rule (4): buffer_store_dwordx3 / x4 with an SGPR soffset directly followed by a vector instruction that writes one of its data registers"""
    errs, nst = [], 0
    lines = [l.strip() for l in text.split("\n") if l.strip() and not l.strip().startswith((";", ".", "//")) and not l.strip().endswith(":")]
    for i, l in enumerate(lines[:-1]):
        m = re.match(r"buffer_store_dwordx[34]\s+(v\[\d+:\d+\]),\s*(\S+),\s*(s\[\d+:\d+\]),\s*(\S+)", l)
        if not m:
            continue
        nst += 1
        if not re.match(r"s\d+$", m.group(4).rstrip(",")):            # `0`, `off`, a literal: hipcc inserts the wait state itself
            continue
        nxt = lines[i + 1]
        if not nxt.startswith("v_") or nxt.startswith(("v_cmp", "v_cmpx", "v_readfirstlane", "v_readlane")):
            continue
        dst = reg_range(nxt.split(None, 1)[1].split(",")[0].strip()) if len(nxt.split(None, 1)) > 1 else None
        lo, hi = reg_range(m.group(1))
        if dst and dst[0] <= hi and dst[1] >= lo:
            errs.append("buffer store of more than 8 bytes with an SGPR soffset, data register overwritten by the next instruction: %s | %s" % (l, nxt))
    return errs, nst


def lint(text, strict=True):
    """list of error strings (empty = clean), and the facts the tests assert on.  Rules 1, 3 and 4 are always errors.  Rule 2 (the census of packed-fp32
    modifier forms: counts of ONE hipcc release) is an error for `strict` callers -- the tests and `make LINT_STRICT=1`, i.e. this repo's own builds -- and a
    WARNING (facts["warnings"]) in a plain `make`: another ROCm release that emits a new or more frequent form must not break a user's build."""
    forms, bad, npk = census(text)
    errs = ["packed fp32 op with one register pair in two slots under op_sel: " + b for b in bad[:5]]
    cens = []
    for key, n in sorted(forms.items()):
        if "SAMEPAIR" in key and "op_sel" in key:
            continue
        if key not in KNOWN_FORMS:
            cens.append("NEW packed-fp32 modifier form (soak it next to bf16 MFMAs, then add it to tools/isa_lint.py KNOWN_FORMS): %s  x %d" % (key, n))
        elif n > KNOWN_FORMS[key] * CAP + 16:
            cens.append("packed-fp32 form grew from %d to %d instructions (re-soak, then update KNOWN_FORMS): %s" % (KNOWN_FORMS[key], n, key))
    e2, seen = check_lds_dma(text)
    e3, nst = check_buffer_store(text)
    facts = {"npk": npk, "forms": forms, "lds_dma_kernels": seen, "wide_buffer_stores": nst, "buffer_store_hazards": len(e3), "warnings": [] if strict else cens}
    return errs + (cens if strict else []) + e2 + e3[:5], facts


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    errs, facts = lint(text, strict="--strict" in sys.argv)
    if "--census" in sys.argv:
        print("KNOWN_FORMS = {")
        for k, n in sorted(facts["forms"].items()):
            print("    %r: %d," % (k, n))
        print("}")
    print("isa_lint: %d packed fp32 instructions, %d modifier forms, %d LDS-DMA kernels, %d error(s)" % (facts["npk"], len(facts["forms"]), facts["lds_dma_kernels"], len(errs)))
    for e in facts["warnings"]:
        print("isa_lint warning (an error under --strict / make LINT_STRICT=1):", e)
    for e in errs:
        print("isa_lint ERROR:", e)
    sys.exit(1 if errs else 0)
