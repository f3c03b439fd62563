#!/usr/bin/env python3
"""Generator of the steady-state loop body of lab/attn128_pipe.h: attn128_pipe_kernel (the text between the GENERATED markers).

The body is a fixed interleave of hand-written statements: every 32x32x16 MFMA is followed by one "quarter" of exponentials
(v_cvt_pk of the PREVIOUS quarter, then two v_exp_f32), so that one wave's vector ALU and the matrix pipe of its SIMD are
both busy (lab/mfma_valu_overlap.hip: about 2.5 exponentials fit beside one MFMA).  Step n of the software pipeline over
32-row blocks n = 4 jt + x holds QK(n + 1), PV(n - 1) and exp(n); the LDS reads of the next fragments follow the MFMAs that
free their registers.

    python lab/gen_attn128_body.py            rewrite the header in place
    python lab/gen_attn128_body.py --check F  check a disassembly / .s file F: inside attn128_pipe_kernel no accumulator-file
                                              operand, v_accvgpr_* or scratch access outside the asm statements"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def gen(fused=True):
    out = []
    emit = out.append
    state = {"prev": None, "t": 0}                 # previous quarter's destination word, temp pair in use

    def quarter_args(S, P, g, q):
        """(cvt destination, cvt inputs, exp destinations and sources) of the next quarter; advances the temp pair"""
        t = state["t"]
        pP, pg, pq = state["prev"] if state["prev"] is not None else ("pb", 3, 3)      # tile start: the previous tile's last quarter
        args = (f"{pP}[{pg}][{pq}]", f"e{1 - t}a", f"e{1 - t}b", f"e{t}a", f"{S}[{g >> 1}][{8 * (g & 1) + 2 * q}]", f"e{t}b", f"{S}[{g >> 1}][{8 * (g & 1) + 2 * q + 1}]")
        state["prev"] = (P, g, q)
        state["t"] = 1 - t
        return args

    def step(mf, S, P, maskx, extra=None):
        """mf: the MFMA statements of the step in order (16 big ones, each with a quarter; LSM = row-sum MFMA, no quarter);
        extra[n] = lines emitted after statement n (and its quarter)"""
        extra = extra or {}
        quarters = [(g, q) for g in range(4) for q in range(4)]
        qi = 0
        for n, m in enumerate(mf):
            big = not m.startswith("LSM")
            if big:
                g, q = quarters[qi]
                qi += 1
                w, ca, cb, ea, xa, eb, xb = quarter_args(S, P, g, q)
            if big and fused and n > 0:
                # QKM(x, ks, i, s) -> QKF(x, ks, i, s, quarter...), PVM -> PVF; a leading "KWAIT"-like prefix stays in front
                head, call = m.rsplit(" ", 1) if " " in m and not m.startswith(("QKM(", "PVM(")) else ("", m)
                ks0 = call.startswith("QKM(") and call.split(",")[1].strip() == "0"
                call = call.replace("QKM(", "QKF0(" if ks0 else "QKF(").replace("PVM(", "PVF(")
                assert call.endswith(")")
                emit("        " + (head + " " if head else "") + call[:-1] + f", {w}, {ca}, {cb}, {ea}, {xa}, {eb}, {xb})")
            else:
                emit("        " + m)
                if n == 0:
                    emit(f"        MASKB({maskx}, {S})")            # (the step's first quarter reads S: the mask goes in front of it)
                if big:
                    emit(f"        CVT({w}, {ca}, {cb}) EXP({ea}, {xa}) EXP({eb}, {xb})")
            for ln in extra.get(n, []):
                emit("        " + ln)
        assert qi == 16, qi

    def qk(x, ks, i, s):
        return f"QKM({x}, {ks}, {i}, {s})"

    def pv(x, i, g, p):
        return f"PVM({x}, {i}, {g}, {p})"

    def ls(x, g, p):
        return f"LSM({x}, {g}, {p})"

    def alternate(xq, sq, xp, pp):
        m = []
        for ks in range(4):
            m += [qk(xq, ks, 0, sq), pv(xp, 0, ks, pp), qk(xq, ks, 1, sq), pv(xp, 1, ks, pp), ls(xp, ks, pp)]
        return m

    emit("        // ---- step A: QK(jt, 1) -> sb, PV(jt - 1, 3) from pb, exp(sa) -> pa; the V fragments of tile jt slot by slot behind")
    emit("        //      the PV MFMAs that read the old ones")
    state["prev"] = None
    # statement index of slot g's last reader PVM(3, 1, g) is 5 g + 3 (LSM at 5 g + 4): read slot g after the next slot's first MFMA
    step(alternate(1, "sb", 3, "pb"), "sa", "pa", 0, extra={5: ["RDV(0)"], 10: ["RDV(1)"], 15: ["RDV(2)"]})
    emit("        stamp(0);")
    emit("        // ---- step B: QK(jt, 2) -> sa, exp(sb) -> pb; the tile boundary; PV(jt, 0) from pa")
    mB = [qk(2, 0, 0, "sa"), qk(2, 0, 1, "sa"), qk(2, 1, 0, "sa"), qk(2, 1, 1, "sa"),
          pv(0, 0, 0, "pa"), pv(0, 1, 0, "pa"), ls(0, 0, "pa"),
          qk(2, 2, 0, "sa"), pv(0, 0, 1, "pa"), qk(2, 2, 1, "sa"), pv(0, 1, 1, "pa"), ls(0, 1, "pa"),
          qk(2, 3, 0, "sa"), pv(0, 0, 2, "pa"), qk(2, 3, 1, "sa"), pv(0, 1, 2, "pa"), ls(0, 2, "pa"),
          pv(0, 0, 3, "pa"), pv(0, 1, 3, "pa"), ls(0, 3, "pa")]
    step(mB, "sb", "pb", 1, extra={0: ["RDV(3)"], 3: ["BOUNDARY()", "LGKM0()"]})
    emit("        stamp(1);")
    emit("        // ---- step C: QK(jt, 3) -> sb, PV(jt, 1) from pb, exp(sa) -> pa; the K fragments of tile jt + 1 k-slice by k-slice")
    emit("        //      behind the QK MFMAs that read the old ones")
    # slot ks's last reader QKM(3, ks, 1) is statement 5 ks + 2: read slot ks after the following PVM (5 ks + 3)
    step(alternate(3, "sb", 1, "pb"), "sa", "pa", 2, extra={3: ["RDK(0)"], 8: ["RDK(1)"], 13: ["RDK(2)"], 19: ["RDK(3)"]})
    emit("        stamp(2);")
    emit("        // ---- step D: PV(jt, 2) from pa, QK(jt + 1, 0) -> sa (on the last tile: of stale K fragments, never read), exp(sb) -> pb")
    mD = [pv(2, 0, 0, "pa"), pv(2, 1, 0, "pa"), ls(2, 0, "pa"), qk(0, 0, 0, "sa"), qk(0, 0, 1, "sa")]
    assert not any(" " in m.split("(")[0] for m in mD)
    for ks in range(1, 4):
        mD += [qk(0, ks, 0, "sa"), pv(2, 0, ks, "pa"), qk(0, ks, 1, "sa"), pv(2, 1, ks, "pa"), ls(2, ks, "pa")]
    # k-slices 0 .. 2 were requested long ago; k-slice 3 (the last two reads) a moment ago
    step(mD, "sb", "pb", 3, extra={2: ['asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");'], 14: ["LGKM0()"]})
    emit("        stamp(3);")
    assert mD[15].startswith("QKM(0, 3, 0")
    # the last quarter of block 3 is converted by the first filler of the next tile's step A (or by the flush after the loop)
    assert state["prev"] == ("pb", 3, 3) and state["t"] == 0
    return "\n".join(out) + "\n"


def check(path):
    src = open(path).read()
    i = src.index("attn128_pipe_kernelILi0E")
    i = src.index(":", i)
    body = src[i:src.index("s_endpgm", i)].split("\n")
    inasm, bad = False, []
    for ln in body:
        if "ASMSTART" in ln:
            inasm = True
        elif "ASMEND" in ln:
            inasm = False
        elif not inasm and ("accvgpr" in ln or "scratch_" in ln or " a[" in ln):
            bad.append(ln)
    print(f"{path}: {len(bad)} accumulator-file / scratch statements outside the hand-written ones")
    for ln in bad[:10]:
        print("   ", ln)
    return 1 if bad else 0


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--check":
        sys.exit(check(sys.argv[2]))
    path = os.path.join(HERE, "attn128_pipe.h")
    src = open(path).read()
    a = src.index("// GENERATED BODY BEGIN")
    b = src.index("// GENERATED BODY END")
    src = src[:a] + "// GENERATED BODY BEGIN (lab/gen_attn128_body.py)\n" + gen() + "        " + src[b:]
    a = src.index("// GENERATED PAIR BODY BEGIN")
    b = src.index("// GENERATED PAIR BODY END")
    src = src[:a] + "// GENERATED PAIR BODY BEGIN (lab/gen_attn128_body.py, separate statements)\n" + gen(fused=False) + "        " + src[b:]
    open(path, "w").write(src)


if __name__ == "__main__":
    main()
