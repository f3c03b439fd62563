#!/usr/bin/env python3
"""Generator of the K-tile bodies of lab/gemm4w_lab.hip (the text between the GENERATED markers).

gemm4w: 256 x 256 x 64 block tile, FOUR waves (one per SIMD), 128 x 128 wave tiles of v_mfma_f32_32x32x16_bf16 with all 256
accumulators in the accumulator file (allocated by hand), operands through two LDS buffers filled from registers
(buffer_load_dwordx4 -> two staging sets -> ds_write_b128), ONE barrier per K-tile.  Every MFMA of a K-tile is followed by at
most one other statement (64 MFMAs : 32 fragment reads + 16 LDS writes + 16 global loads).  In tile kt (LDS buffer kt & 1):

  * slots 0..7 of k-step ks: the fragments of k-step ks + 1 (k-step 3: of k-step 0 of tile kt + 1, other buffer, after the barrier);
  * the 16 LDS writes of tile kt + 1 (staging set (kt + 1) & 1) in the remaining slots of k-steps 0..2 (6 + 5 + 5: four waves that
    write in the same eight consecutive slots exceed the LDS store path), then the barrier behind k-step 2's last statement;
  * the 16 global loads of tile kt + 3 into the staging set the writes have just drained, each behind the write of its own
    registers: a load is 1.5 K-tiles old when its write needs it (0.5 in the first version: the barrier then cost 417 cycles per
    K-tile, the spread of the four waves' memory latencies).

s_waitcnt values are COUNTED: the script replays the issue order of a whole run (prologue + tiles) through the in-order
vector-memory and LDS queues and takes, for every statement that needs an earlier one, the number of later ones in flight.

    python lab/gen_gemm4w_body.py            rewrite lab/gemm4w_lab.hip in place"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
NB = 4                                      # N-blocks of 32 columns per wave: 4 = 256-wide tile, 3 = 192-wide (set by main())


def frag_order():
    return ["W0", "A0", "A1", "A2", "A3"] + [f"W{j}" for j in range(1, NB)]


def writes_per_kstep():
    return [6, 5, 5, 0] if NB == 4 else [5, 5, 4, 0]


def tile_events(kt, nk):
    """statements of tile kt in issue order: list of slots, each a list of (kind, args); MFMA (jn, im) heads each slot.
    kinds: 'need_frags' (ks), 'mfma', 'rd' (op, buf, ks, blk, set), 'wr' (tile, i), 'ld' (tile, i), 'barrier'"""
    b = kt & 1
    ev = []
    nw = 0                                   # writes of tile kt + 1 issued so far
    nl = 0                                   # loads of tile kt + 3 issued so far
    do_w = kt + 1 < nk
    do_l = kt + 3 < nk
    FRAG_ORDER, WRITES_PER_KSTEP, NMEM = frag_order(), writes_per_kstep(), 8 + 2 * NB
    for ks in range(4):
        ev.append(("need_frags", kt, ks))
        slot = 0
        for jn in range(NB):
            for im in range(4):
                ev.append(("mfma", jn, im, ks & 1))
                if slot < len(FRAG_ORDER):
                    op, blk = FRAG_ORDER[slot][0], int(FRAG_ORDER[slot][1])
                    if ks < 3:
                        ev.append(("rd", op, b, ks + 1, blk, (ks + 1) & 1, kt, ks + 1))
                    elif kt + 1 < nk:
                        ev.append(("rd", op, b ^ 1, 0, blk, 0, kt + 1, 0))
                else:
                    wrote = False
                    if do_w and nw < sum(WRITES_PER_KSTEP[:ks + 1]):
                        ev.append(("wr", kt + 1, nw))
                        nw += 1
                        wrote = True
                    # 192-wide tile: 48 MFMAs for 28 reads + 14 writes + 14 loads -- a write slot also carries the load
                    # into the registers it has just drained
                    if do_l and nl < nw and nl < NMEM and (NB == 3 or not wrote):
                        ev.append(("ld", kt + 3, nl))
                        nl += 1
                    if ks == 2 and slot == 4 * NB - 1 and kt + 1 < nk:
                        ev.append(("barrier", kt))
                slot += 1
    assert (not do_w or nw == NMEM) and (not do_l or nl == NMEM), (kt, nw, nl)
    return ev


def run_events(nk, stores_after=None):
    """stores_after = stream K-tile index after which an epilogue issues its 32 global stores (they sit in the same in-order
    vector-memory queue as the loads)"""
    ev = []
    NMEM = 8 + 2 * NB
    for t in (0, 1):                          # prologue: tiles 0 and 1 requested, tile 0 written, tile 2 requested into the set
        for i in range(NMEM):                 # tile 0 left, tile 0's first fragments read
            ev.append(("ld", t, i))
    for i in range(NMEM):
        ev.append(("wr", 0, i))
    for i in range(NMEM):
        ev.append(("ld", 2, i))
    ev.append(("barrier", -1))
    for f in frag_order():
        ev.append(("rd", f[0], 0, 0, int(f[1]), 0, 0, 0))
    marks = {}
    for kt in range(nk):
        marks[kt] = len(ev)
        ev += tile_events(kt, nk)
        if stores_after is not None and kt == stores_after:
            ev += [("st", kt, i) for i in range(8 * NB)]
    marks[nk] = len(ev)
    return ev, marks


def annotate(ev):
    """-> per event index: ('vm', n) / ('lgkm', n) waits that must precede it"""
    waits = {}
    vm = []                                   # outstanding loads in issue order: (tile, i)
    lds = []                                  # LDS operations in issue order: ('rd', tile, ks) / ('wr', ...)
    for n, e in enumerate(ev):
        if e[0] == "ld":
            vm.append((e[1], e[2]))
        elif e[0] == "st":
            vm.append(("store", e[1], e[2]))
        elif e[0] == "wr":
            pos = max(k for k, x in enumerate(vm) if x == (e[1], e[2]))
            waits[n] = ("vm", min(len(vm) - 1 - pos, 63))      # (6-bit counter: a smaller value only waits for more)
            lds.append(("wr", e[1], e[2]))
        elif e[0] == "rd":
            lds.append(("rd", e[6], e[7]))
        elif e[0] == "barrier":
            waits[n] = ("lgkm", 0)
        elif e[0] == "need_frags":
            pos = max(k for k, x in enumerate(lds) if x == ("rd", e[1], e[2]))
            waits[n] = ("lgkm", len(lds) - 1 - pos)
    return waits


def emit_range(ev, waits, lo, hi):
    out = []
    for n in range(lo, hi):
        e = ev[n]
        w = waits.get(n)
        pre = ""
        if w:
            pre = f"VMC({w[1]}) " if w[0] == "vm" else f"LGKM({w[1]}) "
        if e[0] == "need_frags":
            out.append(f"        {pre.strip()}")
        elif e[0] == "mfma":
            out.append(f"        MFMA({e[1]}, {e[2]}, {e[3]})")
        elif e[0] == "rd":
            out.append(f"        RD{e[1]}({e[2]}, {e[3]}, {e[4]}, {e[5]})")
        elif e[0] == "wr":
            t, i = e[1], e[2]
            out.append(f"        {pre}{'WRA' if i < 8 else 'WRW'}({t & 1}, {i & 7})")
        elif e[0] == "ld":
            t, i = e[1], e[2]
            step = "KSTEP() " if (i == 0 and t > 0) else ""        # the scalar K offset moves on before a tile's first load
            out.append(f"        {step}{'LDA' if i < 8 else 'LDW'}({t & 1}, {i & 7})")
        elif e[0] == "barrier":
            out.append(f"        {pre}BARRIER()")
        elif e[0] == "st":
            pass
    return "\n".join(out) + "\n"


def check(path):
    """every instantiation of gemm4w_kernel in a .s file: no accumulator-file operand, v_accvgpr_* or scratch access outside
    the hand-written statements (the compiler must not park values in registers the statements own)"""
    src = open(path).read()
    bad = 0
    pos = 0
    nker = 0
    while True:
        i = src.find("gemm4w_kernelILi", pos)
        if i < 0:
            break
        j = src.index("\n", i)
        ls = src.rfind("\n", 0, i) + 1
        line = src[ls:j]
        if not (line.startswith("_Z") and ":" in line.split(";")[0]):      # the label line "name:   ; @name"
            pos = j
            continue
        e = src.index("s_endpgm", j)
        nker += 1
        inasm = False
        for ln in src[j:e].split("\n"):
            if "ASMSTART" in ln:
                inasm = True
            elif "ASMEND" in ln:
                inasm = False
            elif not inasm and ("accvgpr" in ln or "scratch_" in ln or " a[" in ln):
                bad += 1
                print(src[i:j], ln)
        pos = e
    print(f"{path}: {bad} accumulator-file / scratch statements outside the hand-written ones ({nker} kernels checked)")
    return 1 if bad else 0


def main():
    import sys
    global NB
    if len(sys.argv) > 2 and sys.argv[1] == "--check":
        sys.exit(check(sys.argv[2]))
    for NB in (4, 3):
        generate("" if NB == 4 else "_N3")


def generate(suffix):
    nk = 12                                   # model run: steady tiles 0 .. nk - 4, then three tail tiles
    ev, marks = run_events(nk)
    waits = annotate(ev)
    text = {kt: emit_range(ev, waits, marks[kt], marks[kt + 1]) for kt in range(nk)}
    # steady state: tiles 2 .. nk - 4 of the same parity are identical
    for kt in range(2, nk - 5):
        assert text[kt] == text[kt + 2], kt
    assert text[0] == text[2] and text[1] == text[3], "the first tiles already are the steady state"
    path = os.path.join(HERE, "gemm4w_lab.hip")
    src = open(path).read()
    # the two K-tiles behind an epilogue: their LDS writes need loads that are OLDER than the epilogue's 32 stores -- counted
    # with the stores in the queue (vector-memory operations of a wave complete in order on gfx9 / CDNA), so that they do not
    # wait for the store burst to drain
    ev2, marks2 = run_events(20, stores_after=7)
    waits2 = annotate(ev2)
    text2 = {kt: emit_range(ev2, waits2, marks2[kt], marks2[kt + 1]) for kt in range(20)}
    assert text2[2] == text[2] and text2[7] == text[3] and text2[10] == text[2] and text2[11] == text[3], "steady again from the third K-tile on"
    assert text2[8] != text[2]
    blocks = {"PROLOGUE": emit_range(ev, waits, 0, marks[0]), "STEADY0": text[2], "STEADY1": text[3], "STEADY0B": text[2],
              "POST0": text2[8], "POST1": text2[9],
              "TAIL3": text[nk - 3], "TAIL2": text[nk - 2], "TAIL1": text[nk - 1]}
    assert (nk - 3) & 1 == 1 and (nk - 4) & 1 == 0
    for tag, body in blocks.items():
        tag += suffix
        a = src.index(f"// GENERATED {tag} BEGIN")
        b = src.index(f"// GENERATED {tag} END")
        src = src[:a] + f"// GENERATED {tag} BEGIN (lab/gen_gemm4w_body.py)\n" + body + "        " + src[b:]
    open(path, "w").write(src)


if __name__ == "__main__":
    main()
