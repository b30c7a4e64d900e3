#!/usr/bin/env python3
"""Generator of nerf_sampling_amd/csrc/ns_ob16_asm.inc: hand-scheduled gfx950 instruction streams for the hidden layers
of the 16x16x32 MLP engine (layer_ob16<> in ns_mlp_engine.h is the compiler-scheduled statement of the same layer).

Why: a lone wave per SIMD issues in order, and a v_mfma_f32_16x16x32 holds the vector issue port for 8 of its 16 cycles
(MI355X_MICROARCH.md, cycle constants), so a chunk step (4 MFMAs = 64 matrix cycles) has room for ~8 single-issue
instructions -- IF they sit evenly between the MFMAs.  The compiler's schedule bunches the conversions, recycles
accumulator registers as fragment-read targets (s_nop pads) and waits on lgkmcnt in front of most MFMA groups; the
kernel spends 91 cycles per chunk step instead of 64.  This generator emits the same arithmetic (same MFMA order per
accumulator: results are bit-identical to layer_ob16<>) as one asm statement per layer with a fixed register map:

  v[0:31]    accumulators  acc[parity][tile] = v[16*parity + 4*tile : +3]     (clobbered)
  v[32:47]   A fragments   f[i] = v[32 + 4 i : +3], chunk p lives in f[p % 4]  (in/out: the ring's read-ahead)
  v[48:51]   bias of the sub-block about to start (C operand of its first MFMAs), v[52:55] scratch   (clobbered)
  v[64:95]   the skip layer's embedded point, xs[tile][kb] = v[64 + 8*tile + 4*kb : +3]          (input, skip only)
  a[0:127]   activation set A, hA[tile][kb] = a[32*tile + 4*kb : +3]
  v[128:255] activation set V, hB[tile][kb] = v[128 + 32*tile + 4*kb : +3]

Layers alternate A -> V (conversion results are written by VALU straight into set V) and V -> A (one extra
v_accvgpr_write per dword).  Hazards follow the rules the compiler applies to this MFMA on gfx950 (measured from its own
output, tools/gen_ob16_asm.py --help): D write -> any other access 8 wait states, VALU write -> MFMA read 2, C read ->
overwrite 3; the emitter inserts s_nop where a rule needs it and an independent checker pass re-verifies the final text.
"""
import argparse
import os
import sys

T = 4                 # 16-sample tiles per wave
SLAB = 16             # chunks per slab
RING = 4              # LDS ring slots
DEPTH = 4             # fragment registers (read-ahead 3 chunks)
VM_WAIT = 4           # s_waitcnt vmcnt at a slab start: (AHEAD - 2) * pieces per wave

XDL_WRITE_TO_OTHER = 8    # wait states, v_mfma_f32_16x16x32 D -> VALU / LDS / VMEM access or MFMA A/B read
VALU_WRITE_TO_XDL = 2     # VALU (incl. v_accvgpr_write) write -> MFMA operand read
XDL_READC_TO_WRITE = 3    # MFMA C read -> another instruction overwrites the register


def R(f, base, n=1):
    return (f, base, n)


def regs_of(r):
    f, b, n = r
    return {(f, b + i) for i in range(n)}


def fmt(r):
    f, b, n = r
    return f"{f}{b}" if n == 1 else f"{f}[{b}:{b + n - 1}]"


def ACC(par, t): return R('v', 16 * par + 4 * t, 4)
def FR(i): return R('v', 32 + 4 * (i % DEPTH), 4)
BIAS = R('v', 48, 4)
TMP = [R('v', 52), R('v', 54)]
VOFF = R('v', 53)
def XS(t, kb): return R('v', 64 + 8 * t + 4 * kb, 4)
def SETA(t, kb): return R('a', 32 * t + 4 * kb, 4)
def SETV(t, kb): return R('v', 128 + 32 * t + 4 * kb, 4)


class Ins:
    __slots__ = ("text", "kind", "reads", "writes", "creads", "ws")

    def __init__(self, text, kind, reads=(), writes=(), creads=(), ws=1):
        self.text, self.kind, self.ws = text, kind, ws
        self.reads = set().union(*[regs_of(r) for r in reads]) if reads else set()
        self.writes = set().union(*[regs_of(r) for r in writes]) if writes else set()
        self.creads = set().union(*[regs_of(r) for r in creads]) if creads else set()


class Emitter:
    """Appends instructions, inserting s_nop for the MFMA hazards and s_waitcnt lgkmcnt for LDS results."""

    def __init__(self, dt):
        self.dt = dt
        self.ins = []
        self.pos = 0                     # wait states issued so far
        self.xdl_write = {}              # reg -> position after the MFMA that wrote it
        self.valu_write = {}             # reg -> position after the VALU that wrote it
        self.xdl_cread = {}              # reg -> position after the MFMA that read it as C
        self.lgkm = []                   # outstanding LDS reads (sets of dst regs), in issue order

    def _push(self, i):
        self.ins.append(i)
        self.pos += i.ws

    def nop(self, n):                    # n wait states
        while n > 0:
            k = min(n, 8)
            self._push(Ins(f"s_nop {k - 1}", "nop", ws=k))
            n -= k

    def _need(self, table, regs, dist):
        need = 0
        for r in regs:
            if r in table:
                need = max(need, table[r] + dist - self.pos)
        return need

    def wait_lds(self, regs):
        """the LDS reads that produce `regs` have returned"""
        last = -1
        for i, dst in enumerate(self.lgkm):
            if dst & regs:
                last = i
        if last < 0:
            return
        left = len(self.lgkm) - last - 1
        self._push(Ins(f"s_waitcnt lgkmcnt({left})", "wait"))
        self.lgkm = self.lgkm[last + 1:]

    def mfma(self, d, a, b, c):
        rd = regs_of(a) | regs_of(b) | regs_of(c)
        self.wait_lds(rd | regs_of(d))
        n = self._need(self.valu_write, rd, VALU_WRITE_TO_XDL)
        n = max(n, self._need(self.xdl_write, regs_of(a) | regs_of(b), XDL_WRITE_TO_OTHER))
        if c != d:                       # an untied C that an MFMA has just written, or a D another MFMA is reading as C
            n = max(n, self._need(self.xdl_write, regs_of(c), XDL_WRITE_TO_OTHER))
            n = max(n, self._need(self.xdl_cread, regs_of(d), XDL_READC_TO_WRITE))
        self.nop(n)
        self._push(Ins(f"v_mfma_f32_16x16x32_{self.dt} {fmt(d)}, {fmt(a)}, {fmt(b)}, {fmt(c)}", "mfma",
                       reads=(a, b), writes=(d,), creads=(c,)))
        for r in regs_of(d):
            self.xdl_write[r] = self.pos
        for r in regs_of(c):
            self.xdl_cread[r] = self.pos

    def _other(self, text, kind, reads, writes):
        rd = set().union(*[regs_of(r) for r in reads]) if reads else set()
        wr = set().union(*[regs_of(r) for r in writes]) if writes else set()
        self.wait_lds(rd | wr)
        n = self._need(self.xdl_write, rd | wr, XDL_WRITE_TO_OTHER)
        n = max(n, self._need(self.xdl_cread, wr, XDL_READC_TO_WRITE))
        self.nop(n)
        self._push(Ins(text, kind, reads=reads, writes=writes))
        return wr

    def valu(self, text, reads, writes):
        wr = self._other(text, "valu", reads, writes)
        for r in wr:
            self.valu_write[r] = self.pos

    def lds_read(self, dst, addr_operand, offset):
        assert 0 <= offset < 65536
        wr = self._other(f"ds_read_b128 {fmt(dst)}, {addr_operand}" + (f" offset:{offset}" if offset else ""), "lds", (), (dst,))
        self.lgkm.append(wr)
        assert len(self.lgkm) < 15

    def salu(self, text):
        self._push(Ins(text, "salu"))

    def dma(self, offset):
        self._other(f"global_load_lds_dwordx4 {fmt(VOFF)}, %[sbase]" + (f" offset:{offset}" if offset else ""), "dma", (VOFF,), ())

    def drain_lds(self):
        if self.lgkm:
            self._push(Ins("s_waitcnt lgkmcnt(0)", "wait"))
            self.lgkm = []


def check(ins):
    """Independent re-verification of the finished stream: hazards by wait-state distance, LDS results by lgkmcnt."""
    pos = 0
    xw, vw, xc = {}, {}, {}
    fifo = []
    for i in ins:
        rd_all = i.reads | i.creads
        if i.kind == "wait" and "lgkmcnt" in i.text:
            left = int(i.text.split("lgkmcnt(")[1].split(")")[0])
            while len(fifo) > left:
                fifo.pop(0)
        touched = rd_all | i.writes
        for dst in fifo:
            assert not (dst & touched), f"LDS result not waited for: {i.text}"
        if i.kind == "mfma":
            ab = i.reads
            for r in rd_all:
                assert pos - vw.get(r, -99) >= VALU_WRITE_TO_XDL, f"VALU->MFMA: {i.text}"
            for r in ab:
                assert pos - xw.get(r, -99) >= XDL_WRITE_TO_OTHER, f"MFMA D->A/B: {i.text}"
            if i.creads != i.writes:
                for r in i.creads:
                    assert pos - xw.get(r, -99) >= XDL_WRITE_TO_OTHER, f"MFMA D->untied C: {i.text}"
                for r in i.writes:
                    assert pos - xc.get(r, -99) >= XDL_READC_TO_WRITE, f"MFMA C read->D write: {i.text}"
        elif i.kind in ("valu", "lds", "dma"):
            for r in touched:
                assert pos - xw.get(r, -99) >= XDL_WRITE_TO_OTHER, f"MFMA D->{i.kind}: {i.text}"
            for r in i.writes:
                assert pos - xc.get(r, -99) >= XDL_READC_TO_WRITE, f"MFMA C read->overwrite: {i.text}"
        pos += i.ws
        if i.kind == "mfma":
            for r in i.writes:
                xw[r] = pos
            for r in i.creads:
                xc[r] = pos
        if i.kind == "valu":
            for r in i.writes:
                vw[r] = pos
        if i.kind == "lds":
            fifo.append(i.writes)
    assert not fifo, "LDS reads outstanding at the end of the statement"


def issue_cycles(ins):
    """rough issue-port model of one wave: an MFMA holds the port 8 cycles, anything else 4 (LDS-DMA ~30)"""
    c = 0
    for i in ins:
        c += {"mfma": 8, "dma": 30, "nop": 4 * i.ws}.get(i.kind, 4)
    return c


def gen_layer(dt, in_a, skip, nsb=16, nkb_h=8):
    """One hidden layer, ReLU: set A -> set V (in_a) or set V -> set A; skip: K-blocks 0, 1 are the embedded point."""
    nkb = nkb_h + (2 if skip else 0)
    total = nsb * nkb
    assert total % SLAB == 0 and total % DEPTH == 0
    slabs = total // SLAB
    IN = SETA if in_a else SETV
    OUT = SETV if in_a else SETA
    cvt = "v_cvt_pk_bf16_f32" if dt == "bf16" else "v_cvt_pk_f16_f32"
    e = Emitter(dt)
    e.salu("s_mov_b32 %[keep], m0")
    e.lds_read(BIAS, "%[bias]", 0)
    e.nop(VALU_WRITE_TO_XDL)             # the compiler's own VALU writes of our operands

    def conv_piece(s, piece, gap_accw):
        """dword `J` of finished sub-block s, tile t -> dword 2 (s & 1) + J of K-block s >> 1 of the output set"""
        t, J = piece % T, piece // T
        a = ACC(s & 1, t)
        tmp = TMP[piece & 1]
        dst = OUT(t, s >> 1)
        dword = R(dst[0], dst[1] + 2 * (s & 1) + J)
        e.valu(f"{cvt} {fmt(tmp)}, {fmt(R('v', a[1] + 2 * J))}, {fmt(R('v', a[1] + 2 * J + 1))}",
               (R('v', a[1] + 2 * J, 2),), (tmp,))
        if in_a:
            e.valu(f"v_pk_max_i16 {fmt(dword)}, {fmt(tmp)}, 0", (tmp,), (dword,))
        else:
            e.valu(f"v_pk_max_i16 {fmt(tmp)}, {fmt(tmp)}, 0", (tmp,), (tmp,))
            gap_accw.append((dword, tmp))

    pending_salu = []
    for p in range(total):
        sb, kc = divmod(p, nkb)
        par = sb & 1
        c = p % SLAB
        frag = FR(p)

        def operand(t):
            if skip:
                return XS(t, kc) if kc < 2 else IN(t, kc - 2)
            return IN(t, kc)

        def mm(t):
            e.mfma(ACC(par, t), frag, operand(t), BIAS if kc == 0 else ACC(par, t))

        accw = []
        mm(0)
        # ---- gap 0: slab bookkeeping
        if c == 0:
            e.salu(f"s_waitcnt vmcnt({VM_WAIT})")
            e.salu("s_barrier")
            e.valu(f"v_lshl_add_u32 {fmt(VOFF)}, %[islab], 14, %[loff]", (), (VOFF,))
            e.salu("s_add_u32 m0, %[dsto], %[ldsw]")
            pending_salu = ["s_add_i32 %[islab], %[islab], 1", "s_cmp_lg_u32 %[islab], %[nsl]",
                            "s_cselect_b32 %[islab], %[islab], 0", "s_add_i32 %[dsto], %[dsto], 0x4000",
                            "s_and_b32 %[dsto], %[dsto], 0xc000"]
        elif c >= 4 and pending_salu:
            e.salu(pending_salu.pop(0))
        mm(1)
        # ---- gap 1: fragment read-ahead (chunk p + 3 into the register of chunk p - 1)
        q = p + DEPTH - 1
        e.lds_read(FR(q), f"%[rb{(q // SLAB) % RING}]", (q % SLAB) * 1024)
        mm(2)
        # ---- gap 2: conversion of the previous sub-block, one piece per chunk step
        if sb > 0:
            pps = (2 * T + nkb - 1) // nkb
            for i in range(pps):
                piece = kc * pps + i
                if piece < 2 * T:
                    conv_piece(sb - 1, piece, accw)
        mm(3)
        # ---- gap 3
        for dword, tmp in accw:
            e.valu(f"v_accvgpr_write_b32 {fmt(dword)}, {fmt(tmp)}", (tmp,), (dword,))
        if c < 4:
            e.dma(c * 1024)
        if kc == min(3, nkb - 1) and sb + 1 < nsb:
            e.lds_read(BIAS, "%[bias]", 64 * (sb + 1))
    assert not pending_salu
    # ---- tail: the last sub-block
    accw = []
    for piece in range(2 * T):
        conv_piece(nsb - 1, piece, accw)
        for dword, tmp in accw:
            e.valu(f"v_accvgpr_write_b32 {fmt(dword)}, {fmt(tmp)}", (tmp,), (dword,))
        accw.clear()
    e.drain_lds()
    e.salu("s_mov_b32 m0, %[keep]")
    e.nop(VALU_WRITE_TO_XDL)             # our VALU / accvgpr writes ahead of whatever MFMA the compiler issues next
    check(e.ins)
    return e, slabs


def cpp_function(name, dt, in_a, skip, e, slabs):
    m = "Mma16BF16" if dt == "bf16" else "Mma16F16"
    text = "\\n\\t\"\n      \"".join(i.text for i in e.ins)
    outs, ins = [], []
    for t in range(T):
        for kb in range(8):
            k = 8 * t + kb
            if in_a:
                outs.append(f'"={{{fmt(SETV(t, kb))}}}"(V[{k}])')
                ins.append(f'"{{{fmt(SETA(t, kb))}}}"(A[{k}])')
            else:
                outs.append(f'"={{{fmt(SETA(t, kb))}}}"(A[{k}])')
                ins.append(f'"{{{fmt(SETV(t, kb))}}}"(V[{k}])')
    for i in range(DEPTH):
        outs.append(f'"+{{{fmt(FR(i))}}}"(F[{i}])')
    outs += ['[islab] "+s"(islab)', '[dsto] "+s"(dsto)', '[keep] "=&s"(keep)']
    if skip:
        for t in range(T):
            for kb in range(2):
                ins.append(f'"{{{fmt(XS(t, kb))}}}"(X[{2 * t + kb}])')
    ins += ['[rb0] "v"(rb0)', '[rb1] "v"(rb1)', '[rb2] "v"(rb2)', '[rb3] "v"(rb3)', '[bias] "v"(bias)', '[loff] "v"(loff)',
            '[sbase] "s"(sbase)', '[nsl] "s"(nsl)', '[ldsw] "s"(ldsw)']
    clob = ['"memory"', '"scc"'] + [f'"v{i}"' for i in list(range(0, 32)) + list(range(48, 56))]
    n_mfma = sum(i.kind == "mfma" for i in e.ins)
    cyc = issue_cycles(e.ins)
    return f"""
// {name}: {len(e.ins)} instructions, {n_mfma} MFMAs, {slabs} slabs; issue-port estimate {cyc} cycles = {cyc / slabs:.0f} per slab
// (matrix pipe: {16 * n_mfma / slabs:.0f}); s_nop {sum(i.kind == 'nop' for i in e.ins)}, s_waitcnt {sum(i.kind == 'wait' for i in e.ins)}
template <> struct HiddenAsm<{m}, {'true' if in_a else 'false'}, {'true' if skip else 'false'}> {{
  static constexpr int kSlabs = {slabs};
  static __device__ __forceinline__ void run(u32x4 (&A)[32], u32x4 (&V)[32], const u32x4 (&X)[8], u32x4 (&F)[4],
                                             uint32_t rb0, uint32_t rb1, uint32_t rb2, uint32_t rb3, uint32_t bias, uint32_t loff,
                                             uint64_t sbase, uint32_t nsl, uint32_t ldsw, uint32_t& islab, uint32_t& dsto) {{
    uint32_t keep;
    asm volatile(
      "{text}"
      : {', '.join(outs)}
      : {', '.join(ins)}
      : {', '.join(clob)});
  }}
}};
"""


HEADER = """// GENERATED by tools/gen_ob16_asm.py -- do not edit; regenerate with `python tools/gen_ob16_asm.py`.
// Hand-scheduled hidden layers of the 16x16x32 engine (register map and rules: the generator's docstring).
#pragma once
namespace nsmlp {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <class M, bool IN_A, bool SKIP> struct HiddenAsm;
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nerf_sampling_amd", "csrc", "ns_ob16_asm.inc"))
    ap.add_argument("--dump", help="write the plain instruction stream of one variant (e.g. bf16_AV) here")
    a = ap.parse_args()
    out = [HEADER]
    for dt in ("bf16", "f16"):
        for in_a in (True, False):
            for skip in (False, True):
                e, slabs = gen_layer(dt, in_a, skip)
                name = f"{dt} {'A->V' if in_a else 'V->A'}{' skip' if skip else ''}"
                out.append(cpp_function(name, dt, in_a, skip, e, slabs))
                print(f"{name}: {len(e.ins)} instr, issue estimate {issue_cycles(e.ins) / slabs:.0f} cycles/slab, "
                      f"nops {sum(i.kind == 'nop' for i in e.ins)}, waits {sum(i.kind == 'wait' for i in e.ins)}", file=sys.stderr)
                if a.dump and a.dump == f"{dt}_{'AV' if in_a else 'VA'}{'_skip' if skip else ''}":
                    open(a.dump + ".s", "w").write("\n".join(i.text for i in e.ins) + "\n")
    out.append("}  // namespace nsmlp\n")
    with open(a.o, "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main()
