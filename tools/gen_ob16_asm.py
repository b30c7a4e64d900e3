#!/usr/bin/env python3
"""Generator of nerf_sampling_amd/csrc/ns_ob16_asm.inc: hand-scheduled gfx950 instruction streams for the hidden layers
of the 16x16x32 MLP engine (layer_ob16<> / layer_ob16x3<> in ns_mlp_engine.h are the compiler-scheduled statements of the
same layers).

Why: a lone wave per SIMD issues in order, and a v_mfma_f32_16x16x32 holds the vector issue port for 8 of its 16 cycles
(MI355X_MICROARCH.md, cycle constants), so a chunk step (4 MFMAs = 64 matrix cycles) has room for ~8 single-issue
instructions -- IF they sit evenly between the MFMAs.  The compiler's schedule bunches the conversions, recycles
accumulator registers as fragment-read targets (s_nop pads) and waits on lgkmcnt in front of most MFMA groups; the
kernel spends 91 cycles per chunk step instead of 64.  This generator emits the same arithmetic (same MFMA order per
accumulator: results are bit-identical to the compiled layers) as one asm statement per layer with a fixed register map.
Four-tile wave (64 samples; Map4):

  v[0:31]    accumulators  acc[parity][tile] = v[16*parity + 4*tile : +3]     (clobbered)
  v[32:47]   A fragments   f[i] = v[32 + 4 i : +3], chunk p lives in f[p % 4]  (in/out: the ring's read-ahead)
  v[48:51]   bias of the sub-block about to start (C operand of its first MFMAs), v[52:55] scratch   (clobbered)
  v[64:95]   the skip layer's embedded point, xs[tile][kb] = v[64 + 8*tile + 4*kb : +3]          (input, skip only)
  a[0:127]   activation set A, hA[tile][kb] = a[32*tile + 4*kb : +3]
  v[128:255] activation set V, hB[tile][kb] = v[128 + 32*tile + 4*kb : +3]

Five-tile wave (80 samples; Map5): accumulators v[0:39], fragments v[40:55], bias v[56:59], scratch v[60:63], set A
a[0:159], the skip layer's point a[160:199], set V v[96:255]; the compiler keeps v[64:95].  The split-operand (f16x3)
layers use the four-tile map with (hi, lo) tuple pairs per K-block and two tiles (gen_layer_x3).

Layers alternate A -> V (conversion results are written by VALU straight into set V) and V -> A (one extra
v_accvgpr_write per dword).  Outputs are declared early-clobber: they are written while inputs are still being read (a
plain "=" let the compiler put an address operand inside the output set).  Hazards follow the rules the compiler applies
to this MFMA on gfx950 (read off its own output): D write -> any other access 8 wait states, VALU write -> MFMA read 2,
C read -> overwrite 3, plus a conservative clock model (class Clock); the emitter inserts s_nop / counted lgkmcnt where a
rule needs it and an independent checker pass re-verifies the final text (tests/test_asm_generator.py).

gen_layer() is the layout in use (one dword converted per chunk step, a three-stage pipeline); gen_layer_q() issues the
same micro-ops from a queue, one per MFMA gap -- measured 2-4 % slower and kept for that A/B
(--queue4 / --queue5).  --exp-* flags build timing-only ablations (results wrong on purpose).
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


# Conservative clock model on top of the wait-state rules.  The compiler's rules count issue slots; behind a back-to-back
# MFMA stream execution can lag issue, and an s_nop is the cheapest slot there is.  So the emitter and the checker also
# keep a clock: an MFMA holds the issue port 8 cycles and the matrix pipe 16, it starts when the pipe is free and may
# issue while at most one other MFMA is waiting for the pipe; its result is readable XDL_RESULT_MARGIN cycles after it
# leaves the pipe.  Anything else issues in 4 cycles, s_nop k in k + 1 (a pad is only trusted for one cycle per count).
MFMA_ISSUE, MFMA_PIPE, OTHER_ISSUE, XDL_RESULT_MARGIN = 8, 16, 4, 16


class Clock:
    def __init__(self):
        self.t = 0                       # issue clock
        self.pipe_free = 0               # when the matrix pipe takes the next MFMA
        self.prev_start = 0              # execution start of the previous MFMA
        self.ready = {}                  # register -> cycle its MFMA result can be read

    def issue(self, ins):
        """advance over one instruction; returns the cycle at which it issues"""
        if ins.kind == "mfma":
            t0 = max(self.t, self.prev_start)          # issues once the previous MFMA has entered the pipe
            start = max(t0 + MFMA_ISSUE, self.pipe_free)
            self.prev_start = start
            self.pipe_free = start + MFMA_PIPE
            for r in ins.writes:
                self.ready[r] = start + MFMA_PIPE + XDL_RESULT_MARGIN
            self.t = t0 + MFMA_ISSUE
            return t0
        t0 = self.t
        self.t += ins.ws if ins.kind == "nop" else OTHER_ISSUE
        return t0

    def need(self, regs):
        """cycles still missing before `regs` (MFMA results) may be touched by a non-MFMA instruction issued now"""
        return max([self.ready.get(r, 0) - self.t for r in regs] + [0])


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


class Opt:
    """schedule knobs and timing-only ablations (--exp-*: results are wrong on purpose)"""
    no_wait = False      # --exp-no-wait: no lgkmcnt waits inside the layer
    no_dma = False       # --exp-no-dma
    no_conv = False      # --exp-no-conv
    no_lds = False       # --exp-no-lds: no fragment reads
    no_barrier = False   # --exp-no-barrier
    dma_steps = (0, 2, 4, 6)   # --dma-steps: chunk steps of a slab whose G3 carries one of the four refill pieces


OPT = Opt()


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
        self.clock = Clock()

    def _push(self, i):
        self.ins.append(i)
        self.pos += i.ws
        self.clock.issue(i)

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
        if OPT.no_wait:
            self.lgkm = self.lgkm[last + 1:]
            return
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
        n = max(n, self.clock.need(rd | wr))            # the clock model, one cycle per pad count
        self.nop(n)
        self._push(Ins(text, kind, reads=reads, writes=writes))
        return wr

    def valu(self, text, reads, writes):
        wr = self._other(text, "valu", reads, writes)
        for r in wr:
            self.valu_write[r] = self.pos

    def lds_read(self, dst, addr_operand, offset):
        assert 0 <= offset < 65536
        if OPT.no_lds and "rb" in addr_operand:
            return
        wr = self._other(f"ds_read_b128 {fmt(dst)}, {addr_operand}" + (f" offset:{offset}" if offset else ""), "lds", (), (dst,))
        self.lgkm.append(wr)
        assert len(self.lgkm) < 15

    def salu(self, text):
        self._push(Ins(text, "salu"))

    def dma(self, offset, voff=None):
        if OPT.no_dma:
            return
        voff = voff or VOFF
        self._other(f"global_load_lds_dwordx4 {fmt(voff)}, %[sbase]" + (f" offset:{offset}" if offset else ""), "dma", (voff,), ())

    def settle(self, regs):
        """pads until MFMA results in `regs` can be read by whatever follows the statement"""
        rs = set().union(*[regs_of(r) for r in regs])
        self.nop(max(self._need(self.xdl_write, rs, XDL_WRITE_TO_OTHER), self.clock.need(rs)))

    def drain_lds(self):
        if self.lgkm:
            self._push(Ins("s_waitcnt lgkmcnt(0)", "wait"))
            self.lgkm = []


def check(ins):
    """Independent re-verification of the finished stream: hazards by wait-state distance, LDS results by lgkmcnt."""
    pos = 0
    xw, vw, xc = {}, {}, {}
    fifo = []
    clk = Clock()
    for i in ins:
        if i.kind in ("valu", "lds", "dma"):
            assert clk.need(i.reads | i.writes) == 0, f"clock model: MFMA result not written back yet: {i.text}"
        clk.issue(i)

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


def gen_layer(dt, in_a, skip, m=None, nsb=16, nkb_h=8, act="relu"):
    """One hidden layer: set A -> set V (in_a) or set V -> set A; skip: K-blocks 0, 1 are the embedded point.
    act = "relu" (the radiance field: a packed signed-16-bit max after the conversion) or "leaky" (the DepthNet's
    LeakyReLU(0.01), fp16 only, on the PACKED value: v_pk_mul_f16 by 0.01, v_pk_max_f16 -- one more pipeline stage, issued
    in G0 / G1 of the step after the conversion; the compiled layer applies the same two packed operations).

    A chunk step is  [wait] M0 <G0> M1 <G1> M2 <G2> M3 <G3>  (M = the four tiles' MFMAs of one A fragment).  What rides in
    the gaps: G1 the fragment read three chunks ahead; the conversion of the previous sub-block, one dword per step, as a
    three-stage pipeline so that no VALU instruction directly follows the one it depends on (v_cvt_pk in G2, v_pk_max_i16
    in the NEXT step's G0, v_accvgpr_write in that step's G2); slab bookkeeping (vmcnt + barrier in G0 of a slab's first
    step, its four LDS-DMA pieces in G3 of steps 0..3, scalar updates after them), the next sub-block's bias in G3."""
    m = m or Map4
    T, ACC, FR, BIAS, TMP, VOFF, XS = m.T, m.ACC, m.FR, m.BIAS, m.TMP, m.VOFF, m.XS     # (shadow the four-tile globals)
    nkb = nkb_h + (2 if skip else 0)
    total = nsb * nkb
    assert total % SLAB == 0 and total % DEPTH == 0
    slabs = total // SLAB
    IN = m.SETA if in_a else m.SETV
    OUT = m.SETV if in_a else m.SETA
    cvt = "v_cvt_pk_bf16_f32" if dt == "bf16" else "v_cvt_pk_f16_f32"
    leaky = act == "leaky"
    assert act in ("relu", "leaky") and not (leaky and (dt != "f16" or T != 4 or skip)), "LeakyReLU streams: fp16, four tiles"
    TMP2 = R('v', 55)                    # the scaled copy of the piece in flight (Map4: v55 is free)
    e = Emitter(dt)
    e.salu("s_mov_b32 %[keep], m0")
    e.lds_read(BIAS, "%[bias]", 0)

    def dma_setup():
        e.valu(f"v_lshl_add_u32 {fmt(VOFF)}, %[islab], 14, %[loff]", (), (VOFF,))
        e.salu("s_add_u32 m0, %[dsto], %[ldsw]")

    dma_setup()
    e.nop(VALU_WRITE_TO_XDL)             # the compiler's own VALU writes of our operands

    def piece_regs(s, piece):
        t, J = piece % T, piece // T
        a = ACC(s & 1, t)
        dst = OUT(t, s >> 1)
        return R('v', a[1] + 2 * J), R('v', a[1] + 2 * J + 1), R(dst[0], dst[1] + 2 * (s & 1) + J)

    def op_cvt(s, piece, tmp):
        lo, hi, _ = piece_regs(s, piece)
        e.valu(f"{cvt} {fmt(tmp)}, {fmt(lo)}, {fmt(hi)}", (lo, hi), (tmp,))

    def op_scale(s, piece, tmp):         # LeakyReLU only: 0.01 x, packed
        e.valu(f"v_pk_mul_f16 {fmt(TMP2)}, {fmt(tmp)}, %[slope]", (tmp,), (TMP2,))

    def op_max(s, piece, tmp):
        dword = piece_regs(s, piece)[2]
        dst = dword if in_a else tmp
        if leaky:
            e.valu(f"v_pk_max_f16 {fmt(dst)}, {fmt(tmp)}, {fmt(TMP2)}", (tmp, TMP2), (dst,))
        else:
            e.valu(f"v_pk_max_i16 {fmt(dst)}, {fmt(tmp)}, 0", (tmp,), (dst,))

    def op_accw(s, piece, tmp):
        if not in_a:
            dword = piece_regs(s, piece)[2]
            e.valu(f"v_accvgpr_write_b32 {fmt(dword)}, {fmt(tmp)}", (tmp,), (dword,))

    # one piece per chunk step in the main pipeline; a five-tile wave has ten pieces per sub-block: with eight steps the
    # last two go through gap 3 one after the other (cvt / max / accw at steps 1 / 2 / 3 and 4 / 5 / 6, scratch m.TMP3)
    n_main = min(2 * T, nkb)
    extras = list(range(n_main, 2 * T))
    assert len(extras) <= 2 and (not extras or T == 5)
    max_q, accw_q = [], []               # pieces whose v_pk_max / v_accvgpr_write is due
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

        def g_scale():
            if leaky and max_q and not OPT.no_conv:
                op_scale(*max_q[0])

        def g_max():
            if max_q and not OPT.no_conv:
                op_max(*max_q[0])
                accw_q.append(max_q.pop(0))

        def g_read():                    # fragment read-ahead (chunk p + 3 into the register of chunk p - 1)
            q = p + DEPTH - 1
            e.lds_read(FR(q), f"%[rb{(q // SLAB) % RING}]", (q % SLAB) * 1024)

        mm(0)
        # ---- G0
        if c == 0 and not OPT.no_barrier:
            e.salu(f"s_waitcnt vmcnt({VM_WAIT})")
            e.salu("s_barrier")
        if leaky:
            g_scale()
        else:
            g_max()
        mm(1)
        # ---- G1
        g_read()
        if leaky:
            g_max()
        mm(2)
        # ---- G2
        if sb > 0 and kc < n_main and not OPT.no_conv:
            item = (sb - 1, kc, TMP[kc & 1])
            op_cvt(*item)
            max_q.append(item)
        if accw_q:
            op_accw(*accw_q.pop(0))
        mm(3)
        if T == 5:
            # ---- G3 of a five-tile step: the two extra pieces
            if sb > 0 and not OPT.no_conv:
                for n, x in enumerate(extras):
                    stage = kc - (1 + 3 * n)
                    if stage == 0:
                        op_cvt(sb - 1, x, m.TMP3)
                    elif stage == 1:
                        op_max(sb - 1, x, m.TMP3)
                    elif stage == 2:
                        op_accw(sb - 1, x, m.TMP3)
            mm(4)
        # ---- last gap
        if c in OPT.dma_steps:
            e.dma(OPT.dma_steps.index(c) * 1024, VOFF)
            if c == OPT.dma_steps[-1]:
                pending_salu = ["s_add_i32 %[islab], %[islab], 1", "s_cmp_lg_u32 %[islab], %[nsl]",
                                "s_cselect_b32 %[islab], %[islab], 0", "s_add_i32 %[dsto], %[dsto], 0x4000",
                                "s_and_b32 %[dsto], %[dsto], 0xc000"]
        elif pending_salu:
            e.salu(pending_salu.pop(0))
        elif c == SLAB - 1 and p + 1 < total:
            dma_setup()                  # address and M0 of the next slab's refill
        if kc == min(3, nkb - 1) and sb + 1 < nsb:
            e.lds_read(BIAS, "%[bias]", 64 * (sb + 1))
    assert not pending_salu
    # ---- tail: what is left of the conversion pipeline, then the last sub-block through 8 scratch registers (the other
    # parity's accumulators are free), stage by stage
    if not OPT.no_conv:
        for it in max_q:
            if leaky:
                op_scale(*it)
            op_max(*it)
        for it in accw_q + max_q:
            op_accw(*it)
        s = nsb - 1
        scratch = [R('v', ACC((s & 1) ^ 1, 0)[1] + i) for i in range(2 * T)]
        for t0 in range(0, T, 2):        # two tiles at a time, the tile whose last MFMA is oldest first
            pcs = [t + T * J for t in (t0, t0 + 1) if t < T for J in (0, 1)]
            for piece in pcs:
                op_cvt(s, piece, scratch[piece])
            for piece in pcs:
                if leaky:
                    op_scale(s, piece, scratch[piece])
                op_max(s, piece, scratch[piece])
            for piece in pcs:
                op_accw(s, piece, scratch[piece])
    e.drain_lds()
    e.salu("s_mov_b32 m0, %[keep]")
    e.nop(VALU_WRITE_TO_XDL)             # our VALU / accvgpr writes ahead of whatever MFMA the compiler issues next
    if not (OPT.no_wait or OPT.no_lds):
        check(e.ins)
    return e, slabs


class Map5:
    """register map of the five-tile wave (80 samples): 160 + 160 activation registers, 40 accumulators"""
    T = 5
    @staticmethod
    def ACC(par, t): return R('v', 20 * par + 4 * t, 4)
    @staticmethod
    def FR(i): return R('v', 40 + 4 * (i % DEPTH), 4)
    BIAS = R('v', 56, 4)
    TMP = [R('v', 60), R('v', 62)]
    TMP3 = R('v', 63)
    VOFF = R('v', 61)
    @staticmethod
    def XS(t, kb): return R('a', 160 + 8 * t + 4 * kb, 4)
    @staticmethod
    def SETA(t, kb): return R('a', 32 * t + 4 * kb, 4)
    @staticmethod
    def SETV(t, kb): return R('v', 96 + 32 * t + 4 * kb, 4)
    CLOBBER = list(range(0, 40)) + list(range(56, 64))


class Map4:
    T = 4
    ACC, FR, XS, SETA, SETV = staticmethod(ACC), staticmethod(FR), staticmethod(XS), staticmethod(SETA), staticmethod(SETV)
    BIAS, TMP, VOFF = BIAS, TMP, VOFF
    CLOBBER = list(range(0, 32)) + list(range(48, 56))


def gen_layer_q(dt, in_a, skip, m, nsb=16, nkb_h=8):
    """gen_layer() for any tile count (register map m): the conversion is a queue of VALU micro-ops, stage-interleaved in
    pairs of dwords (cvt, cvt, max, max[, accw, accw]) so that dependent instructions are two gaps apart, one per MFMA gap."""
    TT = m.T
    nkb = nkb_h + (2 if skip else 0)
    total = nsb * nkb
    assert total % SLAB == 0 and total % DEPTH == 0
    slabs = total // SLAB
    IN = m.SETA if in_a else m.SETV
    OUT = m.SETV if in_a else m.SETA
    cvt = "v_cvt_pk_bf16_f32" if dt == "bf16" else "v_cvt_pk_f16_f32"
    e = Emitter(dt)
    e.salu("s_mov_b32 %[keep], m0")
    e.lds_read(m.BIAS, "%[bias]", 0)

    def dma_setup():
        e.valu(f"v_lshl_add_u32 {fmt(m.VOFF)}, %[islab], 14, %[loff]", (), (m.VOFF,))
        e.salu("s_add_u32 m0, %[dsto], %[ldsw]")

    def dma(offset):
        if not OPT.no_dma:
            e._other(f"global_load_lds_dwordx4 {fmt(m.VOFF)}, %[sbase]" + (f" offset:{offset}" if offset else ""), "dma", (m.VOFF,), ())

    dma_setup()
    e.nop(VALU_WRITE_TO_XDL)

    def piece_ops(s, pieces, tmps):
        """micro-ops of up to two dwords (pieces) of finished sub-block s, stage by stage"""
        stages = [[], [], []]
        for piece, tmp in zip(pieces, tmps):
            t, J = piece % TT, piece // TT
            a = m.ACC(s & 1, t)
            lo, hi = R('v', a[1] + 2 * J), R('v', a[1] + 2 * J + 1)
            dst = OUT(t, s >> 1)
            dword = R(dst[0], dst[1] + 2 * (s & 1) + J)
            stages[0].append(lambda lo=lo, hi=hi, tmp=tmp: e.valu(f"{cvt} {fmt(tmp)}, {fmt(lo)}, {fmt(hi)}", (lo, hi), (tmp,)))
            if in_a:
                stages[1].append(lambda dword=dword, tmp=tmp: e.valu(f"v_pk_max_i16 {fmt(dword)}, {fmt(tmp)}, 0", (tmp,), (dword,)))
            else:
                stages[1].append(lambda tmp=tmp: e.valu(f"v_pk_max_i16 {fmt(tmp)}, {fmt(tmp)}, 0", (tmp,), (tmp,)))
                stages[2].append(lambda dword=dword, tmp=tmp: e.valu(f"v_accvgpr_write_b32 {fmt(dword)}, {fmt(tmp)}", (tmp,), (dword,)))
        return [op for st in stages for op in st]

    conv_q = []
    pending_salu = []

    def conv(n=1):
        for _ in range(n):
            if conv_q and not OPT.no_conv:
                conv_q.pop(0)()

    for p in range(total):
        sb, kc = divmod(p, nkb)
        par = sb & 1
        c = p % SLAB
        frag = m.FR(p)

        def operand(t):
            if skip:
                return m.XS(t, kc) if kc < 2 else IN(t, kc - 2)
            return IN(t, kc)

        # this step's share of the conversion queue, spread over its gaps (G1 carries the fragment read; a sub-block's
        # first two gaps stay free): an even pace matters -- the same micro-ops packed into the first steps of a sub-block
        # cost 4 % of the kernel (measured), the issue port is only just not the bottleneck
        quota = -(-len(conv_q) // (nkb - kc)) if conv_q else 0
        cands = [g for g in ([2, 0] + list(range(3, TT))) if g < TT and not (kc == 0 and g < 2)]
        share = {g: 0 for g in range(TT)}
        for i in range(quota):
            share[cands[i % len(cands)]] += 1
        for t in range(TT):
            e.mfma(m.ACC(par, t), frag, operand(t), m.BIAS if kc == 0 else m.ACC(par, t))
            if t == 0 and c == 0 and not OPT.no_barrier:
                e.salu(f"s_waitcnt vmcnt({VM_WAIT})")
                e.salu("s_barrier")
            if t == 1:
                q = p + DEPTH - 1
                e.lds_read(m.FR(q), f"%[rb{(q // SLAB) % RING}]", (q % SLAB) * 1024)
            conv(share[t])
            if t == TT - 1:
                if c in OPT.dma_steps:
                    dma(OPT.dma_steps.index(c) * 1024)
                    if c == OPT.dma_steps[-1]:
                        pending_salu = ["s_add_i32 %[islab], %[islab], 1", "s_cmp_lg_u32 %[islab], %[nsl]",
                                        "s_cselect_b32 %[islab], %[islab], 0", "s_add_i32 %[dsto], %[dsto], 0x4000",
                                        "s_and_b32 %[dsto], %[dsto], 0xc000"]
                elif pending_salu:
                    e.salu(pending_salu.pop(0))
                elif c == SLAB - 1 and p + 1 < total:
                    dma_setup()
                if kc == min(3, nkb - 1) and sb + 1 < nsb:
                    e.lds_read(m.BIAS, "%[bias]", 64 * (sb + 1))
        if kc == nkb - 1 and sb + 1 < nsb:
            assert not conv_q or OPT.no_conv, f"conversion queue did not drain within one sub-block ({len(conv_q)} left)"
            for i in range(0, 2 * TT, 2):
                conv_q.extend(piece_ops(sb, [i, i + 1], m.TMP))
    assert not pending_salu
    if not OPT.no_conv:
        while conv_q:
            conv_q.pop(0)()
        s = nsb - 1
        scratch = [R('v', m.ACC((s & 1) ^ 1, 0)[1] + i) for i in range(2 * TT)]
        for t0 in range(0, TT, 2):       # two tiles at a time, the tile whose last MFMA is oldest first
            pcs = [t + TT * J for t in (t0, t0 + 1) if t < TT for J in (0, 1)]
            for op in piece_ops(s, pcs, [scratch[q] for q in pcs]):
                op()
    e.drain_lds()
    e.salu("s_mov_b32 m0, %[keep]")
    e.nop(VALU_WRITE_TO_XDL)
    if not (OPT.no_wait or OPT.no_lds):
        check(e.ins)
    return e, slabs


def special_spec(kind, m):
    """the three layers of the production network that are not W -> W hidden layers (all write set A or only return raw
    accumulators): layer 0 (embedded point, 64 -> W), the view layer ((h, dirs) -> W/2 with the sigma head as the one
    extra, LAST sub-block, returned raw) and the rgb head (W/2 -> 3, one sub-block, returned raw)"""
    if kind == "layer0":
        return dict(nsb=16, nkb=2, operand=lambda t, kc: m.XS(t, kc), convert=True, convert_last=True, out_acc=False)
    if kind == "views":
        return dict(nsb=9, nkb=9, operand=lambda t, kc: m.SETV(t, kc) if kc < 8 else m.XS(t, 0), convert=True, convert_last=False, out_acc=True)
    if kind == "rgb":
        return dict(nsb=1, nkb=4, operand=lambda t, kc: m.SETA(t, kc), convert=False, convert_last=False, out_acc=True)
    raise ValueError(kind)


def gen_layer_special(dt, kind, m):
    """Layer 0, view layer, rgb head as statements (queue layout of gen_layer_q: together they are 9 of the 67 slabs).
    Their streams are not slab multiples: the chunks are padded to the fragment pipeline depth (pad steps carry no MFMA),
    the last slab may be short (its refill pieces ride on its first four steps)."""
    sp = special_spec(kind, m)
    TT, nsb, nkb = m.T, sp["nsb"], sp["nkb"]
    real = nsb * nkb
    total = -(-real // DEPTH) * DEPTH
    slabs = -(-total // SLAB)
    cvt = "v_cvt_pk_bf16_f32" if dt == "bf16" else "v_cvt_pk_f16_f32"
    e = Emitter(dt)
    e.salu("s_mov_b32 %[keep], m0")
    e.lds_read(m.BIAS, "%[bias]", 0)

    def dma_setup():
        e.valu(f"v_lshl_add_u32 {fmt(m.VOFF)}, %[islab], 14, %[loff]", (), (m.VOFF,))
        e.salu("s_add_u32 m0, %[dsto], %[ldsw]")

    dma_setup()
    e.nop(VALU_WRITE_TO_XDL)

    def piece_ops(s, pieces, tmps):
        stages = [[], [], []]
        for piece, tmp in zip(pieces, tmps):
            t, J = piece % TT, piece // TT
            a = m.ACC(s & 1, t)
            lo, hi = R('v', a[1] + 2 * J), R('v', a[1] + 2 * J + 1)
            dst = m.SETA(t, s >> 1)
            dword = R(dst[0], dst[1] + 2 * (s & 1) + J)
            stages[0].append(lambda lo=lo, hi=hi, tmp=tmp: e.valu(f"{cvt} {fmt(tmp)}, {fmt(lo)}, {fmt(hi)}", (lo, hi), (tmp,)))
            stages[1].append(lambda tmp=tmp: e.valu(f"v_pk_max_i16 {fmt(tmp)}, {fmt(tmp)}, 0", (tmp,), (tmp,)))
            stages[2].append(lambda dword=dword, tmp=tmp: e.valu(f"v_accvgpr_write_b32 {fmt(dword)}, {fmt(tmp)}", (tmp,), (dword,)))
        return [op for st in stages for op in st]

    conv_q, pending_salu = [], []

    def conv(n=1):
        for _ in range(n):
            if conv_q:
                conv_q.pop(0)()

    def read_ahead(p):
        q = p + DEPTH - 1
        if q < real:
            e.lds_read(m.FR(q), f"%[rb{(q // SLAB) % RING}]", (q % SLAB) * 1024)
        elif q >= total:                 # the next statement's first chunks, in the slab after this layer's last
            e.lds_read(m.FR(q), f"%[rb{slabs % RING}]", (q - total) * 1024)

    for p in range(total):
        c = p % SLAB
        slab_len = min(SLAB, total - (p - c))
        dma_steps = OPT.dma_steps if slab_len > OPT.dma_steps[-1] else (0, 1, 2, 3)
        last_of_slab = c == slab_len - 1

        def bookkeeping():
            nonlocal pending_salu
            if c in dma_steps:
                e.dma(dma_steps.index(c) * 1024, m.VOFF)
                if c == dma_steps[-1]:
                    pending_salu = ["s_add_i32 %[islab], %[islab], 1", "s_cmp_lg_u32 %[islab], %[nsl]",
                                    "s_cselect_b32 %[islab], %[islab], 0", "s_add_i32 %[dsto], %[dsto], 0x4000",
                                    "s_and_b32 %[dsto], %[dsto], 0xc000"]
            elif pending_salu:
                e.salu(pending_salu.pop(0))
            if last_of_slab:
                while pending_salu:
                    e.salu(pending_salu.pop(0))
                if p + 1 < total:
                    dma_setup()

        if p >= real:                    # pad step: no MFMA
            if c == 0 and not OPT.no_barrier:
                e.salu(f"s_waitcnt vmcnt({VM_WAIT})")
                e.salu("s_barrier")
            read_ahead(p)
            conv(len(conv_q))
            bookkeeping()
            continue
        sb, kc = divmod(p, nkb)
        par = sb & 1
        frag = m.FR(p)
        quota = -(-len(conv_q) // (nkb - kc)) if conv_q else 0
        cands = [g for g in ([2, 0] + list(range(3, TT))) if g < TT and not (kc == 0 and g < 2)]
        share = {g: 0 for g in range(TT)}
        for i in range(quota):
            share[cands[i % len(cands)]] += 1
        for t in range(TT):
            e.mfma(m.ACC(par, t), frag, sp["operand"](t, kc), m.BIAS if kc == 0 else m.ACC(par, t))
            if t == 0 and c == 0 and not OPT.no_barrier:
                e.salu(f"s_waitcnt vmcnt({VM_WAIT})")
                e.salu("s_barrier")
            if t == 1:
                read_ahead(p)
            conv(share[t])
            if t == TT - 1:
                bookkeeping()
                if kc == min(3, nkb - 1) and sb + 1 < nsb:
                    e.lds_read(m.BIAS, "%[bias]", 64 * (sb + 1))
        if kc == nkb - 1 and sp["convert"] and (sb + 1 < nsb):
            assert not conv_q, f"conversion queue did not drain within one sub-block ({len(conv_q)} left)"
            if sb + 1 < nsb or sp["convert_last"]:
                for i in range(0, 2 * TT, 2):
                    conv_q.extend(piece_ops(sb, [i, i + 1], m.TMP))
    assert not pending_salu
    while conv_q:
        conv_q.pop(0)()
    if sp["convert_last"]:
        s_ = nsb - 1
        scratch = [R('v', m.ACC((s_ & 1) ^ 1, 0)[1] + i) for i in range(2 * TT)]
        for t0 in range(0, TT, 2):
            pcs = [t + TT * J for t in (t0, t0 + 1) if t < TT for J in (0, 1)]
            for op in piece_ops(s_, pcs, [scratch[q] for q in pcs]):
                op()
    if sp["out_acc"]:
        e.settle([m.ACC((nsb - 1) & 1, t) for t in range(TT)])
    e.drain_lds()
    e.salu("s_mov_b32 m0, %[keep]")
    e.nop(VALU_WRITE_TO_XDL)
    if not (OPT.no_wait or OPT.no_lds):
        check(e.ins)
    return e, slabs


def cpp_special(dt, kind, e, slabs, mp):
    """struct SpecialAsm<M, NT, KIND>: KIND 0 = layer 0 (X -> set A), 1 = view layer (set V, D -> set A K-blocks 0..3, raw
    accumulators of the sigma sub-block), 2 = rgb head (set A K-blocks 0..3 -> raw accumulators)"""
    nt = mp.T
    mname = {"bf16": "Mma16BF16", "f16": "Mma16F16"}[dt]
    kid = {"layer0": 0, "views": 1, "rgb": 2}[kind]
    text = "\\n\\t\"\n      \"".join(i.text for i in e.ins)
    outs, ins, params = [], [], []
    acc_out = kind in ("views", "rgb")
    if kind == "layer0":
        params = [f"const u32x4 (&X)[{2 * nt}]", f"u32x4 (&A)[{8 * nt}]"]
        for t in range(nt):
            for kb in range(8):
                outs.append(f'"=&{{{fmt(mp.SETA(t, kb))}}}"(A[{8 * t + kb}])')
            for kb in range(2):
                ins.append(f'"{{{fmt(mp.XS(t, kb))}}}"(X[{2 * t + kb}])')
    elif kind == "views":
        params = [f"const u32x4 (&V)[{8 * nt}]", f"const u32x4 (&D)[{nt}]", f"u32x4 (&A)[{4 * nt}]", f"u32x4 (&ACCO)[{nt}]"]
        for t in range(nt):
            for kb in range(4):
                outs.append(f'"=&{{{fmt(mp.SETA(t, kb))}}}"(A[{4 * t + kb}])')
            for kb in range(8):
                ins.append(f'"{{{fmt(mp.SETV(t, kb))}}}"(V[{8 * t + kb}])')
            ins.append(f'"{{{fmt(mp.XS(t, 0))}}}"(D[{t}])')
    else:
        params = [f"const u32x4 (&A)[{4 * nt}]", f"u32x4 (&ACCO)[{nt}]"]
        for t in range(nt):
            for kb in range(4):
                ins.append(f'"{{{fmt(mp.SETA(t, kb))}}}"(A[{4 * t + kb}])')
    clobber = list(mp.CLOBBER)
    if acc_out:
        for t in range(nt):
            r = mp.ACC(0, t)
            outs.append(f'"=&{{{fmt(r)}}}"(ACCO[{t}])')
            for i in range(4):
                clobber.remove(r[1] + i)
    for i in range(DEPTH):
        outs.append(f'"+{{{fmt(mp.FR(i))}}}"(F[{i}])')
    outs += ['[islab] "+s"(islab)', '[dsto] "+s"(dsto)', '[keep] "=&s"(keep)']
    ins += ['[rb0] "v"(rb0)', '[rb1] "v"(rb1)', '[rb2] "v"(rb2)', '[rb3] "v"(rb3)', '[bias] "v"(bias)', '[loff] "v"(loff)',
            '[sbase] "s"(sbase)', '[nsl] "s"(nsl)', '[ldsw] "s"(ldsw)']
    clob = ['"memory"', '"scc"', '"vcc"'] + [f'"v{i}"' for i in clobber]
    n_mfma = sum(i.kind == "mfma" for i in e.ins)
    return f"""
// {dt} {nt} tiles {kind}: {len(e.ins)} instructions, {n_mfma} MFMAs, {slabs} slabs; s_nop {sum(i.kind == 'nop' for i in e.ins)}
template <> struct SpecialAsm<{mname}, {nt}, {kid}> {{
  static constexpr int kSlabs = {slabs};
  static __device__ __forceinline__ void run({', '.join(params)}, u32x4 (&F)[4],
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


def gen_layer_x3(in_a, skip, nsb=16, nkb_h=8, act="relu"):
    """One hidden layer of the split-operand kernel (f16x3: x = hi + lo in fp16, three MFMAs per product term), T = 2
    tiles: layer_ob16x3<> + convert_last16x3<> of ns_mlp_engine.h as one statement.  A sub-block is 2 nkb chunks: chunk
    2 kc is W_hi of K-block kc (MFMAs on x_hi and x_lo of both tiles), chunk 2 kc + 1 is W_lo (x_hi only).  Same register
    map as the 16-bit layers with (hi, lo) tuple pairs: set element [tile][kb][half] = base + 64 tile + 8 kb + 4 half;
    accumulators v[0:15], conversion scratch v[16:31].  A finished dword pair costs SEVEN VALU instructions (round 4; ten
    before): the NaN-keeping ReLU by compare + select (4), fp16 hi of both (v_cvt_pk_f16_f32), and the fp16 remainders
    lo = fp16(x - hi) as ONE mixed-precision fma each -- v_fma_mixlo_f16 / v_fma_mixhi_f16 compute fma(f32(hi), -1, x) in fp32
    (exact: hi is the nearest fp16) and round it to fp16 into one half of the destination, the same value the
    convert-back / subtract / convert chain produced; nine with the AGPR writes.  act = "leaky" (the DepthNet): max(x, 0.01 x)
    on the fp32 values instead of the ReLU (same count).  They are queued per sub-block and issued one or two per MFMA gap of
    the following sub-block."""
    TX = 2
    nkb = nkb_h + (2 if skip else 0)
    cps = 2 * nkb
    total = nsb * cps
    assert total % SLAB == 0
    slabs = total // SLAB

    def ACCX(par, t): return R('v', 8 * par + 4 * t, 4)
    def SETX(f, base, t, kb, half): return R(f, base + 64 * t + 8 * kb + 4 * half, 4)
    def IN(t, kb, half): return SETX('a', 0, t, kb, half) if in_a else SETX('v', 128, t, kb, half)
    def OUT(t, kb, half): return SETX('v', 128, t, kb, half) if in_a else SETX('a', 0, t, kb, half)
    def XSX(t, kb, half): return R('v', 64 + 16 * t + 8 * kb + 4 * half, 4)

    e = Emitter("f16")
    e.salu("s_mov_b32 %[keep], m0")
    e.lds_read(BIAS, "%[bias]", 0)

    def dma_setup():
        e.valu(f"v_lshl_add_u32 {fmt(VOFF)}, %[islab], 14, %[loff]", (), (VOFF,))
        e.salu("s_add_u32 m0, %[dsto], %[ldsw]")

    dma_setup()
    e.nop(VALU_WRITE_TO_XDL)

    def piece_ops(s, pi):
        """micro-ops (closures) converting dword pair J of tile t of finished sub-block s;  pi = J * TX + t"""
        t, J = pi % TX, pi // TX
        acc = ACCX(s & 1, t)
        a, b = R('v', acc[1] + 2 * J), R('v', acc[1] + 2 * J + 1)
        ba, bb, ht, lt = (R('v', 16 + 4 * pi + i) for i in range(4))
        ohi, olo = OUT(t, s >> 1, 0), OUT(t, s >> 1, 1)
        dw = 2 * (s & 1) + J
        dhi, dlo = R(ohi[0], ohi[1] + dw), R(olo[0], olo[1] + dw)
        H, L = (dhi, dlo) if in_a else (ht, lt)
        ops = []
        if act == "leaky":               # max(x, 0.01 x) on the fp32 values (NaN stays NaN), as convert_piece16x3<kLeaky>
            for x, tmp in ((a, ba), (b, bb)):
                ops.append(lambda x=x, tmp=tmp: e.valu(f"v_mul_f32_e32 {fmt(tmp)}, 0x3c23d70a, {fmt(x)}", (x,), (tmp,)))
                ops.append(lambda x=x, tmp=tmp: e.valu(f"v_max_f32_e32 {fmt(x)}, {fmt(x)}, {fmt(tmp)}", (x, tmp), (x,)))
        else:
            for x in (a, b):             # x < 0 ? 0 : x, NaN stays NaN (as torch.relu)
                ops.append(lambda x=x: e.valu(f"v_cmp_ngt_f32_e32 vcc, 0, {fmt(x)}", (x,), ()))
                ops.append(lambda x=x: e.valu(f"v_cndmask_b32_e32 {fmt(x)}, 0, {fmt(x)}, vcc", (x,), (x,)))
        ops.append(lambda: e.valu(f"v_cvt_pk_f16_f32 {fmt(H)}, {fmt(a)}, {fmt(b)}", (a, b), (H,)))
        # lo = fp16(x - hi): one mixed-precision fma per element (src0 = the fp16 hi, low / high half; src1, src2 fp32);
        # each writes one half of L and keeps the other, so L counts as read too
        ops.append(lambda: e.valu(f"v_fma_mixlo_f16 {fmt(L)}, {fmt(H)}, -1.0, {fmt(a)} op_sel_hi:[1,0,0]", (H, a, L), (L,)))
        ops.append(lambda: e.valu(f"v_fma_mixhi_f16 {fmt(L)}, {fmt(H)}, -1.0, {fmt(b)} op_sel:[1,0,0] op_sel_hi:[1,0,0]", (H, b, L), (L,)))
        if not in_a:
            ops.append(lambda: e.valu(f"v_accvgpr_write_b32 {fmt(dhi)}, {fmt(ht)}", (ht,), (dhi,)))
            ops.append(lambda: e.valu(f"v_accvgpr_write_b32 {fmt(dlo)}, {fmt(lt)}", (lt,), (dlo,)))
        return ops

    conv_q = []
    pending_salu = []

    def conv(n):
        for _ in range(n):
            if conv_q and not OPT.no_conv:
                conv_q.pop(0)()

    for p in range(total):
        sb, cc = divmod(p, cps)
        kc, part = divmod(cc, 2)
        par = sb & 1
        c = p % SLAB
        frag = FR(p)

        def operand(t, half):
            if skip:
                return XSX(t, kc, half) if kc < 2 else IN(t, kc - 2, half)
            return IN(t, kc, half)

        first = [True] * TX

        def mm(t, half):
            cin = BIAS if (cc == 0 and first[t]) else ACCX(par, t)
            first[t] = False
            e.mfma(ACCX(par, t), frag, operand(t, half), cin)

        def g_read():
            q = p + DEPTH - 1
            e.lds_read(FR(q), f"%[rb{(q // SLAB) % RING}]", (q % SLAB) * 1024)

        def g_misc():
            nonlocal pending_salu
            if c in OPT.dma_steps:
                e.dma(OPT.dma_steps.index(c) * 1024)
                if c == OPT.dma_steps[-1]:
                    pending_salu = ["s_add_i32 %[islab], %[islab], 1", "s_cmp_lg_u32 %[islab], %[nsl]",
                                    "s_cselect_b32 %[islab], %[islab], 0", "s_add_i32 %[dsto], %[dsto], 0x4000",
                                    "s_and_b32 %[dsto], %[dsto], 0xc000"]
            elif pending_salu:
                e.salu(pending_salu.pop(0))
            elif c == SLAB - 1 and p + 1 < total:
                dma_setup()
            if cc == min(6, cps - 1) and sb + 1 < nsb:
                e.lds_read(BIAS, "%[bias]", 64 * (sb + 1))

        ncv = 0 if cc == 0 else 1        # the first chunk of a sub-block leaves the finished accumulators their 8 wait states
        if part == 0:                    # W_hi: x_hi and x_lo of both tiles (per accumulator: hi before lo, as the compiled layer)
            mm(0, 0)
            if c == 0 and not OPT.no_barrier:
                e.salu(f"s_waitcnt vmcnt({VM_WAIT})")
                e.salu("s_barrier")
            conv(ncv)
            mm(1, 0)
            g_read()
            conv(ncv)
            mm(0, 1)
            conv(2 * ncv)
            mm(1, 1)
            conv(ncv)
            g_misc()
        else:                            # W_lo: x_hi only
            mm(0, 0)
            conv(ncv)
            mm(1, 0)
            g_read()
            conv(ncv)
            g_misc()
        if cc == cps - 1 and sb + 1 < nsb:
            assert not conv_q or OPT.no_conv, "conversion queue did not drain within one sub-block"
            for pi in range(2 * TX):
                conv_q.extend(piece_ops(sb, pi))
    assert not pending_salu
    if not OPT.no_conv:
        for pi in range(2 * TX):         # tail: the last sub-block, the pieces' chains interleaved
            conv_q.extend(piece_ops(nsb - 1, pi))
        chains = [conv_q[i * (len(conv_q) // (2 * TX)):(i + 1) * (len(conv_q) // (2 * TX))] for i in range(2 * TX)]
        # (vcc couples a compare to its select: pairs stay together, chains advance two ops at a time for the ReLU part)
        order = []
        for k in range(0, len(chains[0]), 2):
            for ch in chains:
                order.extend(ch[k:k + 2])
        for op in order:
            op()
    e.drain_lds()
    e.salu("s_mov_b32 m0, %[keep]")
    e.nop(VALU_WRITE_TO_XDL)
    if not (OPT.no_wait or OPT.no_lds):
        check(e.ins)
    return e, slabs


def cpp_function(name, dt, in_a, skip, e, slabs, mp=None, leaky=False):
    mp = mp or Map4
    nt = mp.T
    m = {"bf16": "Mma16BF16", "f16": "Mma16F16", "f16x3": "Mma16F16x3"}[dt]
    text = "\\n\\t\"\n      \"".join(i.text for i in e.ins)
    outs, ins = [], []
    for t in range(nt):
        for kb in range(8):
            k = 8 * t + kb
            if in_a:
                outs.append(f'"=&{{{fmt(mp.SETV(t, kb))}}}"(V[{k}])')   # early clobber: written while inputs are still read
                ins.append(f'"{{{fmt(mp.SETA(t, kb))}}}"(A[{k}])')
            else:
                outs.append(f'"=&{{{fmt(mp.SETA(t, kb))}}}"(A[{k}])')
                ins.append(f'"{{{fmt(mp.SETV(t, kb))}}}"(V[{k}])')
    for i in range(DEPTH):
        outs.append(f'"+{{{fmt(mp.FR(i))}}}"(F[{i}])')
    outs += ['[islab] "+s"(islab)', '[dsto] "+s"(dsto)', '[keep] "=&s"(keep)']
    if skip:
        for t in range(nt):
            for kb in range(2):
                ins.append(f'"{{{fmt(mp.XS(t, kb))}}}"(X[{2 * t + kb}])')
    ins += ['[rb0] "v"(rb0)', '[rb1] "v"(rb1)', '[rb2] "v"(rb2)', '[rb3] "v"(rb3)', '[bias] "v"(bias)', '[loff] "v"(loff)',
            '[sbase] "s"(sbase)', '[nsl] "s"(nsl)', '[ldsw] "s"(ldsw)']
    clob = ['"memory"', '"scc"', '"vcc"'] + [f'"v{i}"' for i in mp.CLOBBER]
    n_mfma = sum(i.kind == "mfma" for i in e.ins)
    cyc = issue_cycles(e.ins)
    if leaky:
        assert not skip and nt == 4
        ins.append('[slope] "s"(slope)')
        return f"""
// {name}: {len(e.ins)} instructions, {n_mfma} MFMAs, {slabs} slabs; issue-port estimate {cyc} cycles = {cyc / slabs:.0f} per slab
// (matrix pipe: {16 * n_mfma / slabs:.0f}); s_nop {sum(i.kind == 'nop' for i in e.ins)}, s_waitcnt {sum(i.kind == 'wait' for i in e.ins)}
template <> struct HiddenLeakyAsm<{m}, {nt}, {'true' if in_a else 'false'}> {{
  static constexpr int kSlabs = {slabs};
  static __device__ __forceinline__ void run(u32x4 (&A)[{8 * nt}], u32x4 (&V)[{8 * nt}], u32x4 (&F)[4],
                                             uint32_t rb0, uint32_t rb1, uint32_t rb2, uint32_t rb3, uint32_t bias, uint32_t loff,
                                             uint64_t sbase, uint32_t nsl, uint32_t ldsw, uint32_t& islab, uint32_t& dsto, uint32_t slope) {{
    uint32_t keep;
    asm volatile(
      "{text}"
      : {', '.join(outs)}
      : {', '.join(ins)}
      : {', '.join(clob)});
  }}
}};
"""
    return f"""
// {name}: {len(e.ins)} instructions, {n_mfma} MFMAs, {slabs} slabs; issue-port estimate {cyc} cycles = {cyc / slabs:.0f} per slab
// (matrix pipe: {16 * n_mfma / slabs:.0f}); s_nop {sum(i.kind == 'nop' for i in e.ins)}, s_waitcnt {sum(i.kind == 'wait' for i in e.ins)}
template <> struct HiddenAsm<{m}, {nt}, {'true' if in_a else 'false'}, {'true' if skip else 'false'}> {{
  static constexpr int kSlabs = {slabs};
  static __device__ __forceinline__ void run(u32x4 (&A)[{8 * nt}], u32x4 (&V)[{8 * nt}], const u32x4 (&X)[{2 * nt}], u32x4 (&F)[4],
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
struct Mma16BF16;
struct Mma16F16;
struct Mma16F16x3;
template <class M, int NT, bool IN_A, bool SKIP> struct HiddenAsm;   // NT: tiles of 8 K-block tuples per activation set
template <class M, int NT, int KIND> struct SpecialAsm;             // KIND: 0 layer 0, 1 view layer, 2 rgb head
template <class M, int NT, bool IN_A> struct HiddenLeakyAsm;        // the DepthNet's hidden layers: LeakyReLU(0.01), fp16
"""


FOOTER = """
// Hands the weight ring's bookkeeping to a generated layer statement and takes it back.  Same chunk walk, ring protocol
// and arithmetic as the compiled layer_ob16<> / layer_ob16x3<> (+ convert_last16*<>), so results are bit-identical.
struct AsmRingArgs {
  u32x4 F[4];
  uint32_t rb0, rb1, rb2, rb3, bias, loff, nsl, ldsw, islab, dsto;
  uint64_t sbase;
};
template <class PipeT>
__device__ __forceinline__ void asm_ring_begin(PipeT& ring, const float* bias_lds, int g, AsmRingArgs& r) {
  static_assert(PipeT::RING == 4 && PipeT::LPW == 4 && PipeT::kDepth == 4 && kSlabChunks == 16,
                "the generated streams assume the default ring");
  static_for<4>([&](auto i_) { r.F[decltype(i_)::value] = __builtin_bit_cast(u32x4, ring.f[decltype(i_)::value]); });
  const uint32_t lane16 = ring.lds_off + static_cast<uint32_t>(ring.lane) * 16u;
  r.rb0 = lane16 + ((ring.read_slot + 0) & 3) * kSlabBytes; r.rb1 = lane16 + ((ring.read_slot + 1) & 3) * kSlabBytes;
  r.rb2 = lane16 + ((ring.read_slot + 2) & 3) * kSlabBytes; r.rb3 = lane16 + ((ring.read_slot + 3) & 3) * kSlabBytes;
  r.bias = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NS_LDS_PTR(bias_lds))) + 16u * static_cast<uint32_t>(g);
  r.loff = static_cast<uint32_t>(ring.lane) * 16u;
  const uint64_t base = reinterpret_cast<uint64_t>(ring.stream) + static_cast<uint64_t>(ring.wave) * (4 * kChunkBytes);
  const uint32_t blo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base));
  const uint32_t bhi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(base >> 32));
  r.sbase = (static_cast<uint64_t>(bhi) << 32) | blo;
  r.ldsw = __builtin_amdgcn_readfirstlane(ring.lds_off + static_cast<uint32_t>(ring.wave) * (4 * kChunkBytes));
  r.nsl = __builtin_amdgcn_readfirstlane(ring.n_slabs);
  r.islab = __builtin_amdgcn_readfirstlane(ring.issue_slab);
  r.dsto = __builtin_amdgcn_readfirstlane(ring.issue_slot * kSlabBytes);
}
template <class AFrag, class PipeT>
__device__ __forceinline__ void asm_ring_end(PipeT& ring, const AsmRingArgs& r, int slabs) {
  ring.issue_slab = r.islab;
  ring.issue_slot = r.dsto / kSlabBytes;
  ring.read_slot = (ring.read_slot + slabs) & 3;
  ring.nxt = ring.lds_off + ring.read_slot * kSlabBytes + static_cast<uint32_t>(ring.lane) * 16u;
  ring.cur = ring.nxt;
  static_for<4>([&](auto i_) { ring.f[decltype(i_)::value] = __builtin_bit_cast(AFrag, r.F[decltype(i_)::value]); });
}

// one hidden layer: set A (a[..]) -> set V (v[..]) or back; X = the skip layer's embedded point
template <class M, int NT, bool IN_A, bool SKIP, class PipeT>
__device__ __forceinline__ void hidden_asm_run(PipeT& ring, const float* bias_lds, int g, u32x4 (&A)[8 * NT], u32x4 (&V)[8 * NT], const u32x4 (&X)[2 * NT]) {
  using Gen = HiddenAsm<M, NT, IN_A, SKIP>;
  AsmRingArgs r;
  asm_ring_begin(ring, bias_lds, g, r);
  Gen::run(A, V, X, r.F, r.rb0, r.rb1, r.rb2, r.rb3, r.bias, r.loff, r.sbase, r.nsl, r.ldsw, r.islab, r.dsto);
  asm_ring_end<typename M::AFrag>(ring, r, Gen::kSlabs);
}
// one hidden layer of the DepthNet (LeakyReLU(0.01) on the packed fp16 values): set A -> set V or back
template <class M, int NT, bool IN_A, class PipeT>
__device__ __forceinline__ void hidden_leaky_asm_run(PipeT& ring, const float* bias_lds, int g, u32x4 (&A)[8 * NT], u32x4 (&V)[8 * NT]) {
  using Gen = HiddenLeakyAsm<M, NT, IN_A>;
  AsmRingArgs r;
  asm_ring_begin(ring, bias_lds, g, r);
  Gen::run(A, V, r.F, r.rb0, r.rb1, r.rb2, r.rb3, r.bias, r.loff, r.sbase, r.nsl, r.ldsw, r.islab, r.dsto, 0x211f211fu);   // 0.01 as packed fp16
  asm_ring_end<typename M::AFrag>(ring, r, Gen::kSlabs);
}
// layer 0: the embedded point X -> set A
template <class M, int NT, class PipeT>
__device__ __forceinline__ void layer0_asm_run(PipeT& ring, const float* bias_lds, int g, const u32x4 (&X)[2 * NT], u32x4 (&A)[8 * NT]) {
  using Gen = SpecialAsm<M, NT, 0>;
  AsmRingArgs r;
  asm_ring_begin(ring, bias_lds, g, r);
  Gen::run(X, A, r.F, r.rb0, r.rb1, r.rb2, r.rb3, r.bias, r.loff, r.sbase, r.nsl, r.ldsw, r.islab, r.dsto);
  asm_ring_end<typename M::AFrag>(ring, r, Gen::kSlabs);
}
// view layer: (set V, embedded direction D) -> K-blocks 0..3 of set A; ACCO = raw accumulators of the last (sigma) sub-block
template <class M, int NT, class PipeT>
__device__ __forceinline__ void views_asm_run(PipeT& ring, const float* bias_lds, int g, const u32x4 (&V)[8 * NT], const u32x4 (&D)[NT],
                                              u32x4 (&A)[4 * NT], u32x4 (&ACCO)[NT]) {
  using Gen = SpecialAsm<M, NT, 1>;
  AsmRingArgs r;
  asm_ring_begin(ring, bias_lds, g, r);
  Gen::run(V, D, A, ACCO, r.F, r.rb0, r.rb1, r.rb2, r.rb3, r.bias, r.loff, r.sbase, r.nsl, r.ldsw, r.islab, r.dsto);
  asm_ring_end<typename M::AFrag>(ring, r, Gen::kSlabs);
}
// rgb head: K-blocks 0..3 of set A -> raw accumulators
template <class M, int NT, class PipeT>
__device__ __forceinline__ void rgb_asm_run(PipeT& ring, const float* bias_lds, int g, const u32x4 (&A)[4 * NT], u32x4 (&ACCO)[NT]) {
  using Gen = SpecialAsm<M, NT, 2>;
  AsmRingArgs r;
  asm_ring_begin(ring, bias_lds, g, r);
  Gen::run(A, ACCO, r.F, r.rb0, r.rb1, r.rb2, r.rb3, r.bias, r.loff, r.sbase, r.nsl, r.ldsw, r.islab, r.dsto);
  asm_ring_end<typename M::AFrag>(ring, r, Gen::kSlabs);
}
}  // namespace nsmlp
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nerf_sampling_amd", "csrc", "ns_ob16_asm.inc"))
    ap.add_argument("--dump", help="write the plain instruction stream of one variant (e.g. bf16_AV) here")
    for k in ("no_wait", "no_dma", "no_conv", "no_lds", "no_barrier"):
        ap.add_argument("--exp-" + k.replace("_", "-"), dest=k, action="store_true")
    ap.add_argument("--dma-steps", default="0,2,4,6")
    ap.add_argument("--no-tiles5", dest="tiles5", action="store_false", help="leave out the five-tile streams")
    ap.add_argument("--queue5", action="store_true", help="five-tile streams from the generic queue layout")
    ap.add_argument("--queue4", action="store_true", help="four-tile streams from the generic queue layout (gen_layer_q)")
    a = ap.parse_args()
    OPT.dma_steps = tuple(int(x) for x in a.dma_steps.split(","))
    assert len(OPT.dma_steps) == 4 and OPT.dma_steps[-1] <= 9
    for k in ("no_wait", "no_dma", "no_conv", "no_lds", "no_barrier"):
        setattr(OPT, k, getattr(a, k))
    out = [HEADER]
    for dt in ("bf16", "f16"):
        for in_a in (True, False):
            for skip in (False, True):
                e, slabs = gen_layer_q(dt, in_a, skip, Map4) if a.queue4 else gen_layer(dt, in_a, skip)
                name = f"{dt} {'A->V' if in_a else 'V->A'}{' skip' if skip else ''}"
                out.append(cpp_function(name, dt, in_a, skip, e, slabs))
                print(f"{name}: {len(e.ins)} instr, issue estimate {issue_cycles(e.ins) / slabs:.0f} cycles/slab, "
                      f"nops {sum(i.kind == 'nop' for i in e.ins)}, waits {sum(i.kind == 'wait' for i in e.ins)}", file=sys.stderr)
                if a.dump and a.dump == f"{dt}_{'AV' if in_a else 'VA'}{'_skip' if skip else ''}":
                    open(a.dump + ".s", "w").write("\n".join(i.text for i in e.ins) + "\n")
    if a.tiles5:
        for dt in ("bf16", "f16"):
            for in_a in (True, False):
                for skip in (False, True):
                    e, slabs = gen_layer_q(dt, in_a, skip, Map5) if a.queue5 else gen_layer(dt, in_a, skip, Map5)
                    name = f"{dt} five tiles {'A->V' if in_a else 'V->A'}{' skip' if skip else ''}"
                    out.append(cpp_function(name, dt, in_a, skip, e, slabs, Map5))
                    print(f"{name}: {len(e.ins)} instr, issue estimate {issue_cycles(e.ins) / slabs:.0f} cycles/slab (matrix pipe 1280)", file=sys.stderr)
    for dt in ("bf16", "f16"):
        for mp in ((Map4, Map5) if a.tiles5 else (Map4,)):
            for kind in ("layer0", "views", "rgb"):
                e, slabs = gen_layer_special(dt, kind, mp)
                out.append(cpp_special(dt, kind, e, slabs, mp))
    for in_a in (True, False):           # the DepthNet's hidden layers (fp16 operands, four tiles)
        e, slabs = gen_layer("f16", in_a, False, act="leaky")
        name = f"f16 LeakyReLU {'A->V' if in_a else 'V->A'}"
        out.append(cpp_function(name, "f16", in_a, False, e, slabs, leaky=True))
        print(f"{name}: {len(e.ins)} instr, issue estimate {issue_cycles(e.ins) / slabs:.0f} cycles/slab (matrix pipe 1024), "
              f"nops {sum(i.kind == 'nop' for i in e.ins)}, waits {sum(i.kind == 'wait' for i in e.ins)}", file=sys.stderr)
    for in_a in (True, False):           # the DepthNet's hidden layers on split operands (the PSNR guard's DepthNet)
        e, slabs = gen_layer_x3(in_a, False, act="leaky")
        name = f"f16x3 LeakyReLU {'A->V' if in_a else 'V->A'}"
        out.append(cpp_function(name, "f16x3", in_a, False, e, slabs, leaky=True))
        print(f"{name}: {len(e.ins)} instr, issue estimate {issue_cycles(e.ins) / slabs:.0f} cycles/slab (matrix pipe 768), "
              f"nops {sum(i.kind == 'nop' for i in e.ins)}, waits {sum(i.kind == 'wait' for i in e.ins)}", file=sys.stderr)
    for in_a in (True, False):
        for skip in (False, True):
            e, slabs = gen_layer_x3(in_a, skip)
            name = f"f16x3 {'A->V' if in_a else 'V->A'}{' skip' if skip else ''}"
            out.append(cpp_function(name, "f16x3", in_a, skip, e, slabs))
            print(f"{name}: {len(e.ins)} instr, issue estimate {issue_cycles(e.ins) / slabs:.0f} cycles/slab (matrix pipe 768), "
                  f"nops {sum(i.kind == 'nop' for i in e.ins)}, waits {sum(i.kind == 'wait' for i in e.ins)}", file=sys.stderr)
            if a.dump and a.dump == f"f16x3_{'AV' if in_a else 'VA'}{'_skip' if skip else ''}":
                open(a.dump + ".s", "w").write("\n".join(i.text for i in e.ins) + "\n")
    out.append(FOOTER)
    with open(a.o, "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main()
