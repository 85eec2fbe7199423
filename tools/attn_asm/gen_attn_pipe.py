#!/usr/bin/env python3
"""Generator of the SOFTWARE-PIPELINED key-row loop of the row-streaming attention kernel (csrc/attention_pipe.hip, gfx950; round 5).

Why: in the round-3/4 loop (gen_attn_loop.py) a wave's key row is a strict chain  LDS reads -> QK^T MFMAs -> 32 v_exp + 16 v_cvt_pk ->
PV MFMAs; the overlap of matrix and VALU work was left to four waves per SIMD, and the measured loop runs at the SUM of its MFMA and
VALU time (146 ns per 32x32 score tile against 60 + 94).  An in-order wave hides VALU work behind its OWN MFMAs only when the two are
independent and interleaved in program order (MI355X_MICROARCH.md: <= 5 single-issue fillers per 32-cycle MFMA gap; a partner wave's
slack is not cover).  So this loop is a three-stage software pipeline inside each wave -- in one "body" (one key row of 32 keys, two
query tiles):

      L  : LDS reads for row j+2 (K fragments, bias fragment of tile 0) [+ the chunk's carry fragment]
      P  : O += V(j-1)^T P(j-1)^T            4 MFMAs, interleaved with the 16 exponentials of tile 0 of row j
      V  : LDS transpose-reads of V(j)      (after the PV MFMAs have read the previous V fragments)
      S  : P(j) = fp16(exp2(S(j)))          in place, then packed into the P registers once the PV MFMAs have read them
      Q  : S(j+1) = K(j+1) Q + bias         4 MFMAs, interleaved with the 16 exponentials of tile 1 of row j

and there is NO overflow test: the kernel fixes the softmax offsets before the loop (exact row maxima from a QK^T-only pass built
from the same bodies, `mode='max'`: stage S becomes 16 v_max3_f32, no V / P stages).

Registers.  The pipeline state lives in fixed VGPRs that the kernel pins with physical-register constraints ("+{v[0:31]}" ...), so
that sub-registers can be addressed here and the state survives the C++ code between two statements (barrier, DMA of the next chunk):
      X0..X3 = v[0:63]    bias / logit sets: the bias fragment tile 0 gathers for key row r (set r % 4) is the accumulator init of
                          tile 1 for row r + 1 (tile 1 sits one query row below) -- a set lives through gather, srcC, accumulate, exp
      Z      = v[64:79]   logits of tile 0
      P      = v[80:95]   packed fp16 weights: tile 0 -> P[0:7], tile 1 -> P[8:15]      (mode 'max': second Z set)
      KF0/1  = v[96:111]  K fragments (two k-steps) of rows j+1 / j+2
      VF     = v[112:119] V^T fragments of row j-1 / j
      T      = v[120:127] transient: LDS addresses, mask values (clobbered, not state)
A statement = the four bodies of one chunk of four key rows (rotation periods 4, 2, 2 -> static register names), skewed: statement c
holds bodies 4c-2 .. 4c+1, i.e. it READS K / bias of chunk c (rows 4c .. 4c+3), V rows 4c-2 .. 4c+1 (two of chunk c-1, two of chunk
c), multiplies QK^T for rows 4c-1 .. 4c+2 and PV for rows 4c-3 .. 4c.  Variants: FILL (c = 0: stages of rows < 0 dropped), STEADY,
DRAIN (c = number of chunks: stages of rows beyond the last dropped).

Every emitted instruction carries its register reads / writes; `check()` replays a statement sequence and enforces
   * LDS returns: a register with a read in flight is neither read nor written before an s_waitcnt that retires it (in-order queue);
   * MFMA result -> VALU / LDS-address read or any overwrite: >= MFMA_GAP instructions later (dependent MFMA srcC is exempt);
   * v_exp result -> next VALU: >= 1 instruction in between;  VALU write -> MFMA operand: >= 2 instructions in between;
   * MFMA srcC / A / B overwritten (VALU, LDS return issue) only >= WAR_GAP instructions after the MFMA issued;
   * DATAFLOW: symbolic execution of FILL, STEADY x n, DRAIN reproduces, row by row, the textbook expression of every PV operand
     (which V row, which K row, which bias address, which Q fragment) -- `python3 gen_attn_pipe.py --check`.
"""
import argparse
import struct

X = [0, 16, 32, 48]
Z, P, KF, VF, T = 64, 80, [96, 104], 112, 120
PAIRS = [(0, 1), (2, 3), (8, 9), (10, 11), (16, 17), (18, 19), (24, 25), (26, 27)]   # accumulator rows (r&3)+8*(r>>2), in dwords
MASK_L2 = -100.0 * 1.4426950408889634
MFMA_GAP = 13       # wait states (4 cycles) between an MFMA and a VALU read of its result: 8 passes + 3, one to spare
WAR_GAP = 4       # MFMA A / B operand -> overwritten
WAR_GAP_C = 8     # MFMA srcC -> overwritten (8-pass: 7 wait states)
KSTRIDE = 2048      # bytes per key row of a K (or V) chunk buffer
ABL = set()         # timing-only experiments (results are wrong): nobias, nokv, noexp, nomfma, nocvt


def f32bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def vr(base, n=None):
    return f"v{base}" if n is None else f"v[{base}:{base + n - 1}]"


class Ins:
    """text + metadata.  reads / writes: lists of ('v', index) or ('op', name) for compiler-allocated operands."""

    def __init__(self, text, kind, reads=(), writes=(), **kw):
        self.text, self.kind, self.reads, self.writes = text, kind, list(reads), list(writes)
        self.__dict__.update(kw)


def regs(base, n):
    return [("v", base + i) for i in range(n)]


class Emitter:
    def __init__(self):
        self.ins = []
        self.mfma_w = {}      # register -> issue time of the MFMA that last wrote it
        self.t = 0            # issue time in wait states (4 cycles): 1 per instruction, s_nop N: N + 1, and an MFMA issues no
        self.last_mfma = -100  # earlier than 8 states after the previous MFMA of the wave (8 passes occupy the matrix pipe)

    def e(self, *a, **kw):
        I = Ins(*a, **kw)
        if I.kind in ("valu", "lds"):
            # MFMA result -> VALU / address read (or overwrite): pad with s_nop where the schedule itself is too short (drain bodies)
            need = 0
            for r in I.reads + I.writes:
                if r in self.mfma_w:
                    need = max(need, MFMA_GAP - (self.t + 1 - self.mfma_w[r]))
            while need > 0:
                k = min(need, 8)
                self.ins.append(Ins(f"s_nop {k - 1}", "nop", n=k - 1))
                self.t += k
                need -= k
        self.t += 1 + (I.n if I.kind == "nop" else 0)
        if I.kind == "mfma":
            self.t = max(self.t, self.last_mfma + 8)
            self.last_mfma = self.t
            for r in I.writes:
                self.mfma_w[r] = self.t
        self.ins.append(I)

    # ---- SALU / misc ----
    def salu(self, text):
        self.e(text, "salu")

    def label(self, text):
        self.e(text, "label")

    def waitcnt(self, n):
        self.e(f"s_waitcnt lgkmcnt({n})", "wait", n=n)

    def nop(self, n):
        self.e(f"s_nop {n}", "nop", n=n)

    # ---- VALU ----
    def valu(self, text, reads, writes, trans=False, sem=None):
        self.e(text, "valu", reads, writes, trans=trans, sem=sem)

    # ---- LDS ----
    def lds(self, text, addr, dst, n, sem):
        tag = sem[0] if not isinstance(sem[0], tuple) else sem[0][0]
        if ("nobias" in ABL and tag in ("bias", "carry")) or ("nokv" in ABL and tag in ("K", "V")):
            return
        self.e(text, "lds", [("v", addr)], regs(dst, n), sem=sem)

    # ---- MFMA ----
    def mfma(self, d, a, b, c, sem):
        """d, c: ('v', base) sets of 16 or ('op', name);  a: ('v', base) 4 regs; b: same or ('op', name)"""
        def txt(x, n):
            return vr(x[1], n) if x[0] == "v" else f"%[{x[1]}]"

        def rr(x, n):
            return regs(x[1], n) if x[0] == "v" else [x]
        if "nomfma" in ABL:
            return
        self.e(f"v_mfma_f32_32x32x16_f16 {txt(d, 16)}, {txt(a, 4)}, {txt(b, 4)}, {txt(c, 16)}", "mfma",
               rr(a, 4) + rr(b, 4) + rr(c, 16), rr(d, 16), a=a, b=b, c=c, d=d, sem=sem)


def body(E, i, var, mode, mask):
    """Body at position i (0..3) of a statement.  var: 'fill' | 'steady' | 'drain'.  mode: 'main' | 'max'.
    Row indices are relative to the statement's chunk c: row(i) names:  L -> rows i (R0..R3), Q -> row i-1, S -> row i-2, V -> row i-2,
    PV -> row i-3.  fill: rows < 0 do not exist; drain: rows >= 0 do not exist."""
    def exists(rel):
        return rel >= 0 if var == "fill" else (rel < 0 if var == "drain" else True)

    do_L, do_Q, do_S, do_PV = exists(i), exists(i - 1), exists(i - 2), exists(i - 3)
    do_V = do_S and mode == "main"
    do_PV = do_PV and mode == "main"
    do_carry = do_L and i == 1        # tile 1's accumulator init for the first row of the chunk comes from the chunk's own table window
    if not (do_L or do_Q or do_S or do_PV):
        return
    xg, xc, xy, xs = X[i % 4], X[(i + 3) % 4], X[(i + 2) % 4], X[(i + 1) % 4]   # gather target, srcC of tile 0, tile-1 accumulate, tile-1 exp
    kf_new, kf_use = KF[i % 2], KF[(i + 1) % 2]
    # 'max' mode: two logit sets for tile 0 (Z, P) alternating by row parity (written by Q for row i-1, read by S for row i-2)
    zq = Z if mode == "main" else (Z if (i + 1) % 2 == 0 else P)
    zs = Z if mode == "main" else (Z if i % 2 == 0 else P)
    ta, tb, tc_, td = T, T + 1, T + 2, T + 3

    def gather(dst, addr, sem):
        for n_, (o0, o1) in enumerate(PAIRS):
            E.lds(f"ds_read2_b32 {vr(dst + 2 * n_, 2)}, {vr(addr)} offset0:{o0} offset1:{o1}", addr, dst + 2 * n_, 2, (sem, n_))

    # ---------------- top: addresses + LDS reads of row i (K, bias) ----------------
    n_top = 0
    if do_L:
        E.valu(f"v_add_u32 {vr(ta)}, %[kpar], %[ka0]", [("op", "kpar"), ("op", "ka0")], regs(ta, 1), sem=("kaddr", 0))
        E.valu(f"v_xor_b32 {vr(tb)}, 32, {vr(ta)}", regs(ta, 1), regs(tb, 1), sem=("kaddr", 1))
        E.valu(f"v_add_u32 {vr(tc_)}, %[sbw], %[bl]", [("op", "sb"), ("op", "bl")], regs(tc_, 1), sem=("baddr", i))
        E.lds(f"ds_read_b128 {vr(kf_new, 4)}, {vr(ta)} offset:{i * KSTRIDE}", ta, kf_new, 4, ("K", i, 0))
        E.lds(f"ds_read_b128 {vr(kf_new + 4, 4)}, {vr(tb)} offset:{i * KSTRIDE}", tb, kf_new + 4, 4, ("K", i, 1))
        gather(xg, tc_, ("bias", i))
        n_top = 10
    pending_prev = True   # reads issued by the previous body (K / bias at its top, V in its middle) must have landed before M1 / M5
    if do_PV or do_Q or do_S:
        E.waitcnt(n_top)
    if do_carry:
        # bias fragment of (tile 0, key row R0 - 1) of THIS chunk's table window = tile 1's accumulator init for row R0
        # (sb already points at row R1 here: two rows up)
        E.valu(f"v_subrev_u32 {vr(td)}, %[d4], {vr(tc_)}", [("op", "d4")] + regs(tc_, 1), regs(td, 1), sem=("caddr",))
        E.valu(f"v_subrev_u32 {vr(td)}, %[d4], {vr(td)}", [("op", "d4")] + regs(td, 1), regs(td, 1), sem=("caddr",))
        gather(xy, td, ("carry",))
    if do_L:
        E.salu("s_add_u32 %[sbw], %[sbw], %[d4]")

    # ---------------- VALU work of stage S, handed out in slices between the MFMAs ----------------
    if mode == "main":
        def exps(S):
            out = []
            for r in range(16):
                op = "v_mov_b32" if "noexp" in ABL else "v_exp_f32"
                out.append(("valu", f"{op} {vr(S + r)}, {vr(S + r)}", regs(S + r, 1), regs(S + r, 1), True, ("exp",)))
            return out

        def cvts(S, dst):
            if "nocvt" in ABL:
                return []
            return [("valu", f"v_cvt_pk_f16_f32 {vr(dst + n_)}, {vr(S + 2 * n_)}, {vr(S + 2 * n_ + 1)}", regs(S + 2 * n_, 2), regs(dst + n_, 1), False, ("cvt",))
                    for n_ in range(8)]
    else:
        def exps(S):   # 'max': running maxima of the two tiles (operands mx0 / mx1), 8 three-input maxima per tile
            return []

        def cvts(S, dst):
            return []

    def maskvals(row_rel):
        """mask value per (tile, 16-key band) of key row S-stage row: -144.27 * |id_band - id_query| into T+4..T+7 (t: 2 scalars)"""
        out = []
        src = "%[ids0]" if row_rel < 0 else "%[ids1]"
        r4 = (row_rel + 4) % 4
        out.append(("salu", f"s_bfe_u32 %[t0], {src}, 0x{(4 << 16) | (8 * r4):x}"))
        out.append(("salu", f"s_bfe_u32 %[t1], {src}, 0x{(4 << 16) | (8 * r4 + 4):x}"))
        ops_ = (("%[t0]", "idq0"), ("%[t1]", "idq0"), ("%[t0]", "idq1"), ("%[t1]", "idq1"))
        for k in range(4):
            out.append(("valu", f"v_sub_u32 {vr(T + 4 + k)}, {ops_[k][0]}, %[{ops_[k][1]}]", [("op", ops_[k][1])], regs(T + 4 + k, 1), False, ("mv", 0)))
        for k in range(4):
            out.append(("valu", f"v_cvt_f32_i32 {vr(T + 4 + k)}, {vr(T + 4 + k)}", regs(T + 4 + k, 1), regs(T + 4 + k, 1), False, ("mv", 1)))
        for k in range(4):
            out.append(("valu", f"v_mul_f32_e64 {vr(T + 4 + k)}, -|{vr(T + 4 + k)}|, %[t2]", regs(T + 4 + k, 1), regs(T + 4 + k, 1), False, ("mv", 2)))
        return out

    def maskadd(S, tile):
        # logits of keys 0..15 (accumulator registers 0..7) get the low band's value, 16..31 (registers 8..15) the high band's
        return [("valu", f"v_add_f32 {vr(S + r)}, {vr(S + r)}, {vr(T + 4 + 2 * tile + (1 if r >= 8 else 0))}",
                 regs(S + r, 1) + regs(T + 4 + 2 * tile + (1 if r >= 8 else 0), 1), regs(S + r, 1), False, ("madd",)) for r in range(16)]

    def maxes(S, tile):
        nm = f"mx{tile}"
        out = []
        for n_ in range(8):
            out.append(("valu", f"v_max3_f32 %[{nm}], %[{nm}], {vr(S + 2 * n_)}, {vr(S + 2 * n_ + 1)}", [("op", nm)] + regs(S + 2 * n_, 2), [("op", nm)], False, ("max",)))
        return out

    def emit(items):
        for it in items:
            if it[0] == "salu":
                E.salu(it[1])
            else:
                E.valu(it[1], it[2], it[3], trans=it[4], sem=it[5])

    def interleave(a, b):
        """exp, exp, then one cvt per exp pair afterwards (a cvt never directly follows the exp it reads)"""
        out, ai, bi = [], 0, 0
        while ai < len(a) or bi < len(b):
            for _ in range(2):
                if ai < len(a):
                    out.append(a[ai]); ai += 1
            if bi < len(b):
                out.append(b[bi]); bi += 1
        return out

    s0, s1 = zs, xs     # logits of row i-2: tile 0, tile 1
    if do_S:
        pre = maskvals(i - 2) if mask else []
        if mode == "main":
            a0 = (maskadd(s0, 0) if mask else []) + exps(s0)
            c0 = cvts(s0, P)
            a1 = (maskadd(s1, 1) if mask else []) + exps(s1)
            c1 = cvts(s1, P + 8)
        else:
            a0 = (maskadd(s0, 0) if mask else []) + maxes(s0, 0)
            a1 = (maskadd(s1, 1) if mask else []) + maxes(s1, 1)
            c0, c1 = [], []
    else:
        pre, a0, c0, a1, c1 = [], [], [], [], []

    def split(lst, n):
        k, m = divmod(len(lst), n)
        out, p = [], 0
        for q in range(n):
            sz = k + (1 if q < m else 0)
            out.append(lst[p:p + sz]); p += sz
        return out

    # ---------------- first half: PV MFMAs of row i-3 beside the exponentials of tile 0 ----------------
    emit(pre)
    if do_PV:
        sl = split(a0, 4)
        pv = [(("op", "o0"), VF, P), (("op", "o1"), VF, P + 8), (("op", "o0"), VF + 4, P + 4), (("op", "o1"), VF + 4, P + 12)]
        for q, (o, vf, pp) in enumerate(pv):
            E.mfma(o, ("v", vf), ("v", pp), o, ("pv", q))
            emit(sl[q])
    else:
        emit(a0)
    if do_V:
        E.valu(f"v_add_u32 {vr(ta)}, %[{'vpar0' if i < 2 else 'vpar1'}], %[va]", [("op", "vpar0" if i < 2 else "vpar1"), ("op", "va")], regs(ta, 1), sem=("vaddr",))
        vrow = (i - 2) % 4
        for n_ in range(4):
            E.lds(f"ds_read_b64_tr_b16 {vr(VF + 2 * n_, 2)}, {vr(ta)} offset:{vrow * KSTRIDE + 512 * n_}", ta, VF + 2 * n_, 2, ("V", i - 2, n_))
    # the tile-0 weights go to the P registers only now (the PV MFMAs above have read the previous row's)
    # ---------------- second half: QK^T MFMAs of row i-1 beside the exponentials of tile 1 ----------------
    def lagged(a, c):
        """items of a; the convert of exp pair p follows pair p + 1 (never directly behind the exponentials it reads)"""
        out, n_exp, ci = [], 0, 0
        for it in a:
            out.append(it)
            if it[0] == "valu" and it[4]:
                n_exp += 1
                if n_exp % 2 == 0 and n_exp // 2 - 2 >= ci and ci < len(c):
                    out.append(c[ci]); ci += 1
        return out + c[ci:]

    t1 = lagged(a1, c1)
    # tile-0 converts first (M5 overwrites Z): two tile-1 items in front of them and between the first ones, so that no convert
    # directly follows the exponential it reads
    head = interleave(t1[:4], c0) if c0 else []
    rest = t1[4:] if c0 else t1
    emit(head)
    if do_Q:
        if do_carry:
            E.waitcnt(4 if do_V else 0)
        qk = [(("v", zq), kf_use, "q00", ("v", xc)), (("v", xy), kf_use, "q10", ("v", xy)),
              (("v", zq), kf_use + 4, "q01", ("v", zq)), (("v", xy), kf_use + 4, "q11", ("v", xy))]
        sl = split(rest, 4)
        for q, (d, kf, qn, c) in enumerate(qk):
            E.mfma(d, ("v", kf), ("op", qn), c, ("qk", q))
            emit(sl[q])
    else:
        emit(rest)


def statement(var, mode, mask):
    E = Emitter()
    E.waitcnt(0)      # (scalar loads the compiler may have in flight share the counter)
    if var != "drain":
        E.salu("s_mov_b32 %[sbw], %[sb]")      # working copy of the bias address (an early-clobber output: no value flows between statements)
    if mask:
        E.salu(f"s_mov_b32 %[t2], 0x{f32bits(-MASK_L2):08x}")    # +144.27, multiplied by -|id_k - id_q|
    for i in range(4):
        body(E, i, var, mode, mask)
    E.waitcnt(0)      # a statement ends with an empty LDS queue (the caller's barrier / DMA follows)
    # ... and with its last MFMA results readable: the next statement may open with VALU work on them (whatever the compiler puts
    # between two statements is not counted on)
    last = max(E.mfma_w.values()) if E.mfma_w else None
    if last is not None:
        need = MFMA_GAP - (E.t + 1 - last)
        while need > 0:
            k = min(need, 8)
            E.nop(k - 1)
            need -= k
    return E.ins


# ------------------------------------------------------------------------------------------------------------------------------
# checker
# ------------------------------------------------------------------------------------------------------------------------------
class CheckError(Exception):
    pass


def check_hazards(ins, name):
    queue = []              # in-flight LDS reads: list of (index, set(regs))
    last_mfma_w = {}        # reg -> index of the MFMA that writes it
    last_mfma_r = {}        # reg -> index of the last MFMA that reads it
    last_valu_w = {}        # reg -> (index, trans)
    n = -1
    last_mfma_t = -100
    for I in ins:
        n += 1
        if I.kind == "nop":
            n += I.n          # s_nop N = N + 1 wait states
        if I.kind == "mfma":
            n = max(n, last_mfma_t + 8)      # the matrix pipe takes the next MFMA of a wave 8 passes after the previous one
            last_mfma_t = n
        if I.kind in ("label", "salu", "nop"):
            continue
        if I.kind == "wait":
            while len(queue) > I.n:
                queue.pop(0)
            continue
        inflight = set().union(*[q[1] for q in queue]) if queue else set()
        for r in I.reads + I.writes:
            if r in inflight:
                raise CheckError(f"{name}: #{n} `{I.text}` touches {r} with an LDS read in flight")
        if I.kind == "mfma":
            for r in I.reads:
                if r in last_valu_w and n - last_valu_w[r][0] <= 2:
                    raise CheckError(f"{name}: #{n} `{I.text}` reads {r} written by VALU #{last_valu_w[r][0]} (< 2 in between)")
                if r in last_mfma_w and n - last_mfma_w[r] < MFMA_GAP:
                    c_regs = set(regs(I.c[1], 16)) if I.c[0] == "v" else {I.c}
                    if r not in c_regs:
                        raise CheckError(f"{name}: #{n} `{I.text}` reads {r} (A/B) from MFMA #{last_mfma_w[r]} too early")
            for r in I.writes:
                if r in last_mfma_w and n - last_mfma_w[r] < MFMA_GAP and r not in I.reads:
                    raise CheckError(f"{name}: #{n} MFMA overwrites {r} of MFMA #{last_mfma_w[r]}")
            c_set = set(regs(I.c[1], 16)) if I.c[0] == "v" else {I.c}
            for r in I.reads:
                last_mfma_r[r] = (n, WAR_GAP_C if r in c_set else WAR_GAP)
            for r in I.writes:
                last_mfma_w[r] = n
            continue
        # VALU or LDS
        for r in I.reads:
            if r in last_mfma_w and n - last_mfma_w[r] < MFMA_GAP:
                raise CheckError(f"{name}: #{n} `{I.text}` reads {r} {n - last_mfma_w[r]} after MFMA #{last_mfma_w[r]}")
            if r in last_valu_w and last_valu_w[r][1] and n - last_valu_w[r][0] < 2:
                raise CheckError(f"{name}: #{n} `{I.text}` reads transcendental result {r} of the previous instruction")
        for r in I.writes:
            if r in last_mfma_w and n - last_mfma_w[r] < MFMA_GAP:
                raise CheckError(f"{name}: #{n} `{I.text}` overwrites {r} {n - last_mfma_w[r]} after MFMA #{last_mfma_w[r]} wrote it")
            if r in last_mfma_r and n - last_mfma_r[r][0] < last_mfma_r[r][1]:
                raise CheckError(f"{name}: #{n} `{I.text}` overwrites {r} {n - last_mfma_r[r][0]} after MFMA #{last_mfma_r[r][0]} read it")
        if I.kind == "valu":
            for r in I.writes:
                last_valu_w[r] = (n, I.trans)
        else:
            queue.append((n, set(I.writes)))
    if queue:
        raise CheckError(f"{name}: statement ends with LDS reads in flight")


def check_dataflow(mode, mask, nch=3):
    """Symbolic execution of FILL, STEADY x (nch-1), DRAIN against the textbook definition.  Values are nested tuples."""
    R = {}          # ('v', n) / ('op', name) -> symbolic value

    def val(r):
        return R.get(r, ("undef", r))

    pv_log = []     # (mfma q, O operand value) in program order
    mx = {"mx0": ("mx0_init",), "mx1": ("mx1_init",)}
    seq = ["fill"] + ["steady"] * (nch - 1) + ["drain"]
    for c, var in enumerate(seq):
        # per-statement scalar operands as symbols carrying the chunk index
        sb = [0]                 # s_add_u32 sb, sb, d4 advances the row
        for I in statement(var, mode, mask):
            t = I.text
            if I.kind == "salu":
                if t.startswith("s_add_u32 %[sbw]"):
                    sb[0] += 1
                continue
            if I.kind in ("wait", "nop", "label"):
                continue
            sem = I.sem
            if I.kind == "valu":
                if sem[0] == "kaddr":
                    R[I.writes[0]] = ("kaddr", c, sem[1])
                elif sem[0] == "baddr":
                    R[I.writes[0]] = ("brow", c, sb[0])
                elif sem[0] == "caddr":
                    src = val(I.reads[1])
                    R[I.writes[0]] = ("brow", src[1], src[2] - 1)
                elif sem[0] == "vaddr":
                    which = "vpar0" if "vpar0" in t else "vpar1"
                    R[I.writes[0]] = ("vaddr", c - 1 if which == "vpar0" else c)
                elif sem[0] == "exp":
                    R[I.writes[0]] = ("exp", val(I.reads[0]))
                elif sem[0] == "cvt":
                    R[I.writes[0]] = ("cvt", val(I.reads[0]), val(I.reads[1]))
                elif sem[0] == "mv":
                    R[I.writes[0]] = ("mv", sem[1], t.split(",")[1].strip() if sem[1] == 0 else val(I.reads[0]), [r for r in I.reads if r[0] == "op"].__repr__() if sem[1] == 0 else "")
                elif sem[0] == "madd":
                    R[I.writes[0]] = ("madd", val(I.reads[0]), val(I.reads[1]))
                elif sem[0] == "max":
                    nm = I.writes[0][1]
                    mx[nm] = ("max3", mx[nm], val(I.reads[1]), val(I.reads[2]))
                else:
                    raise CheckError("unknown VALU sem " + repr(sem))
            elif I.kind == "lds":
                for k, w in enumerate(I.writes):
                    R[w] = ("lds", t.split()[0], val(I.reads[0]), t.split("offset", 1)[1], k)
            elif I.kind == "mfma":
                def vec(x, n_):
                    return tuple(val(("v", x[1] + k)) for k in range(n_)) if x[0] == "v" else ("op", x[1]) if x[1] not in ("o0", "o1") else val(x)
                a, b, cc = vec(I.a, 4), vec(I.b, 4), vec(I.c, 16)
                if I.d[0] == "op":
                    R[I.d] = ("mfma", a, b, cc)
                    pv_log.append((I.d[1], a, b))
                else:
                    for k in range(16):
                        R[("v", I.d[1] + k)] = ("mfma", a, b, cc, k)

    # ---- textbook ----
    def Krow(c, row, step):        # K fragment of chunk c, row (0..3), k-step
        return tuple(("lds", "ds_read_b128", ("kaddr", c, step), f":{row * KSTRIDE}", k) for k in range(4))

    def gather(addr):
        return tuple(("lds", "ds_read2_b32", addr, f"0:{o0} offset1:{o1}", k) for (o0, o1) in PAIRS for k in range(2))

    def bias(c, row):
        return gather(("brow", c, row))

    def carry(c):                  # gathered in body i = 1: the address register holds sb advanced by one row, minus d4
        return gather(("brow", c, -1))

    def Vrow(c, row, n_):
        return tuple(("lds", "ds_read_b64_tr_b16", ("vaddr", c), f":{row * KSTRIDE + 512 * n_}", k) for k in range(2))

    def mfma16(a, b, cc):
        return tuple(("mfma", a, b, cc, k) for k in range(16))

    def mvals(c, row):
        return None

    want = []
    S_prev_bias = None
    for g in range(4 * nch):
        c, row = divmod(g, 4)
        b0 = bias(c, row)
        b1 = carry(c) if row == 0 else bias(c, row - 1)
        k0, k1 = Krow(c, row, 0), Krow(c, row, 1)
        s0 = mfma16(k1, ("op", "q01"), mfma16(k0, ("op", "q00"), b0))
        s1 = mfma16(k1, ("op", "q11"), mfma16(k0, ("op", "q10"), b1))
        if mode == "max":
            want.append((s0, s1))
            continue
        if mask:
            want.append(None)
            continue
        p0 = tuple(("cvt", ("exp", s0[2 * n_]), ("exp", s0[2 * n_ + 1])) for n_ in range(8))
        p1 = tuple(("cvt", ("exp", s1[2 * n_]), ("exp", s1[2 * n_ + 1])) for n_ in range(8))
        va = Vrow(c, row, 0) + Vrow(c, row, 1)
        vb = Vrow(c, row, 2) + Vrow(c, row, 3)
        want += [("o0", va, p0[0:4]), ("o1", va, p1[0:4]), ("o0", vb, p0[4:8]), ("o1", vb, p1[4:8])]
    if mode == "max":
        # every logit register of every row must have entered exactly one max3 of the right accumulator
        def collect(v, acc):
            while v[0] == "max3":
                acc += [v[2], v[3]]
                v = v[1]
            return acc
        for tile, nm in enumerate(("mx0", "mx1")):
            got = collect(mx[nm], [])
            exp_ = [s[tile][k] for s in want for k in range(16)]
            if mask:
                if len(got) != len(exp_):
                    raise CheckError(f"max/{mask}: {nm} saw {len(got)} values, expected {len(exp_)}")
                continue
            if sorted(map(repr, got)) != sorted(map(repr, exp_)):
                raise CheckError(f"max: {nm} operands differ from the textbook logits")
        return
    if mask:
        if len(pv_log) != len(want) * 4:
            raise CheckError(f"main/mask: {len(pv_log)} PV MFMAs, expected {4 * len(want)}")
        return
    if len(pv_log) != len(want):
        raise CheckError(f"main: {len(pv_log)} PV MFMAs, expected {len(want)}")
    for n, (got, exp_) in enumerate(zip(pv_log, want)):
        if got != exp_:
            raise CheckError(f"main: PV MFMA #{n} (row {n // 4}) differs:\n got  {repr(got)[:600]}\n want {repr(exp_)[:600]}")



# ------------------------------------------------------------------------------------------------------------------------------
# first pass: row maxima of one chunk, batched (stateless between chunks)
# ------------------------------------------------------------------------------------------------------------------------------
MB_B = [0, 16, 32, 48, 64]          # bias fragments: carry (row -1), rows 0..3.  Tile 1 accumulates row r in place on fragment r-1
MB_K = [80, 88, 96, 104]            # K fragments of the four rows
MB_Z = [112, 128]                   # logits of tile 0, alternating by row
MB_T = 144                          # addresses, mask values
MB_NREG = 152


def max_batched(mask):
    """The pipelined bodies are too short in 'max' mode to cover the LDS latency (measured: 790 ticks per row, as slow as the main loop).
    Here all LDS reads of the chunk (4 x 2 K fragments, 5 bias fragments: 48 instructions) are issued up front, the 16 MFMAs follow
    back to back and the 16 three-input maxima of row r sit between the MFMAs of row r + 1.  Nothing lives in registers between two
    chunks except the running maxima (operands mx0 / mx1)."""
    E = Emitter()
    E.waitcnt(0)
    E.salu("s_mov_b32 %[sbw], %[sb]")
    if mask:
        E.salu(f"s_mov_b32 %[t2], 0x{f32bits(-MASK_L2):08x}")
    ta, tb, tc_ = MB_T, MB_T + 1, MB_T + 2
    E.valu(f"v_add_u32 {vr(ta)}, %[kpar], %[ka0]", [("op", "kpar"), ("op", "ka0")], regs(ta, 1), sem=("kaddr", 0))
    E.valu(f"v_xor_b32 {vr(tb)}, 32, {vr(ta)}", regs(ta, 1), regs(tb, 1), sem=("kaddr", 1))
    E.valu(f"v_add_u32 {vr(tc_)}, %[sbw], %[bl]", [("op", "sb"), ("op", "bl")], regs(tc_, 1), sem=("baddr", 0))
    E.valu(f"v_subrev_u32 {vr(tc_)}, %[d4], {vr(tc_)}", [("op", "d4")] + regs(tc_, 1), regs(tc_, 1), sem=("caddr",))      # row -1

    def gather(dst, addr, sem):
        for n_, (o0, o1) in enumerate(PAIRS):
            E.lds(f"ds_read2_b32 {vr(dst + 2 * n_, 2)}, {vr(addr)} offset0:{o0} offset1:{o1}", addr, dst + 2 * n_, 2, (sem, n_))

    issued = []     # rows in issue order: ('K', r) 2 reads, ('B', r) 8 reads
    gather(MB_B[0], tc_, ("carry",))
    issued.append(8)
    for r in range(4):
        E.lds(f"ds_read_b128 {vr(MB_K[r], 4)}, {vr(ta)} offset:{r * KSTRIDE}", ta, MB_K[r], 4, ("K", r, 0))
        E.lds(f"ds_read_b128 {vr(MB_K[r] + 4, 4)}, {vr(tb)} offset:{r * KSTRIDE}", tb, MB_K[r] + 4, 4, ("K", r, 1))
        E.valu(f"v_add_u32 {vr(tc_)}, %[d4], {vr(tc_)}", [("op", "d4")] + regs(tc_, 1), regs(tc_, 1), sem=("baddr+", r))
        gather(MB_B[r + 1], tc_, ("bias", r))
        issued.append(10)
    total = sum(issued)

    def maskvals(r):
        out = []
        out.append(("salu", f"s_bfe_u32 %[t0], %[ids1], 0x{(4 << 16) | (8 * r):x}"))
        out.append(("salu", f"s_bfe_u32 %[t1], %[ids1], 0x{(4 << 16) | (8 * r + 4):x}"))
        ops_ = (("%[t0]", "idq0"), ("%[t1]", "idq0"), ("%[t0]", "idq1"), ("%[t1]", "idq1"))
        for k in range(4):
            out.append(("valu", f"v_sub_u32 {vr(MB_T + 4 + k)}, {ops_[k][0]}, %[{ops_[k][1]}]", [("op", ops_[k][1])], regs(MB_T + 4 + k, 1), False, ("mv", 0)))
        for k in range(4):
            out.append(("valu", f"v_cvt_f32_i32 {vr(MB_T + 4 + k)}, {vr(MB_T + 4 + k)}", regs(MB_T + 4 + k, 1), regs(MB_T + 4 + k, 1), False, ("mv", 1)))
        for k in range(4):
            out.append(("valu", f"v_mul_f32_e64 {vr(MB_T + 4 + k)}, -|{vr(MB_T + 4 + k)}|, %[t2]", regs(MB_T + 4 + k, 1), regs(MB_T + 4 + k, 1), False, ("mv", 2)))
        return out

    def tail_of(r):
        """VALU work on the logits of row r: [mask] + 16 maxima"""
        z, y = MB_Z[r % 2], MB_B[r]
        out = []
        if mask:
            out += maskvals(r)
            for tile, S in enumerate((z, y)):
                out += [("valu", f"v_add_f32 {vr(S + q)}, {vr(S + q)}, {vr(MB_T + 4 + 2 * tile + (1 if q >= 8 else 0))}",
                         regs(S + q, 1) + regs(MB_T + 4 + 2 * tile + (1 if q >= 8 else 0), 1), regs(S + q, 1), False, ("madd",)) for q in range(16)]
        for tile, S in enumerate((z, y)):
            nm = f"mx{tile}"
            for n_ in range(8):
                out.append(("valu", f"v_max3_f32 %[{nm}], %[{nm}], {vr(S + 2 * n_)}, {vr(S + 2 * n_ + 1)}", [("op", nm)] + regs(S + 2 * n_, 2), [("op", nm)], False, ("max",)))
        return out

    def emit(items):
        for it in items:
            if it[0] == "salu":
                E.salu(it[1])
            else:
                E.valu(it[1], it[2], it[3], trans=it[4], sem=it[5])

    def split(lst, n):
        k, m = divmod(len(lst), n)
        out, p_ = [], 0
        for q in range(n):
            sz = k + (1 if q < m else 0)
            out.append(lst[p_:p_ + sz]); p_ += sz
        return out

    done = 8
    for r in range(4):
        done += 10
        E.waitcnt(min(15, total - done))        # carry / bias / K of rows <= r have landed (in-order returns)
        z, y = MB_Z[r % 2], MB_B[r]
        prev = split(tail_of(r - 1), 4) if r > 0 else [[], [], [], []]
        qk = [(("v", z), MB_K[r], "q00", ("v", MB_B[r + 1])), (("v", y), MB_K[r], "q10", ("v", y)),
              (("v", z), MB_K[r] + 4, "q01", ("v", z)), (("v", y), MB_K[r] + 4, "q11", ("v", y))]
        for q, (d, kf, qn, c) in enumerate(qk):
            E.mfma(d, ("v", kf), ("op", qn), c, ("qk", q))
            emit(prev[q])
    emit(tail_of(3))       # (the emitter pads with s_nop up to the MFMA result latency)
    E.waitcnt(0)
    return E.ins


def check_max_batched(mask):
    ins = max_batched(mask)
    check_hazards(ins, f"maxb/mask{mask}")
    check_hazards(ins + ins, f"maxb+maxb/mask{mask}")
    # dataflow: every logit register of the textbook enters exactly one max3 of the right accumulator
    R = {}
    mx = {"mx0": [], "mx1": []}
    brow = [None]

    def val(r):
        return R.get(r, ("undef", r))
    for I in ins:
        if I.kind in ("salu", "wait", "nop", "label"):
            continue
        sem = I.sem
        if I.kind == "valu":
            if sem[0] == "kaddr":
                R[I.writes[0]] = ("kaddr", 0, sem[1])
            elif sem[0] == "baddr":
                R[I.writes[0]] = ("brow", 0, 0)
            elif sem[0] == "caddr":
                src = val(I.reads[1]); R[I.writes[0]] = ("brow", 0, src[2] - 1)
            elif sem[0] == "baddr+":
                src = val(I.reads[1]); R[I.writes[0]] = ("brow", 0, src[2] + 1)
            elif sem[0] == "max":
                mx[I.writes[0][1]] += [val(I.reads[1]), val(I.reads[2])]
            elif sem[0] == "madd":
                R[I.writes[0]] = ("madd", val(I.reads[0]))
            elif sem[0] == "mv":
                R[I.writes[0]] = ("mv",)
        elif I.kind == "lds":
            for k, w in enumerate(I.writes):
                R[w] = ("lds", I.text.split()[0], val(I.reads[0]), I.text.split("offset", 1)[1], k)
        elif I.kind == "mfma":
            def vec(x, n_):
                return tuple(val(("v", x[1] + k)) for k in range(n_)) if x[0] == "v" else ("op", x[1])
            a, b, cc = vec(I.a, 4), vec(I.b, 4), vec(I.c, 16)
            for k in range(16):
                R[("v", I.d[1] + k)] = ("mfma", a, b, cc, k)

    def Krow(row, step):
        return tuple(("lds", "ds_read_b128", ("kaddr", 0, step), f":{row * KSTRIDE}", k) for k in range(4))

    def gat(row):
        return tuple(("lds", "ds_read2_b32", ("brow", 0, row), f"0:{o0} offset1:{o1}", k) for (o0, o1) in PAIRS for k in range(2))

    def mfma16(a, b, cc):
        return tuple(("mfma", a, b, cc, k) for k in range(16))
    for tile, nm in enumerate(("mx0", "mx1")):
        exp_ = []
        for row in range(4):
            b_ = gat(row) if tile == 0 else gat(row - 1)
            qa, qb = ("q00", "q01") if tile == 0 else ("q10", "q11")
            s_ = mfma16(Krow(row, 1), ("op", qb), mfma16(Krow(row, 0), ("op", qa), b_))
            exp_ += [("madd", x) for x in s_] if mask else list(s_)
        if sorted(map(repr, mx[nm])) != sorted(map(repr, exp_)):
            raise CheckError(f"maxb/mask{mask}: {nm} operands differ from the textbook logits")


def check_all():
    for mask in (0, 1):
        check_max_batched(mask)
    for mode in ("main",):
        for mask in (0, 1):
            for var in ("fill", "steady", "drain"):
                ins = statement(var, mode, mask)
                # hazards across statement boundaries: steady follows steady (worst case) -- check the concatenation too
                check_hazards(ins, f"{mode}/{var}/mask{mask}")
            for a, b in (("fill", "steady"), ("steady", "steady"), ("steady", "drain"), ("fill", "drain")):
                check_hazards(statement(a, mode, mask) + statement(b, mode, mask), f"{mode}/{a}+{b}/mask{mask}")
            check_dataflow(mode, mask)
    return True


def stats(ins):
    k = {}
    for I in ins:
        k[I.kind] = k.get(I.kind, 0) + 1
    return k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--dump", default="", help="print one statement: e.g. main/steady/0")
    ap.add_argument("--abl", default="", help="timing experiments: comma list of nobias,nokv,noexp,nomfma,nocvt (results are wrong; no checks)")
    a = ap.parse_args()
    ABL.update(x for x in a.abl.split(",") if x)
    if a.dump:
        mode, var, mask = a.dump.split("/")
        for n, I in enumerate(statement(var, mode, int(mask))):
            print(f"{n:4d}  {I.text}")
        return
    if not ABL:
        check_all()
    if a.check:
        for mode in ("main",):
            for var in ("fill", "steady", "drain"):
                print(mode, var, stats(statement(var, mode, 0)), "| mask:", stats(statement(var, mode, 1)))
        print("max batched", stats(max_batched(0)), "| mask:", stats(max_batched(1)))
        print("gen_attn_pipe: hazard + dataflow checks passed")
    if a.out:
        with open(a.out, "w") as f:
            f.write("// generated by tools/attn_asm/gen_attn_pipe.py -- do not edit (regenerate: python3 tools/attn_asm/gen_attn_pipe.py --out <this file>)\n")
            for mode in ("main",):
                for var in ("fill", "steady", "drain"):
                    for mask in (0, 1):
                        f.write(f"#define ATTN_PIPE_{mode.upper()}_{var.upper()}_MASK{mask} \\\n")
                        f.write(" \\\n".join('    "' + I.text + '\\n"' for I in statement(var, mode, mask)) + "\n\n")
            for mask in (0, 1):
                f.write(f"#define ATTN_PIPE_MAXB_MASK{mask} \\\n")
                f.write(" \\\n".join('    "' + I.text + '\\n"' for I in max_batched(mask)) + "\n\n")
            f.write("#define ATTN_PIPE_MAXB_CLOBBER " + ", ".join(f'"v{i}"' for i in range(MB_NREG)) + ', "vcc", "scc", "memory"\n')
            f.write("#define ATTN_PIPE_CLOBBER " + ", ".join(f'"v{i}"' for i in range(T, T + 8)) + ', "vcc", "scc", "memory"\n')


if __name__ == "__main__":
    main()
