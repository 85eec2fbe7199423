#!/usr/bin/env python3
"""Generator of the hand-scheduled key-row loop of the row-streaming attention kernel (csrc/attention_rows.hip, gfx950).

Emits csrc/attn_rows_asm.inc: one inline-asm string per variant.  A statement processes the ROWS key rows (32 keys each)
of the staged K / V chunk for the two query tiles of a wave, starting at row %[rs]; it leaves early -- before the PV
product of the offending row -- when a packed fp16 weight reaches 2^14 (the caller raises the softmax offsets and
re-enters at that row).  Transient state lives in hard-coded VGPRs v0..v63 that the statement clobbers; persistent state
(O^T accumulators, Q fragments, LDS addresses) are operands, so the compiler keeps them anywhere above.

Register map:
   A  = v[0:15]    bias fragment / logits of tile 1 (A and B swap roles every row: the fragment tile 0 gathers for key
   B  = v[16:31]   row hk is tile 1's fragment for row hk + 1 -- tile 1 sits one query row below)
   Z  = v[32:47]   logits of tile 0.  Weights are packed in place: tile 0 -> Z[0:7], tile 1 -> {A|B}[0:7]
   KF = v[48:55]   K fragments of the row (two k-steps); afterwards scratch (mask values, overflow test)
   VF = v[56:63]   V^T fragments (ds_read_b64_tr_b16)

Hazards handled by hand (nothing inside an asm statement is padded by the compiler):
   MFMA result -> VALU read: >= 12 issue states (s_nop / independent instructions) after the producing MFMA;
   v_exp_f32 result -> VALU read: >= 1 instruction in between;  VALU write -> MFMA operand: >= 2 instructions in between.
"""
import argparse
import struct

A, B, Z, KF, VF = 0, 16, 32, 48, 56
PAIRS = [(0, 1), (2, 3), (8, 9), (10, 11), (16, 17), (18, 19), (24, 25), (26, 27)]   # accumulator rows (r&3)+8*(r>>2)
MASK_L2 = -100.0 * 1.4426950408889634
TRIP = 0x74007400   # both halves 2^14


def v(base, n=None):
    return f"v{base}" if n is None else f"v[{base}:{base + n - 1}]"


def f32bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def gen(rows, mask, kstride=2048, abl=(), d32=False):
    """Two copies of the row loop in one statement: with the overflow test, and -- entered when %[nochk] != 0: the offsets of
    all the wave's queries are within 13.5 of the head's logit bound, nothing can trip -- without it (a per-row skip would be
    a taken branch per row)."""
    L = []
    if mask:
        L.append(f"s_mov_b32 %[t2], 0x{f32bits(-MASK_L2):08x}")    # +144.27: multiplied by -|id_k - id_q|
    L.append("s_cmp_lg_u32 %[nochk], 0")
    L.append("s_cbranch_scc1 200f")
    L += body(rows, mask, kstride, abl, True, 0, d32)
    L.append("s_branch 99f")
    L.append("200:")
    L += body(rows, mask, kstride, abl + ("nocheck",), False, 100, d32)
    if "nocheck" not in abl:
        L.append("s_branch 99f")
        for r in range(rows):
            L.append(f"{90 + r}:")
            L.append("s_waitcnt lgkmcnt(0)")        # the V^T reads of the abandoned row still target VF
            L.append(f"s_mov_b32 %[done], {r}")
            if r + 1 < rows:
                L.append("s_branch 99f")
    L.append("99:")
    return L


def body(rows, mask, kstride, abl, check, lb, d32=False):
    """d32: head_dim 32 fills all 32 slots of the Q / K fragments and of V, so the two things the spare slot 31 carries for
    free elsewhere are done on the VALU: the running offset (-m per query, operands %[nm0] / %[nm1]) is added to the logits
    (folded into the mask value where there is one), and the softmax denominator is summed from the packed fp16 weights
    with v_dot2c_f32_f16 (exactly the values the PV product multiplies) into %[l0] / %[l1] after the row has passed the
    overflow test."""
    L = []
    e = L.append
    lab = lambda n: str(lb + n)

    def gather(dst):
        # the address register is the fragment's own last register: it is read at issue, the data arrives later
        for i, (o0, o1) in enumerate(PAIRS):
            e(f"ds_read2_b32 {v(dst + 2 * i, 2)}, {v(dst + 15)} offset0:{o0} offset1:{o1}")

    # ---- entry: the carry fragment of tile 1 (= tile 0's fragment of the previous key row) goes where row rs expects it
    e("s_bitcmp1_b32 %[rs], 0")
    e(f"s_cbranch_scc1 {lab(80)}f")
    e(f"v_add_u32 {v(B + 15)}, %[sb], %[bl]")
    e(f"v_subrev_u32 {v(B + 15)}, %[d4], {v(B + 15)}")
    gather(B)                              # even start row: carry in B
    e(f"s_branch {lab(81)}f")
    e(f"{lab(80)}:")
    e(f"v_add_u32 {v(A + 15)}, %[sb], %[bl]")
    e(f"v_subrev_u32 {v(A + 15)}, %[d4], {v(A + 15)}")
    gather(A)                              # odd start row: carry in A
    e(f"{lab(81)}:")
    for r in range(1, rows):
        e(f"s_cmp_eq_u32 %[rs], {r}")
        e(f"s_cbranch_scc1 {lab(70 + r)}f")
    for r in range(rows):
        X, Y = (A, B) if r % 2 == 0 else (B, A)
        off = r * kstride
        e(f"{lab(70 + r)}:")
        e(f"v_add_u32 {v(KF)}, %[par], %[ka0]")           # chunk buffer parity
        e(f"v_xor_b32 {v(KF + 4)}, 32, {v(KF)}")          # k-step 1: 16-B segment (2 + half) ^ sw = segment of k-step 0 ^ 2
        e(f"ds_read_b128 {v(KF, 4)}, {v(KF)} offset:{off}")
        e(f"ds_read_b128 {v(KF + 4, 4)}, {v(KF + 4)} offset:{off}")
        e(f"v_add_u32 {v(X + 15)}, %[sb], %[bl]")
        if "nobias" not in abl or r == 0:
            gather(X)
        e(f"v_add_u32 {v(VF + 6)}, %[par], %[va]")
        for i in range(4):
            e(f"ds_read_b64_tr_b16 {v(VF + 2 * i, 2)}, {v(VF + 6)} offset:{off + 512 * i}")
        if mask:
            e(f"s_bfe_u32 %[t0], %[ids], 0x{(4 << 16) | (8 * r):x}")
            e(f"s_bfe_u32 %[t1], %[ids], 0x{(4 << 16) | (8 * r + 4):x}")
        e("s_waitcnt lgkmcnt(4)")
        if "prio" in abl:
            e("s_setprio 1")
        e(f"v_mfma_f32_32x32x16_f16 {v(Z, 16)}, {v(KF, 4)}, %[q00], {v(X, 16)}")
        e(f"v_mfma_f32_32x32x16_f16 {v(Y, 16)}, {v(KF, 4)}, %[q10], {v(Y, 16)}")
        e(f"v_mfma_f32_32x32x16_f16 {v(Z, 16)}, {v(KF + 4, 4)}, %[q01], {v(Z, 16)}")
        e(f"v_mfma_f32_32x32x16_f16 {v(Y, 16)}, {v(KF + 4, 4)}, %[q11], {v(Y, 16)}")
        if "prio" in abl:
            e("s_setprio 0")
        if mask:
            # mask value per (tile, 16-key band): -144.27 * |id_key_band - id_query|  (0 when the regions agree)
            ops = (("%[t0]", "%[idq0]"), ("%[t1]", "%[idq0]"), ("%[t0]", "%[idq1]"), ("%[t1]", "%[idq1]"))
            for ks in ((0, 1), (2, 3)):      # KF+4.. (k-step 1 fragment, read by the MFMA issued last) is written 7 instructions later
                for k in ks:
                    e(f"v_sub_u32 {v(KF + 2 * k)}, {ops[k][0]}, {ops[k][1]}")
                for k in ks:
                    e(f"v_cvt_f32_i32 {v(KF + 2 * k)}, {v(KF + 2 * k)}")
                for k in ks:
                    if d32:
                        e(f"v_fma_f32 {v(KF + 2 * k)}, -|{v(KF + 2 * k)}|, %[t2], %[nm{k // 2}]")
                    else:
                        e(f"v_mul_f32_e64 {v(KF + 2 * k)}, -|{v(KF + 2 * k)}|, %[t2]")
            e("s_nop 0")     # 12 VALU + this >= 12 states behind tile 0's last MFMA (one MFMA in between)
        elif d32:
            e(f"v_mov_b32 {v(KF)}, %[nm0]")
            e(f"v_mov_b32 {v(KF + 2)}, %[nm1]")
            e("s_nop 8")
        else:
            e("s_nop 10")
        for ti, S in enumerate((Z, Y)):
            if mask:
                for i in range(8):
                    e(f"v_pk_add_f32 {v(S + 2 * i, 2)}, {v(S + 2 * i, 2)}, {v(KF + 4 * ti + (2 if i >= 4 else 0), 2)} op_sel_hi:[1,0]")
            elif d32:
                for i in range(8):
                    e(f"v_pk_add_f32 {v(S + 2 * i, 2)}, {v(S + 2 * i, 2)}, {v(KF + 2 * ti, 2)} op_sel_hi:[1,0]")
            for i in range(16):
                if "noexp" in abl:
                    e(f"v_mov_b32 {v(S + i)}, {v(S + i)}")
                else:
                    e(f"v_exp_f32 {v(S + i)}, {v(S + i)}")
                if i % 2 == 1 and i >= 3:
                    j = (i - 3) // 2
                    e(f"v_cvt_pk_f16_f32 {v(S + j)}, {v(S + 2 * j)}, {v(S + 2 * j + 1)}")
            e(f"v_cvt_pk_f16_f32 {v(S + 7)}, {v(S + 14)}, {v(S + 15)}")
        # overflow test: largest packed weight of the two tiles >= 2^14 ?
        t = [KF, KF + 1, KF + 2, KF + 3]
        if "nocheck" in abl:
            e("s_waitcnt lgkmcnt(0)")
            e("s_nop 1")                   # the last cvt_pk is 2 states ahead of the MFMA that reads it
        if "nocheck" not in abl:
          e(f"v_pk_maximum3_f16 {v(t[0])}, {v(Z)}, {v(Z + 1)}, {v(Z + 2)}")
          e(f"v_pk_maximum3_f16 {v(t[1])}, {v(Z + 3)}, {v(Z + 4)}, {v(Z + 5)}")
          e(f"v_pk_maximum3_f16 {v(t[2])}, {v(Z + 6)}, {v(Z + 7)}, {v(Y)}")
          e(f"v_pk_maximum3_f16 {v(t[3])}, {v(Y + 1)}, {v(Y + 2)}, {v(Y + 3)}")
          e(f"v_pk_maximum3_f16 {v(t[0])}, {v(t[0])}, {v(Y + 4)}, {v(Y + 5)}")
          e(f"v_pk_maximum3_f16 {v(t[1])}, {v(t[1])}, {v(Y + 6)}, {v(Y + 7)}")
          e(f"v_pk_maximum3_f16 {v(t[0])}, {v(t[0])}, {v(t[2])}, {v(t[3])}")
          e(f"v_pk_max_f16 {v(t[0])}, {v(t[0])}, {v(t[1])}")
          e(f"v_pk_max_f16 {v(t[0])}, {v(t[0])}, {v(t[0])} op_sel:[0,1] op_sel_hi:[1,0]")
          e(f"v_cmp_le_u32 vcc, 0x{TRIP:08x}, {v(t[0])}")
          e(f"s_cbranch_vccnz {lab(90 + r)}f")
          e("s_waitcnt lgkmcnt(0)")
        if "prio" in abl or "priopv" in abl:
            e("s_setprio 1")
        if "nopv" not in abl:
            e(f"v_mfma_f32_32x32x16_f16 %[o0], {v(VF, 4)}, {v(Z, 4)}, %[o0]")
            e(f"v_mfma_f32_32x32x16_f16 %[o1], {v(VF, 4)}, {v(Y, 4)}, %[o1]")
            e(f"v_mfma_f32_32x32x16_f16 %[o0], {v(VF + 4, 4)}, {v(Z + 4, 4)}, %[o0]")
            e(f"v_mfma_f32_32x32x16_f16 %[o1], {v(VF + 4, 4)}, {v(Y + 4, 4)}, %[o1]")
        if "prio" in abl or "priopv" in abl:
            e("s_setprio 0")
        if d32:
            for j in range(8):
                e(f"v_dot2c_f32_f16 %[l0], 0x3c003c00, {v(Z + j)}")
                e(f"v_dot2c_f32_f16 %[l1], 0x3c003c00, {v(Y + j)}")
        e("s_add_u32 %[sb], %[sb], %[d4]")
    e(f"s_mov_b32 %[done], {rows}")
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--abl", default="", help="timing experiments: comma list of nobias,noexp,nocheck,nopv (results are wrong)")
    a = ap.parse_args()
    abl = tuple(x for x in a.abl.split(",") if x)
    with open(a.out, "w") as f:
        f.write("// generated by tools/attn_asm/gen_attn_loop.py -- do not edit (regenerate: python3 tools/attn_asm/gen_attn_loop.py --out <this file>)\n")
        for d32 in (False, True):
            for mask in (0, 1):
                f.write(f"#define ATTN_ROWS4_MASK{mask}{'_D32' if d32 else ''} \\\n")
                f.write(" \\\n".join('    "' + ln + '\\n"' for ln in gen(4, mask, abl=abl, d32=d32)) + "\n\n")
        f.write("#define ATTN_ROWS_CLOBBER " + ", ".join(f'"v{i}"' for i in range(64)) + ', "vcc", "scc", "memory"\n')


if __name__ == "__main__":
    main()
