#!/usr/bin/env python3
"""Generates voiceconversion.jl_amd/csrc/dtw_fused_asm.inc: the hand-scheduled column loop of the FUSED DTW kernel
(gfx950): observation costs and the recurrence of src/dtw.jl:104-125 in one loop, O never leaves the registers.

    python tools/gen_dtw_fused_asm.py voiceconversion.jl_amd/csrc/dtw_fused_asm.inc

Mapping.  A wave owns 128 consecutive template frames ("rows") of one pair: lane l holds rows r0 = base + 2l and
r1 = r0 + 1 with their DMAX feature values in v[0:2*DMAX-1] / v[2*DMAX:4*DMAX-1] for the whole loop.  Two rows per lane
halve what was the limiter of the stand-alone observation kernel -- the scalar-load return path that delivers the
wave-uniform sequence column as SGPR operands -- and give every FP64 accumulate chain an independent neighbour.

Per column t (one iteration):
  observation (bit-exact contract, src/dtw.jl:33-35: sequential in d, unfused):
      per d:  v_add_f64 df = s - tm ; v_mul_f64 sq = df*df ; v_add_f64 o = o + sq     (rows 0 and 1 interleaved)
      the column arrives by scalar loads in half-column chunks with two SGPR buffers A/B:
      wait(0) -> issue the loads of the NEXT chunk -> FP64 work of the current chunk.
  recurrence (src/dtw.jl:106-121 with fstep = 0, bstep = STEPS in {1,2}; candidates in the reference's scan order,
  strict '<', costs formed as (C[j,t] + o) + transition):
      row r1:  stay (C1+o1)+1 ; j=r1-2: (P1+o1)+2 ; j=r1-1: (C0+o1)+0      (C0 is still the previous column's value)
      row r0:  stay (C0+o0)+1 ; j=r0-2: (P0+o0)+2 ; j=r0-1: (P1+o0)+0
      P0/P1 = previous-column costs of rows r0-2 / r0-1 = the neighbour lane's C0/C1 (DPP wave_shr:1); lane 0 takes them
      from the previous wave's outbox in LDS (sequence-tagged ring, no workgroup barrier: the LDS executes one wave's
      operations in order), from the pair's row-strip boundary array, or +inf when there is no row below.
      "+ 0.0" is omitted: costs are >= 1, so (C+o)+0.0 == C+o bit for bit.
      min value by v_min_f64 (same value as the strict-'<' select), step code (0 stay, 1 from r-1, 2 from r-2) by
      v_cmp_lt_f64 + v_cndmask; codes are packed 2 bits per column, 16 columns per dword, and flushed to the HBM code
      table every 16 columns (global_store_dwordx2: rows r0, r1 adjacent).
  outbox: lane 63 publishes (C0, C1) of column t for the next wave, then the tag t+1.

Operands: see dtw_fused_kernel in dtw.hip.  Fixed registers (all clobbered):
  v[0:159]   template rows (row 0 at 0, row 1 at 2*DMAX)
  v[160:163] O0, O1 accumulators      v[164:171] four FP64 temporaries
  v[172:175] C0, C1 (current costs)   v[176:179] P0, P1 (neighbour costs of the previous column)
  v[180:183] X, Y candidates          v[184:185] W0, W1 packed codes     v[186:187] K0, K1 step codes
  v[188:191] M (mailbox data)         v192 tag     v193..v195 addresses / scratch     v196 this wave's tag address     v[198:201] OP0, OP1: the previous column's observation costs
  s[16:55] buffer A, s[56:95] buffer B, s[96:97] saved exec, s98, s99, vcc: temporaries
"""
import sys

BUF = {"A": 16, "B": 56}
RING = 64            # outbox slots (columns); 16 bytes each, tag at byte RING*16
NSTATE = 42          # state registers above the template rows (O0 .. OP1)


def set_layout(base=160, bufs=(16, 56)):
    """The state registers sit above the two template rows: base = 160 for every DMAX <= 40 (the register map in the
    header), 2 * 2 * stride for the wider kernels (D = 41: rows of 42 registers -> base 168; DMAX = 48: base 192).
    bufs: first SGPR of the two column buffers (40 SGPRs each for the two-chunk loops, 32 for the three-chunk ones)."""
    global O0, O1, T, C0, C1, P0, P1, X, Y, W0, W1, K0, K1, M, VTAG, VA0, VA1, VA2, VA3, OP0, OP1
    O0, O1 = base, base + 2
    T = [base + 4, base + 6, base + 8, base + 10]
    C0, C1, P0, P1 = base + 12, base + 14, base + 16, base + 18
    X, Y = base + 20, base + 22
    W0, W1, K0, K1 = base + 24, base + 25, base + 26, base + 27
    M = base + 28
    VTAG, VA0, VA1, VA2, VA3 = base + 32, base + 33, base + 34, base + 35, base + 36
    OP0, OP1 = base + 38, base + 40      # observation costs of the previous column (the recurrence runs half a column late)
    BUF["A"], BUF["B"] = bufs


set_layout()


def vp(r):
    return f"v[{r}:{r + 1}]"


def sp(r):
    return f"s[{r}:{r + 1}]"


def sloads(buf, byte_off, ndbl):
    """scalar loads of ndbl doubles starting at byte_off into buffer buf (dwordx16 / x8 / x4 / x2 pieces)"""
    out, reg, left, off = [], BUF[buf], 2 * ndbl, byte_off
    temps = ["s98", "s99", "vcc_lo", "vcc_hi"]
    for w in (16, 8, 4, 2):
        while left >= w:
            if off == 0:
                out.append(f"s_load_dwordx{w} s[{reg}:{reg + w - 1}], %[seq], %[off]")
            else:                                      # gfx9 SMEM takes an SGPR offset OR an immediate: add in SALU
                t = temps.pop(0)
                out.append(f"s_add_u32 {t}, %[off], 0x{off:x}")
                out.append(f"s_load_dwordx{w} s[{reg}:{reg + w - 1}], %[seq], {t}")
            reg, left, off = reg + w, left - w, off + 4 * w
    assert left == 0
    return out


def chunk(buf, d0, d1, first, dmax, rpl):
    """observation work of features d0..d1-1 for rpl rows per lane; dmax = registers pairs per template row"""
    out = []
    for d in range(d0, d1):
        s = sp(BUF[buf] + 2 * (d - d0))
        for q in range(rpl):
            t = T[(2 * d + q) % 4] if rpl == 2 else T[d % 2 * 2]
            o = O0 if q == 0 else O1
            tm = 2 * dmax * q + 2 * d
            out.append(f"v_add_f64 {vp(t)}, {s}, -{vp(tm)}")
        for q in range(rpl):
            t = T[(2 * d + q) % 4] if rpl == 2 else T[d % 2 * 2]
            o = O0 if q == 0 else O1
            if first and d == d0:
                out.append(f"v_mul_f64 {vp(o)}, {vp(t)}, {vp(t)}")
            else:
                out.append(f"v_mul_f64 {vp(t)}, {vp(t)}, {vp(t)}")
        if not (first and d == d0):
            for q in range(rpl):
                t = T[(2 * d + q) % 4] if rpl == 2 else T[d % 2 * 2]
                o = O0 if q == 0 else O1
                out.append(f"v_add_f64 {vp(o)}, {vp(o)}, {vp(t)}")
    return out


def template_loads(dmax, rpl):
    L = []
    for q in range(rpl):
        row = "%[rowA]" if q == 0 else "%[rowB]"
        for k in range(dmax // 2):
            b = 2 * dmax * q + 4 * k
            L.append(f"global_load_dwordx4 v[{b}:{b + 3}], {row}, off offset:{16 * k}")
    return L


def advance(col):
    # advance to the next column (stay on the last one: its prefetch is a harmless re-read)
    return ["s_cmp_eq_u32 %[n], 1", f"s_cselect_b32 vcc_lo, 0, 0x{col:x}", "s_add_u32 %[off], %[off], vcc_lo"]


def body_obs_only(dmax, rpl, variant=0):
    """observation costs only, stored to the (S,T) workspace like dtw_obs_asm_kernel (used by tools/microbench_obs2.hip to
    compare one and two rows per lane on otherwise identical loops)"""
    h0 = (dmax // 2 + 1) // 2 * 2
    h1 = dmax - h0
    col = 8 * dmax
    L = ["s_mov_b64 s[96:97], exec"]
    L += template_loads(dmax, rpl)
    L += sloads("A", 0, h0)
    L.append("s_waitcnt vmcnt(0)")
    L.append("1:")
    # timing variants (wrong results on purpose): 1 = the column never advances (scalar cache always hits), 2 = only the
    # first half of the column is loaded, 3 = no loads in the loop, 4 = variant 2 without the second wait
    L.append("s_waitcnt lgkmcnt(0)")
    if variant in (0, 1):
        L += sloads("B", 8 * h0, h1)
    L += chunk("A", 0, h0, True, dmax, rpl)
    if variant != 1:
        L += advance(col)
    if variant != 4:
        L.append("s_waitcnt lgkmcnt(0)")
    if variant != 3:
        L += sloads("A", 0, h0)
    L += chunk("B" if variant in (0, 1) else "A", h0, dmax, False, dmax, rpl)
    if variant == 5:      # no stores: the loop alone (results folded into a running sum so that nothing is dead)
        L += [f"v_add_f64 {vp(T[3])}, {vp(T[3])}, {vp(O0)}"]
    elif rpl == 2:
        L += ["s_mov_b64 exec, %[m0]", f"global_store_dwordx4 %[optr], v[{O0}:{O0 + 3}], off"]
    else:
        L += ["s_mov_b64 exec, %[m0]", f"global_store_dwordx2 %[optr], {vp(O0)}, off"]
    L += ["s_mov_b64 exec, s[96:97]",
          "v_lshl_add_u64 %[optr], %[optr], 0, %[stride]",
          "s_sub_u32 %[n], %[n], 1", "s_cmp_lg_u32 %[n], 0", "s_cbranch_scc1 1b",
          "s_waitcnt lgkmcnt(0)"]
    return L


def rec_row(c, o, pa, pb, k, steps):
    """one row of the recurrence, in place on the cost register c.
    pa: cost of row-2 (previous column), pb: cost of row-1 (previous column); k: step-code register.
    The candidates are formed BEFORE c is overwritten by the caller when they alias (see body_fused)."""
    L = []
    if steps == 2:
        L += [f"v_add_f64 {vp(X)}, {vp(pa)}, {vp(o)}", f"v_add_f64 {vp(X)}, {vp(X)}, 2.0"]
    L += [f"v_add_f64 {vp(Y)}, {vp(pb)}, {vp(o)}"]
    L += [f"v_add_f64 {vp(c)}, {vp(c)}, {vp(o)}", f"v_add_f64 {vp(c)}, {vp(c)}, 1.0"]
    if steps == 2:
        L += [f"v_cmp_lt_f64 vcc, {vp(X)}, {vp(c)}", f"v_cndmask_b32_e64 v{k}, 0, 2, vcc", f"v_min_f64 {vp(c)}, {vp(c)}, {vp(X)}"]
        L += [f"v_cmp_lt_f64 vcc, {vp(Y)}, {vp(c)}", f"v_cndmask_b32_e64 v{k}, v{k}, 1, vcc", f"v_min_f64 {vp(c)}, {vp(c)}, {vp(Y)}"]
    else:
        L += [f"v_cmp_lt_f64 vcc, {vp(Y)}, {vp(c)}", f"v_cndmask_b32_e64 v{k}, 0, 1, vcc", f"v_min_f64 {vp(c)}, {vp(c)}, {vp(Y)}"]
    return L


OUTBOX = RING * 16 + 16   # bytes of one wave's outbox: RING slots, then the tag (one dword)


def rec_block(steps, role, nomail):
    """recurrence of column c = t-1 (its observation costs are in OP0/OP1; lane 0's neighbour costs were requested
    half a column ago into M), code packing / flush, neighbour shift, publication.  Emitted in the middle of column t
    -- so that the LDS traffic it causes (outbox write) and consumes (M) completes under the FP64 work of a half column
    instead of in front of an s_waitcnt -- and once more after the loop for the last column."""
    L = []
    if role != 0:
        # lane 0 takes M; lanes 1..63 already hold their neighbour's costs (DPP at the end of the previous recurrence)
        L += ["s_mov_b64 vcc, 1",
              f"v_cndmask_b32 v{P0}, v{P0}, v{M}, vcc", f"v_cndmask_b32 v{P0 + 1}, v{P0 + 1}, v{M + 1}, vcc",
              f"v_cndmask_b32 v{P1}, v{P1}, v{M + 2}, vcc", f"v_cndmask_b32 v{P1 + 1}, v{P1 + 1}, v{M + 3}, vcc"]
    # row r1 first: its j = r1-1 candidate needs C0 of the previous column
    L += rec_row(C1, OP1, P1, C0, K1, steps)
    L += rec_row(C0, OP0, P0, P1, K0, steps)
    # pack the codes: shift = 2 * (c & 15)
    L += ["s_sub_u32 s99, %[t], 1", "s_and_b32 s98, s99, 15", "s_lshl_b32 s99, s98, 1",
          f"v_lshl_or_b32 v{W0}, v{K0}, s99, v{W0}", f"v_lshl_or_b32 v{W1}, v{K1}, s99, v{W1}"]
    # neighbour costs for the next column: lane l takes lane l-1's (C0, C1); lane 0 keeps its old value (fixed up above)
    L += ["s_nop 1"]
    for dst, src in ((P0, C0), (P0 + 1, C0 + 1), (P1, C1), (P1 + 1, C1 + 1)):
        L.append(f"v_mov_b32_dpp v{dst}, v{src} wave_shr:1 row_mask:0xf bank_mask:0xf")
    # flush 16 columns of codes
    L += ["s_cmp_lg_u32 s98, 15", "s_cbranch_scc1 41f"]
    L += flush_codes()
    if not nomail:
        # back-pressure, once per 16 columns: before the slots of columns c+1..c+16 are rewritten the next wave must be
        # past the columns c-63..c-48 they still hold, i.e. its tag (columns finished) must be >= c-46
        L += ["s_bitcmp1_b32 %[mode], 2", "s_cbranch_scc0 41f", "s_cmp_lt_u32 %[t], 48", "s_cbranch_scc1 41f",
              f"s_add_u32 s99, %[out], {OUTBOX + RING * 16}", f"v_mov_b32 v{VA2}, s99",
              "s_sub_u32 s98, %[t], 47",
              "42:", f"ds_read_b32 v{VTAG}, v{VA2}", "s_waitcnt lgkmcnt(0)", f"v_readfirstlane_b32 s99, v{VTAG}",
              "s_cmp_ge_u32 s99, s98", "s_cbranch_scc1 41f", "s_sleep 1", "s_branch 42b"]
    L += ["41:"]
    if not nomail:
        # publish (C0, C1) of column c (lane 63) and the tag c+1 = t; export the strip's top two rows
        L += ["s_sub_u32 s98, %[t], 1", f"s_and_b32 s98, s98, {RING - 1}", "s_lshl_b32 s98, s98, 4", "s_add_u32 s98, s98, %[out]",
              f"v_mov_b32 v{VA2}, s98", f"v_mov_b32 v{VTAG}, %[t]",
              "s_mov_b32 exec_lo, 0", "s_mov_b32 exec_hi, 0x80000000",
              f"ds_write_b128 v{VA2}, v[{C0}:{C0 + 3}]", f"ds_write_b32 v{VA3}, v{VTAG}",
              "s_mov_b64 exec, s[96:97]",
              "s_cmp_lt_i32 %[explane], 0", "s_cbranch_scc1 50f",
              "s_lshl_b64 s[98:99], 1, %[explane]", "s_mov_b64 exec, s[98:99]",
              # sc1: written through to the agent-wide coherence point -- the strip above may run on another XCD, whose L2 is not
              # coherent with this one; its loads carry sc1 too, and no job needs an L2-wide write-back / invalidate
              f"global_store_dwordx4 %[gout], v[{C0}:{C0 + 3}], off sc1", "s_mov_b64 exec, s[96:97]",
              "v_lshl_add_u64 %[gout], %[gout], 0, 16",
              "50:"]
    return L


def m_request(role):
    """request lane 0's neighbour costs for the recurrence of column c = t-1: they are the costs of column c-1 = t-2"""
    if role == 1:
        return ["s_sub_u32 s98, %[t], 2", f"s_and_b32 s98, s98, {RING - 1}", "s_lshl_b32 s98, s98, 4", "s_add_u32 s98, s98, %[out]",
                f"s_sub_u32 s98, s98, {OUTBOX}", f"v_mov_b32 v{VA0}, s98",   # (VA0, VA1 point into the PREVIOUS wave's outbox)
                f"ds_read_b32 v{VTAG}, v{VA1}", f"ds_read_b128 v[{M}:{M + 3}], v{VA0}"]
    if role == 2:
        return ["s_sub_u32 s98, %[t], 1", "s_lshl_b32 s98, s98, 4", "s_add_u32 s98, s98, %[bnd]", f"v_mov_b32 v{VA0}, s98",
                f"ds_read_b128 v[{M}:{M + 3}], v{VA0}"]
    return []


def m_check(role):
    """the previous wave must have finished column c-1 = t-2, i.e. tag >= t-1; normally it has, half a column ago"""
    if role != 1:
        return []
    return ["s_sub_u32 s98, %[t], 1", f"v_readfirstlane_b32 s99, v{VTAG}", "s_cmp_ge_u32 s99, s98", "s_cbranch_scc1 22f",
            "21:", "s_sleep 1",
            f"ds_read_b32 v{VTAG}, v{VA1}", f"ds_read_b128 v[{M}:{M + 3}], v{VA0}",
            "s_waitcnt lgkmcnt(0)",
            f"v_readfirstlane_b32 s99, v{VTAG}", "s_cmp_ge_u32 s99, s98", "s_cbranch_scc0 21b",
            "22:"]


def fused_loop(dmax, steps, role, variant=0):
    """one specialisation of the column loop.  role: where lane 0 takes its neighbour costs (rows r0-2, r0-1 of the
    previous column) from -- 0: nowhere (the wave starts at row 0 of the pair: +inf stays in place), 1: the previous
    wave's outbox (ring + tag), 2: the strip's boundary array (linear, complete before the loop starts)."""
    h0 = (dmax // 2 + 1) // 2 * 2
    h1 = dmax - h0
    col = 8 * dmax
    noloads = variant in (1, 2)      # timing variants (wrong results): 1 no scalar loads, 2 also no wave coupling, 3 no coupling
    nomail = variant in (2, 3)
    if nomail:
        role = 0
    L = []
    L.append("1:")
    L.append("s_waitcnt lgkmcnt(0)")
    if not noloads:
        L += sloads("B", 8 * h0, h1)
    L += ["s_cmp_eq_u32 %[t], 0", "s_cbranch_scc1 10f"] + m_request(role) + ["10:"]
    L += chunk("A", 0, h0, True, dmax, 2)
    L += advance(col)
    L.append("s_waitcnt lgkmcnt(0)")
    if not noloads:
        L += sloads("A", 0, h0)
    # the recurrence of the PREVIOUS column, between the two halves of this column's observation work
    L += ["s_cmp_eq_u32 %[t], 0", "s_cbranch_scc1 11f"] + m_check(role) + rec_block(steps, role, nomail) + ["11:"]
    L += chunk("B", h0, dmax, False, dmax, 2)
    L += [f"v_mov_b64 {vp(OP0)}, {vp(O0)}", f"v_mov_b64 {vp(OP1)}, {vp(O1)}"]
    L += ["s_add_u32 %[t], %[t], 1", "s_sub_u32 %[n], %[n], 1", "s_cmp_lg_u32 %[n], 0", "s_cbranch_scc1 1b"]
    # the last column's recurrence
    L += m_request(role) + ["s_waitcnt lgkmcnt(0)"] + m_check(role) + rec_block(steps, role, nomail)
    return L


def splits3(d):
    """a column of d doubles in three chunks of at most 16 (two 32-SGPR buffers): [14, 14, 13] for 41, [16, 16, 16] for 48"""
    a = (d + 2) // 3
    a += a & 1 if 3 * (a + (a & 1)) - d <= (a + (a & 1)) else 0
    b = min(a, d - a)
    c = d - a - b
    assert 0 < c <= a <= 16 and b <= 16
    return [a, b, c]


def template_loads_exact(d, stride, rpl):
    """rows of exactly d doubles (not padded in memory): d // 2 16-byte loads and, for odd d, one 8-byte load"""
    L = []
    for q in range(rpl):
        row = "%[rowA]" if q == 0 else "%[rowB]"
        for k in range(d // 2):
            b = 2 * stride * q + 4 * k
            L.append(f"global_load_dwordx4 v[{b}:{b + 3}], {row}, off offset:{16 * k}")
        if d & 1:
            b = 2 * stride * q + 2 * (d - 1)
            L.append(f"global_load_dwordx2 v[{b}:{b + 1}], {row}, off offset:{8 * (d - 1)}")
    return L


def fused_loop3(d, stride, steps, role):
    """The column loop for 40 < d <= 48.  A column no longer fits two half-column SGPR buffers (84 SGPRs are all there
    is), so it arrives in THREE chunks through two 32-SGPR buffers; with an odd chunk count the buffers swap roles from
    one column to the next, hence the loop is unrolled twice (A,B,A | B,A,B).  The recurrence of the previous column
    sits after the first chunk, as in the two-chunk loop: its LDS traffic completes under a third of a column of FP64
    work.  Same arithmetic, same order: sequential in d, unfused."""
    c = splits3(d)
    col = 8 * d
    L = ["1:"]
    for X_, Y_ in (("A", "B"), ("B", "A")):
        L.append("s_waitcnt lgkmcnt(0)")
        L += sloads(Y_, 8 * c[0], c[1])
        L += ["s_cmp_eq_u32 %[t], 0", "s_cbranch_scc1 10f"] + m_request(role) + ["10:"]
        L += chunk(X_, 0, c[0], True, stride, 2)
        L.append("s_waitcnt lgkmcnt(0)")
        L += sloads(X_, 8 * (c[0] + c[1]), c[2])
        L += ["s_cmp_eq_u32 %[t], 0", "s_cbranch_scc1 11f"] + m_check(role) + rec_block(steps, role, False) + ["11:"]
        L += chunk(Y_, c[0], c[0] + c[1], False, stride, 2)
        L += advance(col)
        L.append("s_waitcnt lgkmcnt(0)")
        L += sloads(Y_, 0, c[0])                 # the next column's first chunk
        L += chunk(X_, c[0] + c[1], d, False, stride, 2)
        L += [f"v_mov_b64 {vp(OP0)}, {vp(O0)}", f"v_mov_b64 {vp(OP1)}, {vp(O1)}"]
        L += ["s_add_u32 %[t], %[t], 1", "s_sub_u32 %[n], %[n], 1", "s_cmp_lg_u32 %[n], 0"]
        L += ["s_cbranch_scc0 2f"] if X_ == "A" else ["s_cbranch_scc1 1b"]
    L += ["2:"]
    L += m_request(role) + ["s_waitcnt lgkmcnt(0)"] + m_check(role) + rec_block(steps, role, False)
    return L


def body_fused3(d, stride, steps):
    """dtw_fused_kernel for 40 < d <= 48; operands as body_fused without %[c0] %[c1] %[p0] %[p1] (read from LDS).  stride = register pairs per template row: rows of
    d = 41 doubles are read as they lie in memory (no padded copy: 20 x 16 bytes + 8 bytes per row, column stride 328
    bytes for the scalar loads), d = 48 is the padded kernel of D = 42..48."""
    set_layout(4 * stride, (16, 48))
    c = splits3(d)
    L = ["s_mov_b64 s[96:97], exec"]
    L += template_loads_exact(d, stride, 2)
    L += sloads("A", 0, c[0])
    # The initial costs come from LDS -- (C0, C1) at %[clast], (P0, P1) 512 doubles behind, left there by the C++
    # prologue -- instead of four 64-bit operands: with 192 template registers every operand VGPR counts (256 = two
    # waves per SIMD).  (A lane reads back what it wrote itself; the LDS executes a wave's operations in order.)
    L += [f"ds_read_b128 v[{C0}:{C0 + 3}], %[clast]", f"ds_read_b128 v[{P0}:{P0 + 3}], %[clast] offset:4096"]
    L += [f"v_mov_b32 v{W0}, 0", f"v_mov_b32 v{W1}, 0",
          "s_sub_u32 s98, %[out], 16", f"v_mov_b32 v{VA1}, s98",
          f"s_add_u32 s98, %[out], {RING * 16}", f"v_mov_b32 v{VA3}, s98"]
    L.append("s_waitcnt vmcnt(0)")
    L += ["s_bitcmp1_b32 %[mode], 0", "s_cbranch_scc1 100f", "s_bitcmp1_b32 %[mode], 1", "s_cbranch_scc1 200f"]
    L += fused_loop3(d, stride, steps, 0) + ["s_branch 300f"]
    L += ["100:"] + fused_loop3(d, stride, steps, 1) + ["s_branch 300f"]
    L += ["200:"] + fused_loop3(d, stride, steps, 2)
    L += ["300:"]
    L += ["s_and_b32 s98, %[t], 15", "s_cmp_eq_u32 s98, 0", "s_cbranch_scc1 301f"]
    L += flush_codes()
    L += ["301:", f"ds_write_b128 %[clast], v[{C0}:{C0 + 3}]", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
    set_layout()
    return L


def flush_codes():
    return ["s_bfm_b64 s[98:99], %[cnt], 0", "s_cmp_eq_u32 %[cnt], 64", "s_cselect_b64 exec, s[96:97], s[98:99]",
            f"global_store_dwordx2 %[codes], v[{W0}:{W1}], off", "s_mov_b64 exec, s[96:97]",
            "v_lshl_add_u64 %[codes], %[codes], 0, %[cstride]",
            "s_nop 0", f"v_mov_b32 v{W0}, 0", f"v_mov_b32 v{W1}, 0"]


def body_fused(dmax, steps, variant=0):
    """the fused loop, two rows per lane.  Operands (SGPRs are scarce: 84 of them are the two column buffers):
      %[seq] s64 sequence base     %[off] s32 rw byte offset of the column     %[n] s32 rw columns left
      %[t] s32 rw column index (starts at 0)
      %[cstride] s64 bytes between consecutive 16-column groups of the code table
      %[cnt] s32 number of lanes (from lane 0) whose rows exist: they store step codes
      %[out] s32 LDS byte address of this wave's outbox (RING slots of 16 bytes, then the tag); the outboxes of the
             workgroup's waves are consecutive, OUTBOX bytes apart.  The prologue (C++) leaves the initial costs of the
             wave's top two rows in slot RING-1 and zeroes the tags.
      %[bnd] s32 LDS address of the strip's boundary array: entry 0 = initial costs of the two rows below the strip,
             entry 1+t = their costs after column t
      %[mode] s32 bit 0: a previous wave exists (lane 0 reads its outbox); bit 1: wave 0 of a strip with rows below
             (lane 0 reads the boundary array); bit 2: a next wave exists (back-pressure on its tag)
      %[explane] s32 lane whose (C0, C1) are the top two rows of a non-final strip, exported per column to %[gout]; -1: none
      %[rowA] %[rowB] v64 template row addresses
      %[c0] %[c1] v64 (double) initial costs r0+1, r1+1      %[p0] %[p1] v64 (double) neighbour costs for column 0
      %[codes] v64 rw address of this lane's code dwords (rows r0, r1 adjacent)
      %[clast] v32 LDS byte address where the lane leaves (C0, C1) after the last column
      %[gout] v64 rw global address of the exported boundary pair of column t
    """
    h0 = (dmax // 2 + 1) // 2 * 2
    L = ["s_mov_b64 s[96:97], exec"]
    L += template_loads(dmax, 2)
    L += sloads("A", 0, h0)
    if variant in (1, 2):
        L += sloads("B", 8 * h0, dmax - h0)
    L += [f"v_mov_b64 {vp(C0)}, %[c0]", f"v_mov_b64 {vp(C1)}, %[c1]", f"v_mov_b64 {vp(P0)}, %[p0]", f"v_mov_b64 {vp(P1)}, %[p1]",
          f"v_mov_b32 v{W0}, 0", f"v_mov_b32 v{W1}, 0",
          # loop-invariant LDS addresses: the previous wave's tag (VA1) and this wave's tag (VA3)
          "s_sub_u32 s98, %[out], 16", f"v_mov_b32 v{VA1}, s98",
          f"s_add_u32 s98, %[out], {RING * 16}", f"v_mov_b32 v{VA3}, s98"]
    L.append("s_waitcnt vmcnt(0)")
    L += ["s_bitcmp1_b32 %[mode], 0", "s_cbranch_scc1 100f", "s_bitcmp1_b32 %[mode], 1", "s_cbranch_scc1 200f"]
    L += fused_loop(dmax, steps, 0, variant) + ["s_branch 300f"]
    L += ["100:"] + fused_loop(dmax, steps, 1, variant) + ["s_branch 300f"]
    L += ["200:"] + fused_loop(dmax, steps, 2, variant)
    L += ["300:"]
    # tail of the code table (T not a multiple of 16), the last cost column, drain
    L += ["s_and_b32 s98, %[t], 15", "s_cmp_eq_u32 s98, 0", "s_cbranch_scc1 301f"]
    L += flush_codes()
    L += ["301:", f"ds_write_b128 %[clast], v[{C0}:{C0 + 3}]", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
    return L


def main():
    out = ["// GENERATED by tools/gen_dtw_fused_asm.py -- do not edit.  Column loops of dtw_fused_kernel<DMAX,STEPS> and of the",
           "// observation-only comparison loops of tools/microbench_obs2.hip (see the generator for schedule and register map)."]
    regs = [f'"v{i}"' for i in range(202)] + [f'"s{i}"' for i in range(16, 100)]
    out.append("#define VCMI_FUSED_ASM_CLOBBERS " + ", ".join(regs))
    for name, stride in (("D41", 42), ("D48", 48)):
        regs = [f'"v{i}"' for i in range(4 * stride + NSTATE)] + [f'"s{i}"' for i in range(16, 100)]
        out.append(f"#define VCMI_FUSED_ASM_CLOBBERS_{name} " + ", ".join(regs))
    out.append(f"#define VCMI_FUSED_RING {RING}")
    out.append(f"#define VCMI_FUSED_OUTBOX {OUTBOX}")
    out.append("")

    def emit(name, lines):
        out.append(f"#define {name} \\")
        for i, l in enumerate(lines):
            out.append(f'  "{l}\\n"' + (" \\" if i + 1 < len(lines) else ""))
        out.append("")

    for dmax in (8, 16, 24, 32, 40):
        for steps in (1, 2):
            emit(f"VCMI_DTW_FUSED_ASM_D{dmax}_S{steps}", body_fused(dmax, steps))
    for steps in (1, 2):
        emit(f"VCMI_DTW_FUSED_ASM_D41_S{steps}", body_fused3(41, 42, steps))
        emit(f"VCMI_DTW_FUSED_ASM_D48_S{steps}", body_fused3(48, 48, steps))
    for v in (1, 2, 3):
        emit(f"VCMI_DTW_FUSED_ASM_D40_S2_V{v}", body_fused(40, 2, v))
    for rpl in (1, 2):
        emit(f"VCMI_DTW_OBS2_ASM_D40_R{rpl}", body_obs_only(40, rpl))
    for v in (1, 2, 3, 4, 5):
        emit(f"VCMI_DTW_OBS2_ASM_D40_R2_V{v}", body_obs_only(40, 2, v))
    emit("VCMI_DTW_OBS2_ASM_D40_R1_V5", body_obs_only(40, 1, 5))
    open(sys.argv[1], "w").write("\n".join(out))


if __name__ == "__main__":
    main()
