#!/usr/bin/env python3
"""csrc/gen_fir_loop.py -- writes csrc/fir_loop_asm.inc: the FIR inner loop of the period kernel
(kernels_period.hip, fir_group) as hand-written gfx950 ISA, one inline-asm body per variant.

Why ISA (round 3): the loop is 40-60 v_pk_fma_f32 per trip with wave-uniform SGPR taps, and everything
around them is bookkeeping that hipcc could only do with vector instructions -- the trip count in a VGPR
(v_add_co + branch on vcc), a negative-offset address pair per bank -- because no SGPR is left under the
80-SGPR cap that keeps two workgroups resident per CU; and three copies of the loop (head / main / tail
rows) cost ~25 register moves per group to keep the accumulators where each copy wants them.  Here:
  * the tap SGPRs and the sample VGPRs are named registers, so the three row subsets share them;
  * trip counts are scalar (s_sub_u32 + s_cbranch on the borrow), the tap address is one SGPR offset
    beside an unchanging base (s_load ... sbase, soffset offset:imm), LDS offsets are immediates;
  * single-channel lanes read their two periods straight into the halves of one register pair
    (hipcc re-paired them with ~7 v_mov per 4 steps);
  * the bank padding of a padded window is stepped over with scalar selects;
  * W16 variants read an int16 LDS window (half the bytes: twice the periods per tile for the wide
    windows of down-sampling ratios) and convert behind the wait (v_cvt_f32_i32_sdwa).
Semantics are those of the C++ loop it replaces (which stays in fir_group for layouts without a variant
here): reference deps/speex/resample.c:438-496 / :331-384 with the effective taps, same order of
accumulation per output, so results are bit-identical to the round-2 kernel.

Structure per trip (two tap banks A, B; S steps = S*R taps each):
    wait lgkmcnt(0) | issue loads B | S*R FMAs on A | wait | advance, issue loads A' | S*R FMAs on B
(scalar loads return out of order and share lgkmcnt with LDS, so every wait is a full drain and a bank's
loads are issued right behind the other bank's wait; tools/gen_ubench_loop.py weighed the alternatives.)

usage: python csrc/gen_fir_loop.py   (the .inc is committed; rerun after editing this file)
"""
import os

# SGPR homes of the tap banks: (first register, count) pieces, one scalar load each.  s32 is reserved by
# the backend; the 80-SGPR cap leaves s0..s73.
BANKS = {
    (10, 2): dict(A=[(28, 4), (36, 16)], B=[(52, 16), (68, 4)]),
    (5, 6): dict(A=[(4, 16), (20, 8), (28, 4), (36, 2)], B=[(40, 16), (56, 8), (64, 4), (38, 2)]),
    (5, 4): dict(A=[(28, 4), (36, 16)], B=[(52, 16), (68, 4)]),
}
SAMPLE_BASE = {10: 40, 5: 64}   # first physical VGPR of the sample pairs (then the raw int16 dwords of W16)


def tap_reg(bank, t):
    for first, n in bank:
        if t < n:
            return first + t
        t -= n
    raise ValueError(t)


def bank_regs(bank):
    return [r for first, n in bank for r in range(first, first + n)]


class Variant:
    def __init__(self, R, S, CT, CF, padded, w16):
        self.R, self.S, self.CT, self.CF, self.padded, self.w16 = R, S, CT, CF, padded, w16
        self.banks = BANKS[(R, S)]
        self.eb = 2 if w16 else 4                     # bytes per LDS element
        self.vbase = SAMPLE_BASE[R]
        self.name = "R%d_S%d_CT%d_CF%d_P%d_W%d" % (R, S, CT, CF, int(padded), int(w16))

    # sample pair k (0 .. 2S-1): bank A holds pairs 0..S-1, bank B pairs S..2S-1
    def pair(self, k):
        return self.vbase + 2 * k

    def raw(self, k):
        return self.vbase + 4 * self.S + k

    def vgprs(self):
        n = 4 * self.S + (2 * self.S if (self.w16 and self.CT == 2) else 0)
        return list(range(self.vbase, self.vbase + n))

    def fma_bank(self, which, lo, hi):
        bank = self.banks[which]
        out = []
        for u in range(self.S):
            k = u if which == "A" else self.S + u
            for i in range(lo, hi):
                r = tap_reg(bank, u * self.R + i)
                h = r & 1
                p = r - h
                out.append("v_pk_fma_f32 %%[a%d], s[%d:%d], v[%d:%d], %%[a%d] op_sel:[%d,0,0] op_sel_hi:[%d,1,1]"
                           % (i, p, p + 1, self.pair(k), self.pair(k) + 1, i, h, h))
        return out

    def tap_loads(self, which, byte_off):
        out, t = [], 0
        for f, m in self.banks[which]:
            out.append("s_load_dword%s s[%d:%d], %%[rows], %%[off] offset:0x%x"
                       % ("x%d" % m if m > 1 else "", f, f + m - 1, byte_off + 4 * t))
            t += m
        return out

    def sample_reads(self, which, first_step):
        out = []
        if self.CT == 2 and not self.w16 and self.S % 2 == 0 and os.environ.get("SPEEXHIP_GEN_READ2"):
            # Experiment (SPEEXHIP_GEN_READ2=1 python csrc/gen_fir_loop.py): two steps per instruction -- the pairs of
            # consecutive steps are consecutive registers.  Half the LDS instructions (8.2 M -> 4.1 M in the cfg2
            # 32-stream launch), same time: cfg2 one stream 11.7-11.9 vs 11.7-12.2 us, 32 streams 191.5-193.7 vs
            # 193.0-194.4, cfg4 550 vs 551, float 263 vs 262; 4 channels 415.8 -> 405.4.  Not the default.
            for u in range(0, self.S, 2):
                k = u if which == "A" else self.S + u
                o0 = (first_step + u) * self.CF * 4 // 8
                o1 = (first_step + u + 1) * self.CF * 4 // 8
                p = self.pair(k)
                out.append("ds_read2_b64 v[%d:%d], %%[addr] offset0:%d offset1:%d" % (p, p + 3, o0, o1))
            return out
        for u in range(self.S):
            k = u if which == "A" else self.S + u
            o = (first_step + u) * self.CF * self.eb
            p = self.pair(k)
            if self.CT == 2 and not self.w16:
                out.append("ds_read_b64 v[%d:%d], %%[addr] offset:%d" % (p, p + 1, o))
            elif self.CT == 2:
                out.append("ds_read_b32 v%d, %%[addr] offset:%d" % (self.raw(k), o))
            elif not self.w16:
                out += ["ds_read_b32 v%d, %%[addr] offset:%d" % (p, o), "ds_read_b32 v%d, %%[addr2] offset:%d" % (p + 1, o)]
            else:
                out += ["ds_read_i16 v%d, %%[addr] offset:%d" % (p, o), "ds_read_i16 v%d, %%[addr2] offset:%d" % (p + 1, o)]
        return out

    def converts(self, which):
        if not self.w16:
            return []
        out = []
        for u in range(self.S):
            k = u if which == "A" else self.S + u
            p = self.pair(k)
            if self.CT == 2:
                out += ["v_cvt_f32_i32_sdwa v%d, sext(v%d) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" % (p, self.raw(k)),
                        "v_cvt_f32_i32_sdwa v%d, sext(v%d) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" % (p + 1, self.raw(k))]
            else:
                out += ["v_cvt_f32_i32_e32 v%d, v%d" % (p, p), "v_cvt_f32_i32_e32 v%d, v%d" % (p + 1, p + 1)]
        return out

    def loop(self, label, cnt, lo, hi):
        """one copy of the loop over the rows [lo, hi) of the group; `cnt` trips (an SGPR operand, may be 0).
        Invariant at entry and exit: bank A of the next trip (taps + samples) is in flight."""
        S, R = self.S, self.R
        bank_bytes = 4 * S * R
        adv = 2 * S * self.CF * self.eb                       # window bytes per trip
        body = ["s_sub_u32 %%[%s], %%[%s], 1" % (cnt, cnt), "s_cbranch_scc1 %d1f" % label, "%d0:" % label]
        body += ["s_waitcnt lgkmcnt(0)"] + self.tap_loads("B", bank_bytes) + self.sample_reads("B", S)
        body += self.converts("A") + self.fma_bank("A", lo, hi)
        body += ["s_waitcnt lgkmcnt(0)"]
        if self.padded:
            # the window pointer steps over the bank padding when the trip count-down to the next period
            # boundary reaches zero (wave-uniform; the host placed every boundary between two trips)
            body += ["s_sub_u32 %[wrap], %[wrap], 1", "s_cmp_eq_u32 %[wrap], 0",
                     "s_cselect_b32 %[tmp], %[advpad], " + str(adv), "s_cselect_b32 %[wrap], %[wrapstep], %[wrap]",
                     "v_add_u32 %[addr], %[tmp], %[addr]"]
            if self.CT == 1:
                body += ["v_add_u32 %[addr2], %[tmp], %[addr2]"]
        else:
            body += ["v_add_u32 %%[addr], %d, %%[addr]" % adv]
            if self.CT == 1:
                body += ["v_add_u32 %%[addr2], %d, %%[addr2]" % adv]
        body += self.tap_loads("A", 2 * bank_bytes) + self.sample_reads("A", 0)
        body += self.converts("B") + self.fma_bank("B", lo, hi)
        body += ["s_add_u32 %%[off], %%[off], 0x%x" % (2 * bank_bytes), "s_sub_u32 %%[%s], %%[%s], 1" % (cnt, cnt),
                 "s_cbranch_scc0 %d0b" % label, "%d1:" % label]
        return body

    def lines(self):
        pro = self.tap_loads("A", 0) + self.sample_reads("A", 0)
        if self.R == 10:
            body = self.loop(1, "head", 0, 5) + self.loop(2, "main", 0, 10) + self.loop(3, "tail", 5, 10)
        else:
            body = self.loop(2, "main", 0, self.R)
        return pro + body + ["s_waitcnt lgkmcnt(0)"]

    def function(self):
        R = self.R
        asm = "\n".join('      "%s\\n"' % l for l in self.lines())
        outs = ['[a%d] "+v"(acc[%d])' % (i, i) for i in range(R)] + ['[addr] "+v"(addr)']
        if self.CT == 1:
            outs.append('[addr2] "+v"(addr2)')
        outs += ['[off] "+s"(off)', '[main] "+s"(main)']
        if R == 10:
            outs += ['[head] "+s"(head)', '[tail] "+s"(tail)']
        ins = ['[rows] "s"(rows_g)']
        if self.padded:
            outs += ['[wrap] "+s"(to_wrap)', '[tmp] "=&s"(tmp)']
            ins += ['[advpad] "s"(adv_pad)', '[wrapstep] "s"(wrap_step)']
        clob = ['"s%d"' % r for r in bank_regs(self.banks["A"]) + bank_regs(self.banks["B"])] + ['"v%d"' % r for r in self.vgprs()]
        return '''template <>
struct FirLoopAsm<%d, %d, %d, %s, %s> {
  static constexpr bool available = true;
  static constexpr int steps_per_bank = %d;
  // rows_g: the group's tap rows; addr / addr2: LDS byte addresses of the lane's first sample (second period);
  // head / main / tail: trips on rows [0, R/2) / all rows / rows [R/2, R); to_wrap, wrap_step: trips to the
  // first / between period boundaries of a padded window, adv_pad: window bytes per trip + the padding
  static __device__ __forceinline__ void run(f32x2 (&acc)[%d], const float *rows_g, uint32_t addr, uint32_t addr2,
                                             uint32_t head, uint32_t main, uint32_t tail, uint32_t to_wrap,
                                             uint32_t wrap_step, uint32_t adv_pad) {
    uint32_t off = 0, tmp;
    (void)tmp; (void)addr2; (void)head; (void)tail; (void)to_wrap; (void)wrap_step; (void)adv_pad;
    asm volatile(
%s
      : %s
      : %s
      : %s, "scc", "memory");
  }
};
''' % (R, self.CT, self.CF, "true" if self.padded else "false", "true" if self.w16 else "false", self.S, R, asm,
       ", ".join(outs), ", ".join(ins), ", ".join(clob))


class Variant64(Variant):
    """The loop with an fp64 accumulator (round 4): what FAST mode runs for the reference's double kernels (quality
    9 and 10, deps/speex/resample.c:389-435, :501-558).  Same structure, same SGPR homes -- a bank is half as many
    taps, each a double in an aligned SGPR pair --, the samples widened in place behind the wait (v_cvt_f64_f32),
    one v_fma_f64 per tap and half of the lane's pair: every product exact, the sum fp64 throughout.  v_fma_f64
    issues at the rate of v_pk_fma_f32 (tools/ubench_fma64.hip), so this is the fp32 loop's instruction stream with
    half the multiply-adds per instruction.  No int16 window (its conversions would come on top)."""

    def __init__(self, R, S, CT, CF, padded, w16=False):
        Variant.__init__(self, R, 2 * S, CT, CF, padded, False)   # register homes of the fp32 scheme with 2S steps
        self.S = S                                                # steps per bank here
        self.vbase = {10: 56, 5: 64}[R]   # (R = 10: the top of the 64 VGPRs of 8 waves per SIMD, v0-v55 left in one piece)
        # Round 5: an int16 LDS window here too (wide windows of quality 9 / 10 decimators: twice the periods per tile).
        # Single-channel lanes pay nothing for it -- ds_read_i16 sign-extends, v_cvt_f64_i32 replaces v_cvt_f64_f32 --,
        # channel pairs two instructions per frame (the dword's halves sign-extended before they are widened).
        self.w16 = w16
        self.eb = 2 if w16 else 4
        self.name = "A64_R%d_S%d_CT%d_CF%d_P%d_W%d" % (R, S, CT, CF, int(padded), int(w16))

    # step k (0 .. 2S-1) of a trip: four VGPRs, x as a double in [q, q+1], y in [q+2, q+3]; the raw floats arrive
    # in q (x) and q+2 (y)
    def quad(self, k):
        return self.vbase + 4 * k

    def vgprs(self):
        return list(range(self.vbase, self.vbase + 8 * self.S))

    def tap_pair(self, which, t):
        """first SGPR of the t-th double of the bank"""
        r = tap_reg(self.banks[which], 2 * t)
        assert r % 2 == 0 and tap_reg(self.banks[which], 2 * t + 1) == r + 1
        return r

    def fma_bank(self, which, lo, hi):
        out = []
        for u in range(self.S):
            k = u if which == "A" else self.S + u
            q = self.quad(k)
            for i in range(lo, hi):
                r = self.tap_pair(which, u * self.R + i)
                out.append("v_fma_f64 %%[a%dx], s[%d:%d], v[%d:%d], %%[a%dx]" % (i, r, r + 1, q, q + 1, i))
                out.append("v_fma_f64 %%[a%dy], s[%d:%d], v[%d:%d], %%[a%dy]" % (i, r, r + 1, q + 2, q + 3, i))
        return out

    def tap_loads(self, which, byte_off):
        out, t = [], 0
        for f, m in self.banks[which]:
            out.append("s_load_dword%s s[%d:%d], %%[rows], %%[off] offset:0x%x"
                       % ("x%d" % m if m > 1 else "", f, f + m - 1, byte_off + 4 * t))
            t += m
        return out

    def sample_reads(self, which, first_step):
        out = []
        for u in range(self.S):
            k = u if which == "A" else self.S + u
            o = (first_step + u) * self.CF * self.eb
            q = self.quad(k)
            if self.CT == 2 and self.w16:
                out.append("ds_read_b32 v%d, %%[addr] offset:%d" % (q + 1, o))   # both int16 samples of the frame
            elif self.CT == 2:
                # both channels of the frame: x lands in q, y in q+1 and moves to its own pair when widened
                out.append("ds_read_b64 v[%d:%d], %%[addr] offset:%d" % (q, q + 1, o))
            elif self.w16:
                out += ["ds_read_i16 v%d, %%[addr] offset:%d" % (q, o), "ds_read_i16 v%d, %%[addr2] offset:%d" % (q + 2, o)]
            else:
                out += ["ds_read_b32 v%d, %%[addr] offset:%d" % (q, o), "ds_read_b32 v%d, %%[addr2] offset:%d" % (q + 2, o)]
        return out

    def converts(self, which):
        out = []
        for u in range(self.S):
            k = u if which == "A" else self.S + u
            q = self.quad(k)
            if self.CT == 2 and self.w16:
                # the dword in q+1: y = its upper half (arithmetic shift), x = its lower half (signed bit field), then widened
                out += ["v_ashrrev_i32_e32 v%d, 16, v%d" % (q + 2, q + 1), "v_bfe_i32 v%d, v%d, 0, 16" % (q, q + 1),
                        "v_cvt_f64_i32_e32 v[%d:%d], v%d" % (q + 2, q + 3, q + 2), "v_cvt_f64_i32_e32 v[%d:%d], v%d" % (q, q + 1, q)]
            elif self.CT == 2:
                out += ["v_cvt_f64_f32_e32 v[%d:%d], v%d" % (q + 2, q + 3, q + 1), "v_cvt_f64_f32_e32 v[%d:%d], v%d" % (q, q + 1, q)]
            elif self.w16:
                out += ["v_cvt_f64_i32_e32 v[%d:%d], v%d" % (q, q + 1, q), "v_cvt_f64_i32_e32 v[%d:%d], v%d" % (q + 2, q + 3, q + 2)]
            else:
                out += ["v_cvt_f64_f32_e32 v[%d:%d], v%d" % (q, q + 1, q), "v_cvt_f64_f32_e32 v[%d:%d], v%d" % (q + 2, q + 3, q + 2)]
        return out

    def loop(self, label, cnt, lo, hi):
        S, R = self.S, self.R
        bank_bytes = 8 * S * R
        adv = 2 * S * self.CF * self.eb
        body = ["s_sub_u32 %%[%s], %%[%s], 1" % (cnt, cnt), "s_cbranch_scc1 %d1f" % label, "%d0:" % label]
        body += ["s_waitcnt lgkmcnt(0)"] + self.converts("A") + self.tap_loads("B", bank_bytes) + self.sample_reads("B", S)
        body += self.fma_bank("A", lo, hi)
        body += ["s_waitcnt lgkmcnt(0)"] + self.converts("B")
        if self.padded:
            body += ["s_sub_u32 %[wrap], %[wrap], 1", "s_cmp_eq_u32 %[wrap], 0",
                     "s_cselect_b32 %[tmp], %[advpad], " + str(adv), "s_cselect_b32 %[wrap], %[wrapstep], %[wrap]",
                     "v_add_u32 %[addr], %[tmp], %[addr]"]
            if self.CT == 1:
                body += ["v_add_u32 %[addr2], %[tmp], %[addr2]"]
        else:
            body += ["v_add_u32 %%[addr], %d, %%[addr]" % adv]
            if self.CT == 1:
                body += ["v_add_u32 %%[addr2], %d, %%[addr2]" % adv]
        body += self.tap_loads("A", 2 * bank_bytes) + self.sample_reads("A", 0)
        body += self.fma_bank("B", lo, hi)
        body += ["s_add_u32 %%[off], %%[off], 0x%x" % (2 * bank_bytes), "s_sub_u32 %%[%s], %%[%s], 1" % (cnt, cnt),
                 "s_cbranch_scc0 %d0b" % label, "%d1:" % label]
        return body

    def function(self):
        R = self.R
        asm = "\n".join('      "%s\\n"' % l for l in self.lines())
        outs = []
        for i in range(R):
            outs += ['[a%dx] "+v"(acc[%d][0])' % (i, i), '[a%dy] "+v"(acc[%d][1])' % (i, i)]
        outs.append('[addr] "+v"(addr)')
        if self.CT == 1:
            outs.append('[addr2] "+v"(addr2)')
        outs += ['[off] "+s"(off)', '[main] "+s"(main)']
        if R == 10:
            outs += ['[head] "+s"(head)', '[tail] "+s"(tail)']
        ins = ['[rows] "s"(rows_g)']
        if self.padded:
            outs += ['[wrap] "+s"(to_wrap)', '[tmp] "=&s"(tmp)']
            ins += ['[advpad] "s"(adv_pad)', '[wrapstep] "s"(wrap_step)']
        clob = ['"s%d"' % r for r in bank_regs(self.banks["A"]) + bank_regs(self.banks["B"])] + ['"v%d"' % r for r in self.vgprs()]
        return '''template <>
struct FirLoopAsm64<%d, %d, %d, %s, %s> {
  static constexpr bool available = true;
  static constexpr int steps_per_bank = %d;
  // as FirLoopAsm::run; acc[i][0 / 1]: the two halves of row i (channel pair, or the lane's two periods) in fp64;
  // rows_g: the group's tap rows as doubles
  static __device__ __forceinline__ void run(double (&acc)[%d][2], const double *rows_g, uint32_t addr, uint32_t addr2,
                                             uint32_t head, uint32_t main, uint32_t tail, uint32_t to_wrap,
                                             uint32_t wrap_step, uint32_t adv_pad) {
    uint32_t off = 0, tmp;
    (void)tmp; (void)addr2; (void)head; (void)tail; (void)to_wrap; (void)wrap_step; (void)adv_pad;
    asm volatile(
%s
      : %s
      : %s
      : %s, "scc", "memory");
  }
};
''' % (R, self.CT, self.CF, "true" if self.padded else "false", "true" if self.w16 else "false", self.S, R, asm,
       ", ".join(outs), ", ".join(ins), ", ".join(clob))

    def lines(self):
        # the prologue leaves bank A in flight, like the fp32 loop; its conversions run behind the first wait
        pro = self.tap_loads("A", 0) + self.sample_reads("A", 0)
        if self.R == 10:
            body = self.loop(1, "head", 0, 5) + self.loop(2, "main", 0, 10) + self.loop(3, "tail", 5, 10)
        else:
            body = self.loop(2, "main", 0, self.R)
        return pro + body + ["s_waitcnt lgkmcnt(0)"]


class VariantPP(Variant):
    """Phase pairs for single-channel lanes (round 4): a lane owns ONE period and 2R phases of it -- the two halves
    of a packed FMA are two PHASES of one sample (tap pair from an aligned SGPR pair, the sample broadcast:
    op_sel_hi picks its half for both lanes of the pack) where the CT = 1 variants above give the halves to two
    PERIODS of one tap.  A tile is then 64 periods instead of 128: half the LDS window for the same lanes -- what the
    wide windows of down-sampling ratios need (mono 48k -> 11.025k: 640 frames per period) -- at twice the tap bytes
    per FMA.  Same SGPR homes as the fp32 scheme with 2S steps; a bank is S steps of 2R taps; the bank's samples sit
    in the low halves of aligned VGPR pairs (one pair per step).  W16: an int16 window, converted behind the wait."""

    def __init__(self, R, S, CF, padded, w16):
        # CF: samples per frame = channels (1, 2, 3): a lane is (period, channel), its samples CF elements apart
        Variant.__init__(self, R, 2 * S, 1, CF, padded, w16)
        self.S = S
        self.vbase = {10: 56, 5: 64}[R]
        self.name = "PP_R%d_S%d_CF%d_P%d_W%d" % (R, S, CF, int(padded), int(w16))

    def reg(self, k):           # step k (0 .. 2S-1) of a trip: the sample in the low half of v[reg : reg+1]
        return self.vbase + 2 * k

    def vgprs(self):
        return list(range(self.vbase, self.vbase + 4 * self.S))

    def fma_bank(self, which, lo, hi):
        out = []
        for u in range(self.S):
            k = u if which == "A" else self.S + u
            x = self.reg(k)
            for i in range(lo, hi):
                r = tap_reg(self.banks[which], (u * self.R + i) * 2)
                assert r % 2 == 0
                out.append("v_pk_fma_f32 %%[a%d], s[%d:%d], v[%d:%d], %%[a%d] op_sel:[0,0,0] op_sel_hi:[1,0,1]" % (i, r, r + 1, x, x + 1, i))
        return out

    def sample_reads(self, which, first_step):
        out = []
        for u in range(self.S):
            k = u if which == "A" else self.S + u
            o = (first_step + u) * self.CF * self.eb
            out.append("ds_read_%s v%d, %%[addr] offset:%d" % ("i16" if self.w16 else "b32", self.reg(k), o))
        return out

    def converts(self, which):
        if not self.w16:
            return []
        return ["v_cvt_f32_i32_e32 v%d, v%d" % (self.reg(k), self.reg(k)) for k in
                ([u for u in range(self.S)] if which == "A" else [self.S + u for u in range(self.S)])]

    def loop(self, label, cnt, lo, hi):
        S, R = self.S, self.R
        bank_bytes = 4 * S * 2 * R
        adv = 2 * S * self.CF * self.eb
        body = ["s_sub_u32 %%[%s], %%[%s], 1" % (cnt, cnt), "s_cbranch_scc1 %d1f" % label, "%d0:" % label]
        body += ["s_waitcnt lgkmcnt(0)"] + self.tap_loads("B", bank_bytes) + self.sample_reads("B", S)
        body += self.converts("A") + self.fma_bank("A", lo, hi)
        body += ["s_waitcnt lgkmcnt(0)"]
        if self.padded:
            body += ["s_sub_u32 %[wrap], %[wrap], 1", "s_cmp_eq_u32 %[wrap], 0",
                     "s_cselect_b32 %[tmp], %[advpad], " + str(adv), "s_cselect_b32 %[wrap], %[wrapstep], %[wrap]",
                     "v_add_u32 %[addr], %[tmp], %[addr]"]
        else:
            body += ["v_add_u32 %%[addr], %d, %%[addr]" % adv]
        body += self.tap_loads("A", 2 * bank_bytes) + self.sample_reads("A", 0)
        body += self.converts("B") + self.fma_bank("B", lo, hi)
        body += ["s_add_u32 %%[off], %%[off], 0x%x" % (2 * bank_bytes), "s_sub_u32 %%[%s], %%[%s], 1" % (cnt, cnt),
                 "s_cbranch_scc0 %d0b" % label, "%d1:" % label]
        return body

    def function(self):
        R = self.R
        asm = "\n".join('      "%s\\n"' % l for l in self.lines())
        outs = ['[a%d] "+v"(acc[%d])' % (i, i) for i in range(R)] + ['[addr] "+v"(addr)']
        outs += ['[off] "+s"(off)', '[main] "+s"(main)']
        if R == 10:
            outs += ['[head] "+s"(head)', '[tail] "+s"(tail)']
        ins = ['[rows] "s"(rows_g)']
        if self.padded:
            outs += ['[wrap] "+s"(to_wrap)', '[tmp] "=&s"(tmp)']
            ins += ['[advpad] "s"(adv_pad)', '[wrapstep] "s"(wrap_step)']
        clob = ['"s%d"' % r for r in bank_regs(self.banks["A"]) + bank_regs(self.banks["B"])] + ['"v%d"' % r for r in self.vgprs()]
        return '''template <>
struct FirLoopAsmPP<%d, %d, %s, %s> {
  static constexpr bool available = true;
  static constexpr int steps_per_bank = %d;
  // as FirLoopAsm::run; acc[i] = phases 2i and 2i + 1 of the lane's period; rows_g: [trip][step][2R] floats
  static __device__ __forceinline__ void run(f32x2 (&acc)[%d], const float *rows_g, uint32_t addr, uint32_t head, uint32_t main,
                                             uint32_t tail, uint32_t to_wrap, uint32_t wrap_step, uint32_t adv_pad) {
    uint32_t off = 0, tmp;
    (void)tmp; (void)head; (void)tail; (void)to_wrap; (void)wrap_step; (void)adv_pad;
    asm volatile(
%s
      : %s
      : %s
      : %s, "scc", "memory");
  }
};
''' % (R, self.CF, "true" if self.padded else "false", "true" if self.w16 else "false", self.S, R, asm,
       ", ".join(outs), ", ".join(ins), ", ".join(clob))

    def lines(self):
        pro = self.tap_loads("A", 0) + self.sample_reads("A", 0)
        if self.R == 10:
            body = self.loop(1, "head", 0, 5) + self.loop(2, "main", 0, 10) + self.loop(3, "tail", 5, 10)
        else:
            body = self.loop(2, "main", 0, self.R)
        return pro + body + ["s_waitcnt lgkmcnt(0)"]


def variants_pp():
    out = []
    for CF in (1, 2, 3):
        for w16 in (False, True):
            for padded in (False, True):
                out.append(VariantPP(10, 1, CF, padded, w16))
            out.append(VariantPP(5, R5_STEPS // 2, CF, False, w16))
    return out


def variants64():
    out = []
    for w16 in (False, True):                                                              # (round 5: int16 window)
        for CT, CF in ((2, 2), (2, 4), (2, 6), (2, 8), (1, 1), (1, 3), (1, 5), (1, 7)):   # (round 5: odd frames too)
            for padded in (False, True):
                out.append(Variant64(10, 1, CT, CF, padded, w16))
            out.append(Variant64(5, R5_STEPS // 2, CT, CF, False, w16))
    return out


def variants():
    out = []
    for w16 in (False, True):
        # (1, 3), (1, 5), (1, 7): three-, five- and seven-channel frames on single-channel lanes (round 5): until then
        # those layouts ran the C++ loop, without int16 window or tap-range shares (three channels: an int16 window
        # and shares only through their phase-pair plans)
        # (2, 10), (2, 12), (2, 16): frames of 10, 12 and 16 channels as 5, 6, 8 channel pairs (late in round 5,
        # kernels_period_frames.hip): until then those ran the C++ loop -- no int16 window, no tap-range shares
        for CT, CF in ((2, 2), (2, 4), (2, 6), (2, 8), (1, 1), (1, 3), (1, 5), (1, 7), (2, 10), (2, 12), (2, 16)):
            for padded in (False, True):
                out.append(Variant(10, 2, CT, CF, padded, w16))
            out.append(Variant(5, R5_STEPS, CT, CF, False, w16))
    return out


R5_STEPS = 6

HEAD = '''// csrc/fir_loop_asm.inc -- GENERATED by csrc/gen_fir_loop.py; do not edit (see that file for the why and the how).
// The FIR inner loop of the period kernel as gfx950 ISA, one specialisation per (phases per wave, channels
// per lane, floats per frame, padded window, int16 window).  Included by kernels_period.hip only.

// primary template: no ISA variant for this layout -> fir_group runs its C++ loop
template <int R, int CT, int CF, bool PADDED, bool W16>
struct FirLoopAsm {
  static constexpr bool available = false;
};

'''


HEAD64 = '''
// ---- fp64 accumulator (round 4; gen_fir_loop.py, Variant64) ----
template <int R, int CT, int CF, bool PADDED, bool W16 = false>
struct FirLoopAsm64 {
  static constexpr bool available = false;
};

'''


HEADPP = '''
// ---- phase pairs: lane = (period, channel), two phases per packed FMA (round 4; gen_fir_loop.py, VariantPP) ----
template <int R, int CF, bool PADDED, bool W16>
struct FirLoopAsmPP {
  static constexpr bool available = false;
};

'''


def main():
    src = (HEAD + "\n".join(v.function() for v in variants()) + HEAD64 + "\n".join(v.function() for v in variants64()) +
           HEADPP + "\n".join(v.function() for v in variants_pp()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fir_loop_asm.inc")
    open(path, "w").write(src)
    print("wrote %s: %d + %d + %d variants, %d lines" % (path, len(variants()), len(variants64()), len(variants_pp()), src.count("\n")))


if __name__ == "__main__":
    main()
