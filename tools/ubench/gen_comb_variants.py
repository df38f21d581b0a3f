#!/usr/bin/env python3
"""Micro-benchmark support: emits tools/ubench/comb_variants.h — candidate instruction schedules of the
wave-uniform GF(2)[x] 192x192 comb product (same contract as iopx/gfx950_comb.h: r[0..11] = a (x) c, c in SGPRs).
tools/ubench/comb_rates.hip times each of them at 1..4 waves per SIMD and checks them against the general product.

Variants (name -> what changes against the shipped schedule "v0"):
  v0   entry-major table v[TB + 7u + i], per window  s_bfe, s_mul, s_set_gpr_idx_idx  (3 SALU)          [shipped in round 2]
  v1   word-major table  v[TB + 16i + u]: the nibble IS the GPR index, per window s_bfe, s_set_gpr_idx_idx (2 SALU)
  v2   v1 + all 48 indices extracted into s[SB:SB+47] before the VALU stream, per window s_set_gpr_idx_idx (1 SALU)
  v3   v1 with 6-word table entries: the top 3 bits of a are split off (a = a_lo + a_hi x^189), their product with c is
       three masked XORs of uniform shifted copies of c; 96 table VGPRs, 288 window XORs instead of 336
  v4   v3 + v2's hoisted indices
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))


class Gen:
    def __init__(self, name, TB, layout, salu, words, SB=36):
        self.name, self.TB, self.layout, self.salu, self.W, self.SB = name, TB, layout, salu, words, SB
        self.lines = []
        self.nregs = 16 * words if layout == "word" else 16 * words
        self.build()

    def A(self, s):
        self.lines.append(s)

    def E(self, u, i):
        if self.layout == "entry":
            return "v%d" % (self.TB + self.W * u + i)
        return "v%d" % (self.TB + 16 * i + u)

    def shl1(self, dst, src):
        W = self.W
        for i in range(W - 1, 0, -1):
            self.A("v_alignbit_b32 %s, %s, %s, 31" % (self.E(dst, i), self.E(src, i), self.E(src, i - 1)))
        self.A("v_lshlrev_b32 %s, 1, %s" % (self.E(dst, 0), self.E(src, 0)))

    def xor(self, dst, x, y):
        for i in range(self.W):
            self.A("v_xor_b32 %s, %s, %s" % (self.E(dst, i), self.E(x, i), self.E(y, i)))

    def build(self):
        A, E, W = self.A, self.E, self.W
        A("s_mov_b32 %[sm0], m0")
        if self.salu == 1:
            # all 48 window indices up front: s[SB + 6 * o + k]
            for o in range(8):
                for k in range(6):
                    A("s_bfe_u32 s%d, %%[c%d], 0x%x" % (self.SB + 6 * o + k, k, (4 * o) | (4 << 16)))
                    if self.layout == "entry":
                        A("s_mul_i32 s%d, s%d, %d" % (self.SB + 6 * o + k, self.SB + 6 * o + k, W))
        for i in range(W):
            A("v_mov_b32 %s, 0" % E(0, i))
        if W == 7:
            for i in range(6):
                A("v_mov_b32 %s, %%[a%d]" % (E(1, i), i))
            A("v_mov_b32 %s, 0" % E(1, 6))
        else:
            for i in range(5):
                A("v_mov_b32 %s, %%[a%d]" % (E(1, i), i))
            A("v_and_b32 %s, 0x1fffffff, %%[a5]" % E(1, 5))
        self.shl1(2, 1); self.xor(3, 2, 1)
        self.shl1(4, 2); self.xor(5, 4, 1); self.xor(6, 4, 2); self.xor(7, 6, 1)
        self.shl1(8, 4)
        for v in range(1, 8):
            self.xor(8 + v, 8, v)
        for i in range(12):
            A("v_mov_b32 %%[r%d], 0" % i)
        A("s_mov_b32 %[st], 0")
        A("s_set_gpr_idx_on %[st], 2")
        for o in range(7, -1, -1):
            if o != 7:
                A("s_set_gpr_idx_idx 0")
                for i in range(11, 0, -1):
                    A("v_alignbit_b32 %%[r%d], %%[r%d], %%[r%d], 28" % (i, i, i - 1))
                A("v_lshlrev_b32 %[r0], 4, %[r0]")
            for k in range(6):
                if self.salu == 1:
                    A("s_set_gpr_idx_idx s%d" % (self.SB + 6 * o + k))
                else:
                    A("s_bfe_u32 %%[st], %%[c%d], 0x%x" % (k, (4 * o) | (4 << 16)))
                    if self.layout == "entry":
                        A("s_mul_i32 %%[st], %%[st], %d" % W)
                    A("s_set_gpr_idx_idx %[st]")
                for i in range(W):
                    A("v_xor_b32 %%[r%d], %%[r%d], %s" % (k + i, k + i, E(0, i)))
        A("s_set_gpr_idx_off")
        if W == 6:
            # a_hi = a[5] >> 29 (3 bits): r ^= sum_b bit_b(a_hi) * (c << (189 + b)); 189 = 5 * 32 + 29
            # the uniform shifted copies are built on the scalar unit in s[SB2 .. ] (7 words each), reusing SB when indices are not hoisted
            for b in range(3):
                sh = 29 + b
                A("v_bfe_i32 %s, %%[a5], %d, 1" % (E(0, 0), sh))      # mask in a scratch VGPR (the zero entry is dead now)
                # word j of (c << sh), j = 0..6: (c[j] << sh) | (c[j-1] >> (32 - sh))
                for j in range(7):
                    if j == 0:
                        A("s_lshl_b32 %%[st], %%[c0], %d" % sh)
                    elif j == 6:
                        A("s_lshr_b32 %%[st], %%[c5], %d" % (32 - sh))
                    else:
                        A("s_lshr_b32 %%[st], %%[c%d], %d" % (j - 1, 32 - sh))
                        A("s_lshl_b32 %%[st2], %%[c%d], %d" % (j, sh))
                        A("s_or_b32 %[st], %[st], %[st2]")
                    A("v_bitop3_b32 %%[r%d], %%[r%d], %s, %%[st] bitop3:0x78" % (5 + j, 5 + j, E(0, 0)))
        A("s_mov_b32 m0, %[sm0]")

    def emit(self):
        body = "\n".join('        "%s\\n\\t"' % l for l in self.lines)
        outs = ", ".join('[r%d] "=&v"(r[%d])' % (i, i) for i in range(12)) + ', [st] "=&s"(st), [st2] "=&s"(st2), [sm0] "=&s"(sm0)'
        ins = ", ".join('[a%d] "v"(a[%d])' % (i, i) for i in range(6)) + ", " + ", ".join('[c%d] "s"(c[%d])' % (i, i) for i in range(6))
        nv = 16 * self.W if self.layout == "entry" else 16 * (self.W - 1) + 16
        clob = ['"v%d"' % v for v in range(self.TB, self.TB + nv)] + ['"scc"']
        if self.salu == 1:
            clob += ['"s%d"' % s for s in range(self.SB, self.SB + 48)]
        nvalu = sum(1 for l in self.lines if l.startswith("v_"))
        nsalu = sum(1 for l in self.lines if l.startswith("s_"))
        return '''// %s: %d VALU, %d SALU, table v[%d:%d]
__device__ __forceinline__ void comb_%s(uint32_t (&r)[12], const uint32_t (&a)[6], const uint32_t (&c)[6])
{
    uint32_t st, st2, sm0;
    asm volatile(
%s
        : %s
        : %s
        : %s);
    (void)st; (void)st2; (void)sm0;
}
''' % (self.name, nvalu, nsalu, self.TB, self.TB + nv - 1, self.name, body, outs, ins, ", ".join(clob)), nvalu, nsalu


VARIANTS = [
    Gen("v0", 56, "entry", 3, 7),
    Gen("v1", 56, "word", 2, 7),
    Gen("v2", 56, "word", 1, 7),
    Gen("v3", 32, "word", 2, 6),
    Gen("v4", 32, "word", 1, 6),
    Gen("v3hi", 56, "word", 2, 6),     # v3 at the product's register window (3 waves): separates the occupancy effect
]

class GenJump:
    """Uniform-branch comb: the 4-bit window value is wave-uniform, so instead of selecting the table entry with GPR-index
    relative addressing (measured: every relative v_xor issues at the slow 4.16-cycle rate) the wave JUMPS to one of 16 code
    blocks with hard-coded registers.  All window XORs are then plain fast-class VALU ops; entries that are the XOR of two
    materialised entries are applied with one v_xor3 per word, so only a, 2a, 4a, 8a, 3a, 12a live in registers (35 VGPRs)."""
    BASIS = {1: None, 2: 0, 4: 1, 8: 2, 3: 3, 12: 4}      # entry -> slot in the VGPR window (a itself is the input operand)
    SPLIT = {5: (4, 1), 6: (4, 2), 7: (4, 3), 9: (8, 1), 10: (8, 2), 11: (8, 3), 13: (12, 1), 14: (12, 2), 15: (12, 3)}

    def __init__(self, name, TB, SB=36, mode=0):
        # mode 0: a block's XORs, then the dispatch of the next window; 1: the dispatch's scalar ops interleaved with the XORs (they do
        # not depend on them), the jump last; 2: the scalar ops first, the XORs, the jump
        self.name, self.TB, self.SB, self.mode = name, TB, SB, mode
        self.lines = []
        self.build()

    def A(self, s):
        self.lines.append(s)

    def E(self, u, i):
        """register (or operand) of word i of materialised entry u; None when that word is identically zero"""
        if u == 1:
            return None if i == 6 else "%%[a%d]" % i
        return "v%d" % (self.TB + 7 * self.BASIS[u] + i)

    def shl1(self, dst, src):
        for i in range(6, 0, -1):
            hi = self.E(src, i) or "0"
            if i == 6 and src == 1:
                self.A("v_lshrrev_b32 %s, 31, %s" % (self.E(dst, 6), self.E(src, 5)))
            else:
                self.A("v_alignbit_b32 %s, %s, %s, 31" % (self.E(dst, i), hi, self.E(src, i - 1)))
        self.A("v_lshlrev_b32 %s, 1, %s" % (self.E(dst, 0), self.E(src, 0)))

    def dispatch_ops(self, k):
        SB = self.SB
        return ["s_bfe_u32 s%d, %%[c%d], s%d" % (SB + 17, k, SB + 16),
                "s_lshl_b32 s%d, s%d, 7" % (SB + 17, SB + 17),
                "s_add_u32 s%d, s%d, s%d" % (SB, SB + 2 + 2 * k, SB + 17),
                "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 3 + 2 * k),
                "s_setpc_b64 s[%d:%d]" % (SB, SB + 1)]

    def dispatch(self, k):
        for l in self.dispatch_ops(k):
            self.A(l)

    def block(self, valu, k_next):
        """one window block: its XORs and the dispatch of table k_next, ordered by self.mode"""
        d = self.dispatch_ops(k_next)
        if self.mode == 0:
            seq = valu + d
        elif self.mode == 2:
            seq = d[:4] + valu + d[4:]
        else:
            seq, sal = [], d[:4]
            for i, v in enumerate(valu):
                if i < len(sal):
                    seq.append(sal[i])
                seq.append(v)
            seq += sal[len(valu):] + d[4:]
        for l in seq:
            self.A(l)

    def build(self):
        A, E, SB = self.A, self.E, self.SB
        L = ".Lcj%=_"
        self.shl1(2, 1); self.shl1(4, 2); self.shl1(8, 4)
        for i in range(7):
            if E(1, i):
                A("v_xor_b32 %s, %s, %s" % (E(3, i), E(2, i), E(1, i)))
            else:
                A("v_mov_b32 %s, %s" % (E(3, i), E(2, i)))
        for i in range(7):
            A("v_xor_b32 %s, %s, %s" % (E(12, i), E(8, i), E(4, i)))
        for i in range(12):
            A("v_mov_b32 %%[r%d], 0" % i)
        A("s_getpc_b64 s[%d:%d]" % (SB + 14, SB + 15))
        A(L + "anchor:")
        for k in range(6):
            A("s_add_u32 s%d, s%d, %st%d-%sanchor" % (SB + 2 + 2 * k, SB + 14, L, k, L))
            A("s_addc_u32 s%d, s%d, 0" % (SB + 3 + 2 * k, SB + 15))
        A("s_mov_b32 s%d, 0x4001c" % (SB + 16))              # field descriptor of s_bfe: offset 28, width 4
        self.dispatch(0)
        for k in range(6):
            A(".p2align 7")
            A(L + "t%d:" % k)
            for u in range(16):
                A(".p2align 7")
                valu = []
                if u:
                    if u in self.BASIS:
                        for i in range(7):
                            if E(u, i):
                                valu.append("v_xor_b32 %%[r%d], %%[r%d], %s" % (k + i, k + i, E(u, i)))
                    else:
                        x, y = self.SPLIT[u]
                        for i in range(7):
                            if E(y, i):
                                valu.append("v_bitop3_b32 %%[r%d], %%[r%d], %s, %s bitop3:0x96" % (k + i, k + i, E(x, i), E(y, i)))
                            else:
                                valu.append("v_xor_b32 %%[r%d], %%[r%d], %s" % (k + i, k + i, E(x, i)))
                if k < 5:
                    self.block(valu, k + 1)
                else:
                    for l in valu:
                        A(l)
                    A("s_branch %srend" % L)
        A(".p2align 7")
        A(L + "rend:")
        A("s_cmp_eq_u32 s%d, 0x40000" % (SB + 16))
        A("s_cbranch_scc1 %sexit" % L)
        A("s_sub_u32 s%d, s%d, 4" % (SB + 16, SB + 16))
        for i in range(11, 0, -1):
            A("v_alignbit_b32 %%[r%d], %%[r%d], %%[r%d], 28" % (i, i, i - 1))
        A("v_lshlrev_b32 %[r0], 4, %[r0]")
        self.dispatch(0)
        A(L + "exit:")

    def emit(self):
        body = "\n".join('        "%s\\n\\t"' % l for l in self.lines)
        outs = ", ".join('[r%d] "=&v"(r[%d])' % (i, i) for i in range(12))
        ins = ", ".join('[a%d] "v"(a[%d])' % (i, i) for i in range(6)) + ", " + ", ".join('[c%d] "s"(c[%d])' % (i, i) for i in range(6))
        clob = ['"v%d"' % v for v in range(self.TB, self.TB + 35)] + ['"scc"'] + ['"s%d"' % s for s in range(self.SB, self.SB + 18)]
        nvalu = sum(1 for l in self.lines if l.startswith("v_"))
        return """// %s: uniform-branch comb, table v[%d:%d], scratch s[%d:%d]
__device__ __forceinline__ void comb_%s(uint32_t (&r)[12], const uint32_t (&a)[6], const uint32_t (&c)[6])
{
    asm volatile(
%s
        : %s
        : %s
        : %s);
}
""" % (self.name, self.TB, self.TB + 34, self.SB, self.SB + 17, self.name, body, outs, ins, ", ".join(clob)), nvalu, 0


VARIANTS += [GenJump("j0", 40), GenJump("j1", 40, mode=1), GenJump("j2", 40, mode=2)]


if __name__ == "__main__":
    out = ["// GENERATED by tools/ubench/gen_comb_variants.py (micro-benchmark only)", "#pragma once", "#include <hip/hip_runtime.h>", "#include <stdint.h>", ""]
    for g in VARIANTS:
        txt, nv, ns = g.emit()
        out.append(txt)
        print(g.name, "VALU", nv, "SALU", ns)
    open(os.path.join(HERE, "comb_variants.h"), "w").write("\n".join(out))
