#!/usr/bin/env python3
"""Generates libiop_amd/csrc/include/iopx/gfx950_comb.h: the GF(2)[x] 192x192 -> 383-bit carry-less product by a
wave-uniform multiplier as ONE hand-scheduled gfx950 inline-asm block.

Round 3 schedule ("uniform-branch comb").  Left-to-right comb with 4-bit windows as before, but the window value — a nibble
of the wave-uniform c — no longer selects its table entry through S_SET_GPR_IDX relative addressing: the micro-benchmark
(tools/ubench/comb_rates.hip, profiles/r03_comb_rates.txt) showed that EVERY VALU op with a relative operand issues at the slow
4.16-cycle rate of shifts / multiplies instead of the 2.5 cycles of a plain v_xor, whatever the scalar work around it.  Instead the
wave JUMPS (s_setpc_b64 into a 16 x 128-byte block table per word offset) to code with hard-coded registers:
  * all 336 window XORs are plain fast-class ops;
  * an entry that is the XOR of two materialised entries is applied with one three-input XOR per word, so only
    2a, 4a, 8a, 3a, 12a are materialised next to a itself: 35 table VGPRs instead of 112, 35 build ops instead of 124;
  * each block carries the dispatch of the next window, its scalar ops between the block's XORs (one taken branch per window, 56 per
    product); the branch latency
    (~58 cycles) is hidden by the other wavefronts of the SIMD: the callers run 5-6 waves per SIMD, which the small table permits.
Measured (uniform products/s, chip-wide): 5.9e10 (round 2 schedule, 3 waves/SIMD) -> 8.8e10 at 6 waves/SIMD, 9.2e10 at 8.
"""
import os

TB = int(os.environ.get("IOPX_COMB_TB", "40"))     # first table VGPR: v[TB:TB+34]
SB = int(os.environ.get("IOPX_COMB_SB", "36"))     # scalar scratch: s[SB:SB+17]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "libiop_amd", "csrc", "include", "iopx", "gfx950_comb.h")

BASIS = {1: None, 2: 0, 4: 1, 8: 2, 3: 3, 12: 4}      # entry -> slot in the VGPR window (a itself is the input operand)
SPLIT = {5: (4, 1), 6: (4, 2), 7: (4, 3), 9: (8, 1), 10: (8, 2), 11: (8, 3), 13: (12, 1), 14: (12, 2), 15: (12, 3)}

def generate(TB, SB, two=False):
    """(asm body, clobber list) of the product with its table in v[TB:TB+34] and its scalar scratch in s[SB:SB+17] (two: s[SB:SB+26])"""
    lines = []
    A = lines.append
    L = ".Lcj%=_"


    def E(u, i):
        """register (or operand) of word i of materialised entry u; None when that word is identically zero"""
        if u == 1:
            return None if i == 6 else "%%[a%d]" % i
        return "v%d" % (TB + 7 * BASIS[u] + i)


    def shl1(dst, src):
        for i in range(6, 0, -1):
            if i == 6 and src == 1:
                A("v_lshrrev_b32 %s, 31, %s" % (E(dst, 6), E(src, 5)))
            else:
                A("v_alignbit_b32 %s, %s, %s, 31" % (E(dst, i), E(src, i), E(src, i - 1)))
        A("v_lshlrev_b32 %s, 1, %s" % (E(dst, 0), E(src, 0)))


    def dispatch_ops(k):
        """jump to block (nibble of c[k] selected by the field descriptor in s[SB+16]) of table k"""
        csrc = ("s%d" % (SB + 21 + k)) if two else ("%%[c%d]" % k)                 # two-twiddle form: the current half's twiddle sits in scratch SGPRs
        return ["s_bfe_u32 s%d, %s, s%d" % (SB + 17, csrc, SB + 16),
                "s_lshl_b32 s%d, s%d, 7" % (SB + 17, SB + 17),
                "s_add_u32 s%d, s%d, s%d" % (SB, SB + 2 + 2 * k, SB + 17),
                "s_addc_u32 s%d, s%d, 0" % (SB + 1, SB + 3 + 2 * k),
                "s_setpc_b64 s[%d:%d]" % (SB, SB + 1)]


    def dispatch(k):
        for l in dispatch_ops(k):
            A(l)


    def block(valu, k_next):
        """One window block: its XORs with the four scalar ops of the next window's dispatch between them (they depend on c only, and a
        wavefront's scalar op issues while its vector op occupies the SIMD), the jump last.  Against XORs-then-dispatch: +4 % products/s at
        6 waves per SIMD (tools/ubench/comb_rates: variants j0 / j1)."""
        d = dispatch_ops(k_next)
        sal = d[:4]
        for i, v in enumerate(valu):
            if i < len(sal):
                A(sal[i])
            A(v)
        for l in sal[len(valu):] + d[4:]:
            A(l)


    shl1(2, 1); shl1(4, 2); shl1(8, 4)
    for i in range(7):
        if E(1, i):
            A("v_xor_b32 %s, %s, %s" % (E(3, i), E(2, i), E(1, i)))
        else:
            A("v_mov_b32 %s, %s" % (E(3, i), E(2, i)))
    for i in range(7):
        A("v_xor_b32 %s, %s, %s" % (E(12, i), E(8, i), E(4, i)))
    for i in range(12):
        A("v_mov_b32 %%[r%d], 0" % i)
    # absolute addresses of the six block tables
    A("s_getpc_b64 s[%d:%d]" % (SB + 14, SB + 15))
    A(L + "anchor:")
    for k in range(6):
        A("s_add_u32 s%d, s%d, %st%d-%sanchor" % (SB + 2 + 2 * k, SB + 14, L, k, L))
        A("s_addc_u32 s%d, s%d, 0" % (SB + 3 + 2 * k, SB + 15))
    if two:
        # lanes 0..31 first (twiddle c), lanes 32..63 second (twiddle d): the table is built once for all lanes, the window loop runs per half
        A("s_mov_b64 s[%d:%d], exec" % (SB + 18, SB + 19))
        A("s_mov_b32 s%d, 0" % (SB + 20))
        for k in range(6):
            A("s_mov_b32 s%d, %%[c%d]" % (SB + 21 + k, k))
        A("s_mov_b32 exec_hi, 0")
    A("s_mov_b32 s%d, 0x4001c" % (SB + 16))              # s_bfe field descriptor: offset 28, width 4 (the top nibbles first)
    dispatch(0)
    for k in range(6):
        A(".p2align 7")
        A(L + "t%d:" % k)
        for u in range(16):
            A(".p2align 7")
            valu = []
            if u:
                if u in BASIS:
                    for i in range(7):
                        if E(u, i):
                            valu.append("v_xor_b32 %%[r%d], %%[r%d], %s" % (k + i, k + i, E(u, i)))
                else:
                    x, y = SPLIT[u]
                    for i in range(7):
                        if E(y, i):
                            valu.append("v_bitop3_b32 %%[r%d], %%[r%d], %s, %s bitop3:0x96" % (k + i, k + i, E(x, i), E(y, i)))
                        else:
                            valu.append("v_xor_b32 %%[r%d], %%[r%d], %s" % (k + i, k + i, E(x, i)))
            if k < 5:
                block(valu, k + 1)
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
    dispatch(0)
    A(L + "exit:")
    if two:
        A("s_cmp_eq_u32 s%d, 1" % (SB + 20))
        A("s_cbranch_scc1 %sdone" % L)
        A("s_mov_b32 s%d, 1" % (SB + 20))
        for k in range(6):
            A("s_mov_b32 s%d, %%[d%d]" % (SB + 21 + k, k))
        A("s_mov_b32 exec_lo, 0")
        A("s_mov_b32 exec_hi, s%d" % (SB + 19))
        A("s_mov_b32 s%d, 0x4001c" % (SB + 16))
        dispatch(0)
        A(L + "done:")
        A("s_mov_b64 exec, s[%d:%d]" % (SB + 18, SB + 19))

    body = "\n".join('        "%s\\n\\t"' % l for l in lines)
    clob = ", ".join(['"v%d"' % v for v in range(TB, TB + 35)] + ['"s%d"' % s for s in range(SB, SB + (27 if two else 18))] + ['"scc"'])
    return body, clob


body, clob = generate(TB, SB)
body2, clob2 = generate(TB, SB, two=True)
outs = ", ".join('[r%d] "=&v"(r[%d])' % (i, i) for i in range(12))
ins = ", ".join('[a%d] "v"(a[%d])' % (i, i) for i in range(6)) + ", " + ", ".join('[c%d] "s"(c[%d])' % (i, i) for i in range(6))
ins2 = ins + ", " + ", ".join('[d%d] "s"(d[%d])' % (i, i) for i in range(6))

hdr = '''// GENERATED by tools/gen_comb_asm.py — do not edit.
//
// r[0..11] = a[0..5] (x) c[0..5]: the 383-bit carry-less product of a per-lane 192-bit a and a WAVE-UNIFORM
// 192-bit c (c in SGPRs), as one gfx950 inline-asm block.  Left-to-right comb with 4-bit windows (Lopez-Dahab).
// The window value is wave-uniform, so the wave BRANCHES (s_setpc_b64 into a table of 16 x 128-byte code blocks per
// word offset) to the XORs of that table entry with hard-coded registers: GPR-index relative addressing, which the
// round-2 schedule used, makes every VALU op that has a relative operand issue at the slow (shift-class) rate on gfx950.
// Table: a (the input operand), 2a, 4a, 8a, 3a, 12a in v[%d:%d]; the other entries are applied as one three-input XOR per
// word.  Scalar scratch s[%d:%d].  One taken branch per window: run it at >= 5 waves per SIMD.
// The caller guarantees that c is identical in every lane (the branches are taken on c alone; EXEC is untouched).
// tests/emu models this function in plain C++.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void comb_clmul_192_uniform(uint32_t (&r)[12], const uint32_t (&a)[6], const uint32_t (&c)[6])
{
    asm volatile(
%s
        : %s
        : %s
        : %s);
}

// The product for a wavefront whose HALVES share a multiplier each (lanes 0..31: c, lanes 32..63: d, both in SGPRs): the table of multiples of a is
// built once for all 64 lanes, the window loop runs twice, under EXEC = the low half with c's windows and under EXEC = the high half with d's
// (modelled 155 + 2 x 1240 + 140 = 2775 cycles per wave against 3209 for the general product with per-lane multipliers).  Scalar scratch s[%d:%d].
// EXEC: saved on entry, narrowed to one half at a time (an AND with the half: lanes that were inactive on entry stay inactive), restored before the
// block ends — it holds the same value after the statement as before it, so it is NOT on the clobber list (the compiler, which keeps EXEC reserved,
// warns about any asm that names it there); tests/test_gpu_parity.py::test_half_wavefront_product_keeps_exec calls the product under a divergent
// branch with a ballot right after it.  Used where a block's butterflies fill half a wavefront: pair bit 3 of the four-polynomial last pass.
__device__ __forceinline__ void comb_clmul_192_halves(uint32_t (&r)[12], const uint32_t (&a)[6], const uint32_t (&c)[6], const uint32_t (&d)[6])
{
    asm volatile(
%s
        : %s
        : %s
        : %s);
}

// the CPU emulation of comb_clmul_192_halves runs one lane at a time and must be told which half the lane is in; on the GPU EXEC does that
__device__ __forceinline__ void iopx_set_emu_lane(int) {}

// A 64-bit load through the scalar unit: the address must be wave-uniform (the caller passes an index made uniform with
// readfirstlane).  The constant address space makes the compiler emit s_load instead of a per-lane global_load followed by
// v_readfirstlane: the twiddle of a comb product is needed in SGPRs anyway.  The data must not be written by the same kernel.
__device__ __forceinline__ uint64_t uniform_load64(const uint64_t *p)
{
    return *(const __attribute__((address_space(4))) uint64_t *)(uintptr_t)p;
}
''' % (TB, TB + 34, SB, SB + 17, body, outs, ins, clob, SB, SB + 26, body2, outs, ins2, clob2)
open(OUT, "w").write(hdr)
print("wrote", OUT)
