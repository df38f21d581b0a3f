#!/usr/bin/env python3
"""ALU ceiling of the GF(2^192) butterfly kernels from their ISA and the measured per-class VALU rates.

For each kernel: the VALU instructions one butterfly executes, by issue class, taken from the compiler's assembly of the hot loop
(hipcc -S of libiop_amd/csrc/fft_add.hip; instructions inside the comb product's inline-asm block are replaced by the block's
DYNAMIC count — one of its sixteen window blocks runs per window, so a static count would be 16x too high), priced with the
per-class issue cost measured by tools/ubench/valu_rates (profiles/r05_valu_rates.txt, 8 waves per SIMD):

    ceiling [products/s] = SIMDs x clock / sum_class(count_class x cycles_class) x 64 lanes

bench.py divides the rate it measures live (field products per launch / HIP-event time) by this ceiling: roofline.alu_ceiling_frac.
Writes profiles/r06_alu_model.json.  Needs hipcc (cross-compiles without a GPU)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, CLOCK = 1024, 2.4e9
RATES_FILE = "r05_valu_rates.txt"          # tools/ubench/valu_rates on the MI355X (the newest collection)

# mnemonic -> micro-benchmark line it is priced by (profiles/r05_valu_rates.txt); anything else that starts with v_ is priced as the
# slow class and listed under "unmeasured" in the output
CLASS_OF = {
    "v_xor_b32": "v_xor_b32 (VOP2)", "v_and_b32": "v_and_b32 (VOP2)", "v_or_b32": "v_and_b32 (VOP2)", "v_not_b32": "v_and_b32 (VOP2)",
    "v_mov_b32": "v_and_b32 (VOP2)", "v_add_u32": "v_add_u32", "v_sub_u32": "v_add_u32", "v_subrev_u32": "v_add_u32",
    "v_bitop3_b32": "v_bitop3_b32 (3 vgpr)", "v_lshlrev_b32": "v_lshlrev_b32 (VOP2)", "v_lshrrev_b32": "v_lshlrev_b32 (VOP2)",
    "v_ashrrev_i32": "v_lshlrev_b32 (VOP2)", "v_alignbit_b32": "v_alignbit_b32", "v_bfe_u32": "v_bfe_i32", "v_bfe_i32": "v_bfe_i32",
    "v_perm_b32": "v_perm_b32", "v_mul_u32_u24": "v_mul_u32_u24", "v_mul_lo_u32": "v_mul_lo_u32", "v_mad_u64_u32": "v_mad_u64_u32",
    "v_lshl_add_u64": "v_lshl_add_u64 (64-bit add)", "v_and_or_b32": "v_and_or_b32", "v_lshl_or_b32": "v_and_or_b32", "v_add3_u32": "v_add3_u32",
    "v_lshl_add_u32": "v_add3_u32", "v_lshrrev_b64": "v_lshrrev_b64", "v_lshlrev_b64": "v_lshrrev_b64",
}
SLOW_DEFAULT = "v_alignbit_b32"


def measured_rates():
    rates = {}
    for line in open(os.path.join(ROOT, "profiles", RATES_FILE)):
        m = re.match(r"(.+?)\s+[\d.]+ ms\s+([\d.]+) T lane-ops/s", line)
        if m:
            rates[m.group(1).strip()] = float(m.group(2)) * 1e12
    return rates


def cycles_of(rate):
    return SIMDS * 64 * CLOCK / rate


def assembly(source):
    out = "/tmp/alu_model_%s.s" % os.path.basename(source)
    csrc = os.path.join(ROOT, "libiop_amd", "csrc")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-I" + os.path.join(csrc, "include"),
                           "-mllvm", "-pragma-unroll-threshold=1000000", "--cuda-device-only", "-S", os.path.join(csrc, source), "-o", out],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def kernel_body(asm, mangled_prefix):
    m = re.search(r"^(%s\w*):.*?\n(.*?)\n\s*s_endpgm" % re.escape(mangled_prefix), asm, re.S | re.M)
    if not m:
        raise SystemExit("kernel %s not found" % mangled_prefix)
    return m.group(1), m.group(2).split("\n")


def hot_loop(lines, want_asm, mads=(250, 10 ** 9)):
    """The butterfly loop: the shortest loop (label ... backward branch to it) that contains the field product — the inline-asm block of
    the comb product (want_asm) or a number of v_mad_u64_u32 in `mads`: >= 250 for the general product (288), 90-120 for the one-word
    numerator product of the last level (6 word products = 96), 130-200 for the two-word one (9 word products = 144)."""
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] < i:
                loops.append((labels[tgt], i))
    best = None
    for a, b in loops:
        body = lines[a:b + 1]
        if want_asm and not any("#ASMSTART" in x for x in body):
            continue
        if not want_asm and not (mads[0] <= sum(1 for x in body if "v_mad_u64_u32" in x) < mads[1]):
            continue
        if best is None or b - a < best[0]:
            best = (b - a, a, b)
    if best is None:
        raise SystemExit("no loop found")
    return lines[best[1]:best[2] + 1]


def classify(body, rates):
    counts, unmeasured, in_asm = {}, {}, False
    sgpr_operand_ops = [0]
    other = {"salu": 0, "lds": 0, "vmem": 0, "branch": 0}
    for l in body:
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        if in_asm:
            continue
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m:
            continue
        op = m.group(1)
        base = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", op)
        if base.startswith("v_"):
            if base == "v_bitop3_b32" and re.search(r",\s*s\d+|,\s*0x", l):
                # Back to back, an op with an SGPR operand issues at the slow rate (tools/ubench/valu_rates: 37.8 T against 61.9 T).
                # Inside the general product it does not: moving every mask into VGPRs (no SGPR operand left, checked in the ISA) changed
                # the measured product rate by -1.5 % (tools/ubench/comb_rates "general": 4.45e10 -> 4.38e10/s).  So these are priced at
                # the three-VGPR rate, which makes the ceiling higher (the reported fraction lower) than the pessimistic reading would.
                cls = "v_bitop3_b32 (3 vgpr)"
                sgpr_operand_ops[0] += 1
            elif base in CLASS_OF:
                cls = CLASS_OF[base]
            else:
                cls = SLOW_DEFAULT
                unmeasured[base] = unmeasured.get(base, 0) + 1
            counts[cls] = counts.get(cls, 0) + 1
        elif base.startswith("s_cbranch") or base in ("s_branch", "s_setpc_b64"):
            other["branch"] += 1
        elif base.startswith("s_"):
            other["salu"] += 1
        elif base.startswith("ds_"):
            other["lds"] += 1
        elif base.startswith(("global_", "buffer_", "flat_", "scratch_")):
            other["vmem"] += 1
    cycles = sum(n * cycles_of(rates[c]) for c, n in counts.items())
    other["valu_with_sgpr_operand_priced_fast"] = sgpr_operand_ops[0]
    return counts, other, unmeasured, cycles


def comb_dynamic(rates):
    """Expected instruction counts of one comb_clmul_192_uniform call (tools/gen_comb_asm.py), window nibbles uniform over 0..15."""
    fast, slow = "v_xor_b32 (VOP2)", "v_alignbit_b32"
    build = {slow: 21, fast: 14}                      # 2a, 4a, 8a (7 shift-class ops each), 3a and 12a (7 XORs each)
    acc_init = {fast: 12}
    # a window: nibble 0 nothing; 1: 6 XORs (the top word of a is zero); 2,4,8,3,12: 7 XORs; the nine others 7 three-input XORs
    per_window = (0 + 6 + 5 * 7) / 16.0
    per_window3 = 9 * 7 / 16.0
    windows = {fast: 48 * per_window, "v_bitop3_b32 (3 vgpr)": 48 * per_window3}
    shifts = {slow: 7 * 12}
    total = {}
    for part in (build, acc_init, windows, shifts):
        for k, v in part.items():
            total[k] = total.get(k, 0) + v
    cycles = sum(n * cycles_of(rates[c]) for c, n in total.items())
    salu = 2 + 12 + 1 + 56 * 5 + 8 * 3          # getpc + table bases + field descriptor, 56 dispatches of 5, 8 round ends
    return {"valu_by_class": total, "valu": sum(total.values()), "salu": salu, "taken_branches": 48 + 8 + 8, "cycles_per_wave_product": cycles}


def main():
    rates = measured_rates()
    asm = assembly("fft_add.hip")
    out = {"source": "tools/alu_model.py", "clock_hz": CLOCK, "simds": SIMDS,
           "class_cycles_per_wave_instruction": {k: round(cycles_of(v), 3) for k, v in rates.items()},
           "rates_file": "profiles/" + RATES_FILE, "kernels": {}}
    comb = comb_dynamic(rates)
    for name, prefix, want_asm in (("k_bfly_upper", "_ZN4iopx12k_bfly_upperILb0ELb1EEE", True), ("k_bfly_edge", "_ZN4iopx11k_bfly_edgeILb0ELb0EEE", False)):
        sym, lines = kernel_body(asm, prefix)
        body = hot_loop(lines, want_asm)
        counts, other, unmeasured, cycles = classify(body, rates)
        entry = {"symbol": sym, "unit": "butterfly (one field product)", "loop_valu_by_class_outside_asm": counts, "loop_other": other,
                 "unmeasured_opcodes_priced_as_slow": unmeasured, "loop_cycles_outside_asm": round(cycles, 1)}
        total_cycles = cycles
        if want_asm:
            entry["comb_product_dynamic"] = {k: (round(v, 1) if isinstance(v, float) else v) for k, v in comb.items()}
            total_cycles += comb["cycles_per_wave_product"]
        if name == "k_bfly_edge":
            # over the standard basis (the prover's domains) the two lowest of the pass's six levels take the one- and two-word numerator
            # products; the figure bench.py divides by is the mean over the six levels
            levels = {"general (pair bits 5..2)": round(cycles, 1)}
            mix = 4 * cycles
            for label, rng in (("one-word numerators (pair bit 0)", (90, 121)), ("two-word numerators (pair bit 1)", (130, 200))):
                c2, o2, u2, cyc2 = classify(hot_loop(lines, False, rng), rates)
                levels[label] = round(cyc2, 1)
                entry["loop_valu_by_class " + label] = c2
                mix += cyc2
            entry["cycles_per_wave_butterfly_by_level"] = levels
            entry["general_product_ceiling_products_per_s"] = SIMDS * CLOCK / cycles * 64
            total_cycles = mix / 6.0
            entry["unit"] = "butterfly (mean over the pass's six levels: four general products, one one-word and one two-word numerator product)"
        entry["cycles_per_wave_butterfly"] = round(total_cycles, 1)
        entry["alu_ceiling_products_per_s"] = SIMDS * CLOCK / total_cycles * 64
        out["kernels"][name] = entry
        print(name, "cycles per wave-butterfly %.0f" % total_cycles, "ceiling %.3e products/s" % entry["alu_ceiling_products_per_s"], "unmeasured:", unmeasured)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r06_alu_model.json"), "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
