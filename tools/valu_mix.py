#!/usr/bin/env python3
"""Static share of the 2-cycle vector opcodes in a kernel's ISA (bench.py's roofline_valu.issue_weight).

profiles/r03/valu_rates.txt (tools/valu_rates.hip on the device): with four or more waves on a SIMD the opcodes in FAST below
issue one wave64 instruction per ~2 cycles, every other vector opcode one per ~4.  The weight printed is
(2 * fast + 4 * slow) / (4 * all) per barrier-separated region of the kernel, i.e. what fraction of the "4 cycles per instruction"
count the instructions really occupy.   usage: valu_mix.py k_fast.s kernel-name-substring
(k_fast.s: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -I include -I csrc csrc/k_fast.hip)"""
import re
import sys

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_bitop3_b32"}


def main():
    text = open(sys.argv[1]).read().split("\n")
    name = sys.argv[2]
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*:", l) and name in l)
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    regions, cur = [], [0, 0]
    for l in text[start:end]:
        if "s_barrier" in l:
            regions.append(cur)
            cur = [0, 0]
        m = re.match(r"\s+(v_[a-z0-9_]+)", l)
        if m:
            op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", m.group(1))
            cur[0 if op in FAST else 1] += 1
    regions.append(cur)
    tf = ts = 0
    for i, (f, s) in enumerate(regions):
        if f + s:
            print("region %d: fast %4d  slow %4d  weight %.3f" % (i, f, s, (2 * f + 4 * s) / (4.0 * (f + s))))
        tf, ts = tf + f, ts + s
    print("whole kernel: fast %d slow %d weight %.3f" % (tf, ts, (2 * tf + 4 * ts) / (4.0 * (tf + ts))))


if __name__ == "__main__":
    main()
