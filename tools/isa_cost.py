#!/usr/bin/env python3
"""isa_cost.py file.s kernel-substring [min_instrs] -- per-basic-block VALU issue-cost estimate for a gfx950 kernel.

Applies the per-instruction issue costs measured by tools/valu_rates.hip (cycles per wave64 instruction per SIMD) to
the compiler's assembly (hipcc -S --cuda-device-only), block by block, so that a rewrite of the DP inner loops can be
judged before a GPU run.  Not part of the product."""
import re
import sys
from collections import Counter

CHEAP = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_sub_u32", "v_add_u32", "v_subrev_u32",
         "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32",
         "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_not_b32"}


def cost(line):
    t = line.split()
    op = t[0]
    base = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", op)
    ops = " ".join(t[1:])
    has_s = bool(re.search(r"(?<![a-z_])s\[?\d+|vcc|exec", ops)) and not op.startswith("v_cmp") and \
        not op.startswith("v_cndmask") and not op.startswith("v_readlane") and not op.startswith("v_readfirstlane")
    if not op.startswith("v_"):
        return 0.0, "other"
    if "dpp" in op or "wave_sh" in ops or "row_sh" in ops:
        return 4.4, "dpp"
    if base.startswith("v_pk_"):
        return (5.1 if has_s else 4.8), "pk"
    if base in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32"):
        return 8.3, "trans"
    if base.startswith("v_readlane") or base.startswith("v_readfirstlane") or base.startswith("v_writelane"):
        return 4.4, "lane"
    if has_s:
        return 4.5, "sgpr-src"
    if base in CHEAP:
        return 2.8, "cheap"
    return 4.4, "full"


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], ["entry", start, []]
    blocks.append(cur)
    for i in range(start + 1, end + 1):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = [m.group(1), i, []]
            blocks.append(cur)
            continue
        t = l.split(";")[0].strip()
        if t and not t.startswith("."):
            cur[2].append(t)
    tot = 0.0
    for name, i, ins in blocks:
        cyc, cls, n = 0.0, Counter(), Counter()
        for t in ins:
            c, k = cost(t)
            cyc += c
            cls[k] += c
            n[k] += 1
        if len(ins) >= min_ins:
            print(f"{name:10s} line {i:6d} instrs {len(ins):4d} valu {sum(v for k, v in n.items() if k != 'other'):4d} "
                  f"est {cyc:6.0f} cyc  " + " ".join(f"{k}:{n[k]}" for k in sorted(n)))


if __name__ == "__main__":
    main()
