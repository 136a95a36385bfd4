#!/usr/bin/env python3
"""Average issue cost of the VALU instructions of a kernel, from its ISA text and the per-instruction
costs measured by tools/valu_rates.hip (profiles/r02_valu_rates.json).

gfx950 issues a wave64 VALU instruction in ~2.5 cycles when it is one of the simple two-operand ops with
register operands only (v_add/sub/mul_f32, v_add/sub_u32, v_and/or/xor, v_lshrrev, v_mov) and in ~4.2
cycles otherwise (every VOP3-only op, conversions, compares, DPP, packed f32, anything that reads an
SGPR or a literal); v_cndmask_b32 in its VOP2 form (implicit VCC) measured ~23 cycles.  A counter gives
the NUMBER of VALU instructions a launch executes; this script gives the STATIC mix of the kernel's code
(every instruction of the function counted once), i.e. an estimate of the dynamic average cost that is
exact only if all blocks ran equally often.  bench.py prices the VALU unit with it.

usage: valu_mix.py <file.s> <kernel-name-substring> [...]   ->  JSON {kernel: {...}}"""
import json
import re
import sys

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32"}
COST_FAST, COST_SLOW, COST_WIDE, COST_CNDMASK_VCC = 2.5, 4.2, 4.4, 22.8


def classify(line):
    toks = line.replace(",", " ").split()
    op = toks[0]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    ops = toks[1:]
    if base == "v_cndmask_b32" and (op.endswith("_e32") or "vcc" in ops):
        return "cndmask_vcc", COST_CNDMASK_VCC
    if base.startswith("v_pk_") or base.endswith("_u64") or base.endswith("_b64") or base.endswith("_i64"):
        return "wide", COST_WIDE
    scalar_src = any(re.match(r"^(s\d+|s\[\d+:\d+\]|vcc|exec|m0|0x[0-9a-f]+)$", o) for o in ops[1:])
    if base in FAST and not scalar_src and not op.endswith("_dpp"):
        return "fast", COST_FAST
    return "slow", COST_SLOW


def kernel_mix(text, name_part):
    """instructions between '<mangled name containing name_part>:' and its s_endpgm"""
    inside, counts, cycles, salu = False, {}, 0.0, 0
    for raw in text.splitlines():
        if not inside:
            if re.match(r"^[A-Za-z_][\w$.]*:", raw) and name_part in raw.split(":")[0]:
                inside = True
            continue
        line = raw.split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        if line.startswith("v_"):
            k, c = classify(line)
            counts[k] = counts.get(k, 0) + 1
            cycles += c
        elif line.startswith("s_") and not re.match(r"^s_(waitcnt|nop|barrier|endpgm|load|buffer_load|branch|cbranch)", line):
            salu += 1
        if line.startswith("s_endpgm"):
            break
    n = sum(counts.values())
    return {"static_valu": n, "classes": counts, "avg_cycles_per_valu": (cycles / n) if n else None,
            "static_salu": salu}


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    print(json.dumps({k: kernel_mix(text, k) for k in sys.argv[2:]}, indent=1))
