#!/usr/bin/env python3
"""Instruction mix of one kernel from hipcc's -save-temps assembly, loop by loop.

    tools/isa_mix.py FILE.s MANGLED_KERNEL_NAME [--json OUT]

Splits the kernel into basic blocks at labels, finds natural loops from backward branches (a branch whose target label
lies above it), and prints for every innermost loop — and for the straight-line remainder — the number of instructions by
class: fp64 VALU (v_add_f64, v_max_f64, v_cmp_*_f64 ...: quarter rate, 4 cycles per wave64 instruction), 32-bit VALU
(v_cndmask, integer, moves: 2 cycles), LDS (ds_*), vector memory, SALU, waits, branches.  The per-iteration counts
times the trip counts of the workload reconcile with the PMC totals (SQ_INSTS_VALU / SQ_INSTS_SALU per pair)."""
import collections
import json
import re
import sys


def classify(op):
    if op.startswith("v_"):
        if op in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"):
            return "valu_lane"
        if "_f64" in op or op.endswith("_u64") or op.endswith("_i64") or "b64" in op and op.startswith("v_lsh"):
            return "valu_f64" if "_f64" in op else "valu_64bit_int"
        return "valu_32"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op == "s_nop":
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def parse(path, kernel):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    instrs = []            # (index, label or None, op, operands)
    labels = {}
    for l in lines[start + 1:end]:
        s = l.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        instrs.append((parts[0], parts[1] if len(parts) > 1 else ""))
    return instrs, labels


def loops(instrs, labels):
    out = []
    for i, (op, args) in enumerate(instrs):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = args.strip()
            if tgt in labels and labels[tgt] <= i:
                out.append((labels[tgt], i))
    # innermost: loops that contain no other loop
    inner = [(a, b) for (a, b) in out if not any((c, d) != (a, b) and a <= c and d <= b for (c, d) in out)]
    return sorted(set(out)), sorted(set(inner))


def mix(instrs, a, b):
    c = collections.Counter()
    ops = collections.Counter()
    for op, _ in instrs[a:b + 1]:
        c[classify(op)] += 1
        ops[op] += 1
    return c, ops


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    instrs, labels = parse(path, kernel)
    allv, inner = loops(instrs, labels)
    report = {"kernel": kernel, "instructions": len(instrs), "loops": []}
    covered = set()
    for (a, b) in allv:
        c, ops = mix(instrs, a, b)
        is_inner = (a, b) in inner
        report["loops"].append({"first": a, "last": b, "innermost": is_inner, "n": b - a + 1, "classes": dict(c),
                                "top_ops": dict(ops.most_common(24))})
        if is_inner:
            covered.update(range(a, b + 1))
    rest = collections.Counter(classify(op) for i, (op, _) in enumerate(instrs) if i not in covered)
    report["outside_innermost_loops"] = dict(rest)
    if "--json" in sys.argv:
        json.dump(report, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    print("%s: %d instructions, %d loops (%d innermost)" % (kernel, len(instrs), len(allv), len(inner)))
    for L in report["loops"]:
        if L["n"] < 12:
            continue
        c = L["classes"]
        print("  loop [%5d..%5d] %s n=%4d  f64=%3d v32=%3d lane=%2d lds=%2d vmem=%2d salu=%3d wait=%2d br=%2d | %s" % (
            L["first"], L["last"], "inner" if L["innermost"] else "outer", L["n"], c.get("valu_f64", 0), c.get("valu_32", 0) + c.get("valu_64bit_int", 0),
            c.get("valu_lane", 0), c.get("lds", 0), c.get("vmem", 0), c.get("salu", 0) + c.get("smem", 0), c.get("wait", 0), c.get("branch", 0),
            " ".join("%s:%d" % kv for kv in list(L["top_ops"].items())[:10])))


if __name__ == "__main__":
    main()
