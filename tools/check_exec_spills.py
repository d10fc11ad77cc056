#!/usr/bin/env python3
"""Static check of hipcc's gfx950 assembly for one miscompile pattern found in round 4 (ROCm 7.2 / LLVM):

    .LBB_join:                                   ; join block of an `if` over part of the wavefront
        v_accvgpr_write_b32 a13, v11             ; <- register-allocator spill, inserted IN FRONT of the exec restore
        s_or_b64 exec, exec, s[30:31]            ; exec is widened only here

The spill (or reload) runs under the narrow mask of the `if`: the lanes that sat out the branch never save the value, and a reload under the
full mask later hands them whatever the AGPR / scratch slot held before.  In dd_hmm_kernel<11, 6, GBT> that was the LDS address of a lane's
haplotype constants, lost for lane 63 from a workgroup's second item on (wrong likelihoods for reads at the right end of 639..702-bp haplotypes;
tests/test_gpu_persistent_rounds.py is the run-time check).

    tools/check_exec_spills.py FILE.s [FILE.s ...]      (assembly from `hipcc -save-temps`)

Reports every basic block in which a VGPR spill store / reload (v_accvgpr_write_b32 aN, vM / v_accvgpr_read_b32 / scratch_store / scratch_load)
stands between the block's label and the instruction of the same block that switches lanes back on (`s_or_b64 exec, exec, ...`,
`s_mov_b64 exec, ...`, `s_or_saveexec_b64`, ...), with nothing but scalar code around it (check), and every spill store that stands inside a
straight-line `s_and_saveexec_b64 ... s_or_b64 exec, exec` region without the region having written its source (check_regions: the second form the
fault took, two instructions in front of the restore and no label in between).  Exit status 1 if any kernel has one.  It knows these two forms, not
the fault in general; `--broad` lists every spill store at a deeper exec nesting than its source's last write instead — candidates to
read, false positives included, exit status 0."""
import re
import sys

SPILL = ("v_accvgpr_write_b32", "v_accvgpr_read_b32", "scratch_store", "scratch_load")


def kernels(path):
    lines = open(path).read().split("\n")
    name = None
    body = []
    for l in lines:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", l)
        if m:
            name, body = m.group(1), []
            continue
        if name and l.startswith(".Lfunc_end"):
            yield name, body
            name = None
            continue
        if name is not None:
            body.append(l)


def vregs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"(?<![\[:\w])v(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def widens_exec(op, args):
    """an instruction that may switch lanes back ON: the end of a branch over part of the wavefront (or the flip to its else side)"""
    a = args.replace(" ", "")
    return ((op in ("s_or_b64", "s_xor_b64") and a.startswith("exec,exec,")) or (op == "s_mov_b64" and a.startswith("exec,")) or
            (op in ("s_or_saveexec_b64", "s_xor_saveexec_b64", "s_andn2_saveexec_b64", "s_orn2_saveexec_b64") and not a.endswith(",-1")))
            # (`s_or_saveexec_b64 sN, -1` opens a whole-wave section around the spill VGPRs of the scalar registers: not a join)


def check(body):
    """A JOIN HEAD is the run of instructions from a label to the `s_or_b64 exec, exec, ...` of the same block when that run holds nothing but
    spill code, scalar instructions, lane moves and waits.  A spill STORE in it is a finding unless the stored VGPR was written in the branch body
    that falls through into the label (then it is the ordinary spill-at-definition of a value that only exists for the branch's lanes); a RELOAD
    in it is always one (its consumer lies behind the restore)."""
    instrs = []                                            # (label or None, op, args)
    for l in body:
        s = l.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            instrs.append((m.group(1), None, None))
            continue
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        instrs.append((None, parts[0], parts[1] if len(parts) > 1 else ""))
    found = []
    for i, (label, _, _) in enumerate(instrs):
        if label is None:
            continue
        head, j, pure = [], i + 1, True
        while j < len(instrs):
            lab, op, args = instrs[j]
            if lab is not None:
                pure = False
                break
            if widens_exec(op, args):
                break
            if op.startswith(SPILL) or op.startswith(("s_mov", "s_waitcnt", "s_nop", "v_readlane", "v_writelane", "s_add", "s_and_b", "s_or_b", "s_lshl", "s_cselect", "s_cmp")):
                if "exec" in args.split(",")[0]:
                    pure = False
                    break
                head.append((op, args))
                j += 1
                continue
            pure = False
            break
        if not pure or j >= len(instrs):
            continue
        if instrs[j][1] == "s_mov_b64" and any(op == "s_mov_b64" and a.replace(" ", "").endswith(",exec") for op, a in head):
            continue                                       # exec saved in this head, then narrowed: the entry of a branch, not its join
        stores = [(op, a) for op, a in head if op in ("v_accvgpr_write_b32",) or op.startswith("scratch_store")]
        loads = [(op, a) for op, a in head if op in ("v_accvgpr_read_b32",) or op.startswith("scratch_load")]
        if not stores and not loads:
            continue
        # registers written in the branch body in front of the label (back to the s_and_saveexec / branch that opens it)
        written = set()
        k = i - 1
        while k >= 0:
            lab, op, args = instrs[k]
            if lab is not None or op.startswith(("s_cbranch", "s_branch")) or op.endswith("saveexec_b64"):
                break
            if op.startswith(("v_", "ds_read", "global_load", "scratch_load", "buffer_load")) and not op.startswith(("v_cmp", "v_accvgpr_write", "v_writelane")):
                written |= vregs(args.split(",")[0])
            k -= 1
        bad = []
        for op, a in stores:
            src = a.split(",")[1] if op == "v_accvgpr_write_b32" else a.split(",")[1]
            if not (vregs(src) <= written):
                bad.append(op + " " + a)
        bad += [op + " " + a for op, a in loads]
        if bad:
            found.append((label, bad, instrs[j][1] + " " + instrs[j][2]))
    return found


def check_regions(body):
    """The same fault without a label in between: `s_and_saveexec_b64 sN, ...` ... straight-line code ... `s_or_b64 exec, exec, sN`, and inside it a
    spill store of a VGPR that the region itself did not write (dd_hmm_kernel<8, 6, GBT>: `v_accvgpr_write_b32 a3, v58` two instructions in front of
    the restore; v58 is reloaded from a3 at the item loop's back edge)."""
    found = []
    open_at, saved, written, stores = None, None, set(), []
    for l in body:
        s = l.split(";")[0].strip()
        if not s or (s.startswith(".") and not s.startswith(".LBB")):
            continue
        if s.startswith(".LBB"):
            open_at = None                                 # a label: not straight-line any more (the join-head rule covers that form)
            continue
        parts = s.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        a = args.replace(" ", "")
        if op == "s_and_saveexec_b64":
            open_at, saved, written, stores = s, a.split(",")[0], set(), []
            continue
        if open_at is None:
            continue
        if op == "s_or_b64" and a.startswith("exec,exec,"):
            if a.split(",")[2] == saved and stores:
                found.append((open_at, stores, s))
            open_at = None
            continue
        if op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm")) or "exec" in a.split(",")[0]:
            open_at = None
            continue
        if op == "v_accvgpr_write_b32" or op.startswith("scratch_store"):
            if not (vregs(args.split(",")[1]) <= written):
                stores.append(s)
            continue
        if op.startswith(("v_", "ds_read", "global_load", "scratch_load", "buffer_load")) and not op.startswith(("v_cmp", "v_writelane", "v_readlane", "v_readfirstlane")):
            written |= vregs(args.split(",")[0])
    return found


def broad_scan(body):
    """--broad: every spill store that runs at a deeper exec nesting than the last write of its source register (nesting counted along the straight
    listing: `s_and_saveexec` / saved-and-narrowed `s_mov_b64 exec` open a level, `s_or_b64 exec, exec, <that mask>` closes it).  A superset of the two
    checks with false positives (loops, values that only exist for the branch's lanes): a list of places to look at, not a verdict."""
    stack, last_save, defd, pending = [], None, {}, []
    for l in body:
        s = l.split(";")[0].strip()
        if not s or s.startswith("."):
            continue
        parts = s.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        a = args.replace(" ", "")
        ops = [x.strip() for x in args.split(",")]
        if op.endswith("saveexec_b64") and not a.endswith(",-1"):
            stack.append(ops[0]); continue
        if op == "s_mov_b64" and a.endswith(",exec"):
            last_save = ops[0]; continue
        if op == "s_mov_b64" and a.startswith("exec,"):
            if ops[1] in stack:
                while stack and stack[-1] != ops[1]: stack.pop()
                if stack: stack.pop()
            elif last_save is not None:
                stack.append(last_save)
            continue
        if op == "s_or_b64" and a.startswith("exec,exec,"):
            if ops[2] in stack:
                while stack and stack[-1] != ops[2]: stack.pop()
                stack.pop()
            continue
        d = len(stack)
        # a store like that is harmless when its reload comes back inside the same branch (a register borrowed for the branch's own lanes):
        # it stays a candidate only if the nesting drops below d before the slot is read again
        for c in pending:
            if c[2] is None and d < c[1]:
                c[2] = True                                # the branch closed first
        if op == "v_accvgpr_read_b32" or op.startswith("scratch_load"):
            slot = ops[1] if op == "v_accvgpr_read_b32" else a.split("offset:")[-1] if "offset:" in a else "0"
            for c in pending:
                if c[2] is None and c[3] == slot:
                    c[2] = False                           # read back inside the branch
        if op == "v_accvgpr_write_b32" or op.startswith("scratch_store"):
            if any(defd.get(v, 0) < d for v in vregs(ops[1])):
                slot = ops[0] if op == "v_accvgpr_write_b32" else a.split("offset:")[-1] if "offset:" in a else "0"
                pending.append([s, d, None, slot])
            continue
        if op.startswith(("v_", "ds_read", "global_load", "scratch_load", "buffer_load")) and not op.startswith(("v_cmp", "v_writelane", "v_readlane", "v_readfirstlane")):
            for v in vregs(ops[0]):
                defd[v] = d
    return [(c[0], c[1]) for c in pending if c[2]]


def main():
    if "--broad" in sys.argv:
        n = 0
        for path in [x for x in sys.argv[1:] if x != "--broad"]:
            for name, body in kernels(path):
                f = broad_scan(body)
                if f:
                    n += 1
                    print("%s: %s: %d spill store(s) deeper than their source's last write, e.g. %s (nesting %d)" % (path.split("/")[-2] if "/" in path else path, name, len(f), f[0][0], f[0][1]))
        print("kernels listed by the broad scan: %d" % n)
        return 0
    bad = 0
    for path in sys.argv[1:]:
        for name, body in kernels(path):
            f = check(body) + check_regions(body)
            if f:
                bad += 1
                print("%s: %s: %d block(s) with spill code in front of the exec restore" % (path.split("/")[-1], name, len(f)))
                for block, pend, restore in f[:6]:
                    print("    %s: %s  |  %s" % (block, "; ".join(pend[:4]) + (" ..." if len(pend) > 4 else ""), restore))
    print("kernels with the pattern: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
