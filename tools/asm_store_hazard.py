#!/usr/bin/env python3
"""Scan hipcc assembly (-S) for the gfx950 store-data hazard: a 12- or 16-byte store followed, before TWO wait states have passed, by an
instruction that writes one of its data registers.

Measured on MI355X (ultrazoom_amd/csrc/mz_probe.hip, tests/test_store_hazard_gpu.py, profiles/r04_store_hazard_probe.json): the store
reads its data registers late --
  * buffer_store_dwordx4 with soffset in an SGPR: corrupted with 0 wait states, clean with 1.  hipcc inserts NONE for this form
    (its hazard recogniser exempts stores whose soffset is a register);
  * buffer_store_dwordx4 with soffset 0 and global_store_dwordx4 (either address form): see the profile for what was measured;
    hipcc inserts ONE wait state for them.
This scanner demands two wait states for every form (one of margin over the worst measured case) unless --measured is given, and
follows branches: a store at the end of a loop body is checked against the writes at the loop head.

Wait states counted per instruction: 1, `s_nop N`: N + 1.  Writers: anything that is not a store / LDS write / scalar instruction and
whose first operand (the destination; loads, VALU, v_pk_*, v_mfma, v_permlane*_swap alike) overlaps the data registers.
v_permlane*_swap writes BOTH its operands.

usage: python tools/asm_store_hazard.py file.s [...]   exit code 1 if any is found."""
import re
import sys

NEED = 2
STORE_RE = re.compile(
    r"(?:buffer_store_dwordx[34]\s+(?P<bdata>v\[\d+:\d+\]),\s*\S+\s+s\[\d+:\d+\],\s*(?P<soff>\S+)"
    r"|(?:global|flat|scratch)_store_dwordx[34]\s+(?P<gaddr>\S+),\s*(?P<gdata>v\[\d+:\d+\]))")


def regs(tok):
    tok = tok.strip()
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        base = 0 if tok[0] == "v" else 1000  # AGPRs: a separate file
        return set(range(base + int(m.group(1)), base + int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    m = re.match(r"a(\d+)$", tok)
    return {1000 + int(m.group(1))} if m else set()


def written(instr):
    """VGPRs an instruction writes (empty for stores, LDS writes and scalar instructions)."""
    ops = instr.split(None, 1)
    if len(ops) < 2 or instr.startswith(("s_", "buffer_store", "global_store", "flat_store", "scratch_store", "ds_write", "ds_store",
                                         "buffer_atomic", "global_atomic", ";")):
        return set()
    fields = [f.strip() for f in ops[1].split(",")]
    out = regs(fields[0])
    if ops[0].startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap")) and len(fields) > 1:
        out |= regs(fields[1])
    return out


def parse(path):
    """-> (code: list of (kernel, instruction), labels: name -> index of the first instruction behind it)"""
    code, labels = [], {}
    kernel = "?"
    for raw in open(path):
        l = raw.split(";")[0].split("//")[0].strip()  # (comments off first: hipcc writes `label:   ; @label`)
        if not l:
            continue
        if l.endswith(":") and not l.startswith("."):
            kernel = l[:-1]
            labels[l[:-1]] = len(code)
            continue
        if l.endswith(":"):
            labels[l[:-1]] = len(code)
            continue
        if l.startswith("."):
            continue
        code.append((kernel, l))
    return code, labels


def scan(path, need=NEED):
    code, labels = parse(path)
    found = 0
    for i, (k, l) in enumerate(code):
        m = STORE_RE.match(l)
        if not m:
            continue
        data = regs(m.group("bdata") or m.group("gdata"))
        # walk every path of at most `need` wait states behind the store: (index, wait states so far)
        todo, seen = [(i + 1, 0)], set()
        while todo:
            j, waited = todo.pop()
            if waited >= need or j >= len(code) or (j, waited) in seen:
                continue
            seen.add((j, waited))
            k2, nxt = code[j]
            if k2 != k and not nxt:  # (left the function)
                continue
            mn = re.match(r"s_nop\s+(\d+)", nxt)
            if mn:
                todo.append((j + 1, waited + int(mn.group(1)) + 1))
                continue
            if nxt.startswith("s_endpgm"):
                continue
            mb = re.match(r"s_(cbranch_\w+|branch)\s+(\S+)", nxt)
            if mb:
                tgt = labels.get(mb.group(2))
                if tgt is not None:
                    todo.append((tgt, waited + 1))
                if mb.group(1) != "branch":
                    todo.append((j + 1, waited + 1))
                continue
            if written(nxt) & data:
                found += 1
                print(f"{path}: {k[:70]}: `{l}` then (after {waited} wait states) `{nxt}`")
                break
            todo.append((j + 1, waited + 1))
    return found


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = sum(scan(p) for p in args)
    print("store-data hazards:", n)
    sys.exit(1 if n else 0)
