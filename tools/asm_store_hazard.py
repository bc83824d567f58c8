#!/usr/bin/env python3
"""Scan hipcc assembly (-S) for the gfx950 store-data hazard hipcc's hazard recogniser does not cover: a buffer_store_dwordx3/x4 whose
offset comes from an SGPR, followed by a VALU (or VMEM-load) write to one of its data registers before two wait states have passed.
usage: python tools/asm_store_hazard.py file.s [...]   exit code 1 if any is found."""
import re, sys

def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()

def scan(path):
    found = 0
    kernel = "?"
    lines = [l.strip() for l in open(path)]
    code = []
    for l in lines:
        if l.endswith(":") and l.startswith("_Z"): kernel = l[:-1]
        if not l or l.startswith((";", ".", "//")) or l.endswith(":"): continue
        code.append((kernel, l))
    for i, (k, l) in enumerate(code):
        m = re.match(r"buffer_store_dwordx[34] (v\[\d+:\d+\]), \S+ s\[\d+:\d+\], (\S+)", l)
        if not m or not re.match(r"s\d+", m.group(2)): continue
        data = regs(m.group(1))
        waited = 0   # wait states between the store and the instruction looked at: one per instruction, N + 1 for `s_nop N`
        for k2, nxt in code[i + 1:i + 4]:
            if waited >= 2: break
            m2 = re.match(r"s_nop (\d+)", nxt)
            if m2:
                waited += int(m2.group(1)) + 1
                continue
            ops = nxt.split(None, 1)
            dst = ops[1].split(",")[0].strip() if len(ops) > 1 else ""
            if not nxt.startswith(("s_", "buffer_store", "global_store", "ds_write")) and regs(dst) & data:
                found += 1
                print(f"{path}: {k[:60]}: `{l}` then (after {waited} wait states) `{nxt}`")
                break
            waited += 1
    return found

if __name__ == "__main__":
    n = sum(scan(p) for p in sys.argv[1:])
    print("store-data hazards:", n)
    sys.exit(1 if n else 0)
