"""The store-data hazard of round 3, pinned in isolation.

`mix16b_kernel` once wrote garbage into parts of some output entries; the diagnosis was that on gfx950 a
`buffer_store_dwordx4 ..., s<N> offen` (soffset in an SGPR) still reads its data registers a cycle or two after issue, so that a
vector instruction writing one of them right behind it corrupts the store -- a pair LLVM's hazard recogniser does not separate (it
inserts the VMEM-store-data wait state only when soffset is NOT a register).  `mz_debug_store_hazard` (csrc/mz_probe.hip) issues
exactly that pair with hard registers and nothing else around it, for every follower kind / store form (buffer_store with soffset 0 or
in an SGPR, global_store with a 64-bit vaddr or with saddr) / wait-state count / data register.  This test runs the whole matrix ONCE, writes the table to gpurun_out/store_hazard_probe.json (committed copy:
profiles/r04_store_hazard_probe.json) and asserts what the kernels rely on:

  * two wait states (`s_nop 1`, what `store16_soff()` in mz_device.h pins behind every such store) are always enough;
  * the table is deterministic enough to be a record: the run is repeated and must give the same zero / non-zero pattern.

What the table says about ZERO wait states is recorded, not asserted: it is the finding (store-data hazard)."""

import ctypes
import json
from pathlib import Path

import pytest

from ultrazoom_amd import _ffi

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parent.parent
FOLLOWERS = ["v_mov_b32", "v_mul_f32", "v_cvt_pk_bf16_f32", "v_exp_f32", "v_pk_mul_f32", "v_mfma_f32_16x16x32_bf16"]


FORMS = ["buffer_store soffset=0", "buffer_store soffset=sgpr", "global_store vaddr64 off", "global_store saddr"]


def probe(follower, form, waits, dword, iters=64, blocks=1024):
    counts = (ctypes.c_uint * 5)()
    lib = _ffi.lib()
    lib.mz_debug_store_hazard.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_uint)]
    lib.mz_debug_store_hazard.restype = ctypes.c_int
    rc = lib.mz_debug_store_hazard(follower, form, waits, dword, iters, blocks, counts)
    assert rc == 0, f"mz_debug_store_hazard returned {rc}"
    return list(counts)


def run_matrix():
    rows = []
    for f, fname in enumerate(FOLLOWERS):
        for form, form_name in enumerate(FORMS):
            for waits in (0, 1, 2):
                for dword in range(4):
                    c = probe(f, form, waits, dword)
                    rows.append({"follower": fname, "store": form_name, "wait_states": waits, "dword_written": dword,
                                 "entries": 64 * 1024 * 4 * 64, "bad_entries": c[0], "bad_per_dword": c[1:]})
    return rows


def test_store_data_hazard_matrix():
    first = run_matrix()
    second = run_matrix()
    out = REPO / "gpurun_out"
    out.mkdir(exist_ok=True)
    summary = {}
    for r in first:
        key = f"{r['store']} / {r['wait_states']} wait states"
        summary.setdefault(key, {"combinations": 0, "corrupting": 0, "bad_entries": 0})
        summary[key]["combinations"] += 1
        summary[key]["corrupting"] += 1 if r["bad_entries"] else 0
        summary[key]["bad_entries"] += r["bad_entries"]
    (out / "store_hazard_probe.json").write_text(json.dumps({"summary": summary, "rows": first, "repeat_rows": second}, indent=1))
    print("\nstore-data hazard probe (bad 16-byte entries of 16.8 M per combination):")
    for k, v in summary.items():
        print(f"  {k}: {v['corrupting']} of {v['combinations']} follower/dword combinations corrupt, {v['bad_entries']} entries in total")
    # what the kernels rely on: s_nop 1 behind the store is enough, whatever follows and whichever register it writes
    for rows in (first, second):
        for r in rows:
            if r["wait_states"] == 2:
                assert r["bad_entries"] == 0, f"two wait states are NOT enough: {r}"
    # a record, not noise: the zero / non-zero pattern repeats
    assert [bool(r["bad_entries"]) for r in first] == [bool(r["bad_entries"]) for r in second]
