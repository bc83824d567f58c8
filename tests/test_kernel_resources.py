"""Build-time guard on the properties the hot kernels depend on: no scratch memory and no spilled vector registers.

Several 2x regressions of earlier rounds came from hipcc spilling a handful of VGPRs of a kernel that lives at its register
cap; nothing failed when that happened.  This test compiles the kernel translation units for gfx950
(device code only, no GPU needed) with -Rpass-analysis=kernel-resource-usage and fails if a persistent 3x3 kernel, the mix
kernel or the image head reports ScratchSize != 0 or VGPR spills."""

import os
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

import pytest

CSRC = Path(__file__).resolve().parent.parent / "ultrazoom_amd" / "csrc"
TOOLS = Path(__file__).resolve().parent.parent / "tools"
LISTINGS = {}
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# kernels the execution plans of the 16-bit modes use (name prefix of the demangled-ish symbol)
HOT = ("conv3r_kernel", "conv3t_kernel", "conv3s_kernel", "mix16_kernel", "mix16b_kernel", "conv3w_kernel")
# Instantiations that are NOT on the default plans of the 16-bit modes and are known to spill (fallbacks / A-B knobs):
#   conv3s_kernel<T, NT = 3, *, FUSE>: the C = 65..96 fused conv2 + mix; conv3r_kernel's fused variant takes it wherever the tile has six
#                                      or more chunks (hidden_ratio >= 2), so this one only runs for hidden_ratio 1 or MZ_NO_R=1
#   conv3w_kernel<T16, NT = 3, ..>:    per-tile wide kernels of the 16-bit types, reachable with MZ_NO_PERSIST=1 / MZ_NO_FUSE16=1 only
EXEMPT = (re.compile(r"conv3s_kernelINS_\w+ELi3ELi\dELb1E"), re.compile(r"conv3w_kernelINS_\w+ELi3ELi\dELb[01]E"))


def resource_usage(src: str):
    """One device-only compile per translation unit: the resource remarks on stderr, the assembly listing kept for the hazard scan."""
    asm = Path(tempfile.gettempdir()) / f"mz_guard_{os.getpid()}_{src}.s"
    p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-Rpass-analysis=kernel-resource-usage",
                        str(CSRC / src), "-o", str(asm)], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    LISTINGS[src] = asm
    out = {}
    name = None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\])?: (\d+)", line)
        if m and name:
            out[name][m.group(1).strip()] = int(m.group(2))
    return out


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
def test_hot_kernels_use_no_scratch():
    from concurrent.futures import ThreadPoolExecutor

    sources = ["mz_kernels.hip", "mz_conv3r.hip", "mz_conv3t.hip"]
    with ThreadPoolExecutor(max_workers=4) as ex:  # four hipcc processes side by side: ~2 minutes in total
        usages = dict(zip(sources, ex.map(resource_usage, sources)))
    for src, usage in usages.items():
        hot = {k: v for k, v in usage.items() if any(h in k for h in HOT) and ("TBF16" in k or "TF16" in k) and not any(e.search(k) for e in EXEMPT)}
        assert hot, f"no hot kernel found in {src}: {list(usage)[:5]}"
        bad = {k: v for k, v in hot.items() if v.get("ScratchSize", 0) != 0 or v.get("VGPRs Spill", 0) != 0}
        assert not bad, "kernels with scratch memory / spilled VGPRs: " + ", ".join(f"{k}: {v}" for k, v in bad.items())
        for k, v in hot.items():
            if "conv3r_kernel" in k or "conv3t_kernel" in k:
                assert v.get("Occupancy", v.get("Occupancy [waves/SIMD]", 2)) >= 2, f"{k}: two waves per SIMD are the design"
    # The same listings, scanned for the store-data hazard of gfx950: a 12- / 16-byte store (buffer or global, any offset form) whose data
    # registers are overwritten within two wait states -- hipcc inserts none behind a buffer store with an SGPR offset and one behind the
    # other forms, and tests/test_store_hazard_gpu.py measured corruption at that distance (tools/asm_store_hazard.py follows branches).
    sys.path.insert(0, str(TOOLS))
    import asm_store_hazard

    try:
        hazards = {src: asm_store_hazard.scan(str(path)) for src, path in LISTINGS.items()}
    finally:
        for path in LISTINGS.values():
            path.unlink(missing_ok=True)
    assert not any(hazards.values()), f"store-data hazards in the listings: {hazards}"


def test_store_hazard_scanner_on_synthetic_listings(tmp_path):
    """The scanner itself.  The pair hipcc produced in mix16b_kernel is flagged and the pinned `s_nop 1` clears it; every 16-byte store
    form is held to two wait states (profiles/r04_store_hazard_probe.json: the buffer store with soffset 0 still corrupts with the ONE
    wait state hipcc inserts); writers include loads, packed and matrix instructions and both operands of a lane swap; a store at the end
    of a loop body is checked against the loop head; writes to other registers, stores and scalar code are not flagged."""
    sys.path.insert(0, str(TOOLS))
    import asm_store_hazard

    def count(body: str) -> int:
        f = tmp_path / "k.s"
        f.write_text("_Z6kernelv:                             ; @_Z6kernelv\n" + body)
        return asm_store_hazard.scan(str(f))

    store = "\tbuffer_store_dwordx4 v[48:51], v183, s[16:19], s2 offen\n"
    mul50 = "\tv_mul_f32_e32 v50, 0xbfb8aa3b, v166\n"
    assert count(store + mul50) == 1
    assert count(store + "\ts_mov_b32 s4, 0\n" + mul50) == 1       # one wait state is not enough
    assert count(store + "\ts_nop 0\n" + mul50) == 1
    assert count(store + "\ts_nop 1\n" + mul50) == 0
    assert count(store + "\ts_mov_b32 s4, 0\n\ts_mov_b32 s5, 0\n" + mul50) == 0
    assert count(store + "\tv_mul_f32_e32 v52, 0xbfb8aa3b, v166\n\tv_exp_f32_e32 v53, v52\n") == 0
    assert count(store + "\tbuffer_load_dwordx4 v[48:51], v185, s[8:11], s2 offen\n") == 1
    assert count(store + "\tbuffer_store_dwordx4 v[48:51], v184, s[16:19], s2 offen\n" + "\ts_nop 1\n" + mul50) == 0   # stores read, they do not write
    # the other store forms: soffset 0 / an immediate, global stores with and without saddr, 12-byte stores
    for st in ("\tbuffer_store_dwordx4 v[48:51], v183, s[16:19], 0 offen\n", "\tbuffer_store_dwordx4 v[48:51], v183, s[16:19], 0 offen offset:64\n",
               "\tglobal_store_dwordx4 v[190:191], v[48:51], off\n", "\tglobal_store_dwordx4 v7, v[48:51], s[4:5]\n",
               "\tglobal_store_dwordx3 v[190:191], v[48:50], off offset:16\n"):
        assert count(st + mul50) == 1, st
        assert count(st + "\ts_nop 0\n" + mul50) == 1, st      # what hipcc inserts for these forms: not enough on gfx950
        assert count(st + "\ts_nop 1\n" + mul50) == 0, st
    # 8-byte stores read their data at issue: not this hazard
    assert count("\tglobal_store_dwordx2 v[190:191], v[50:51], off\n" + mul50) == 0
    # writers: packed f32 (a register pair), MFMA (the accumulator tuple), lane swaps (both operands)
    assert count(store + "\tv_pk_mul_f32 v[50:51], v[4:5], v[6:7]\n") == 1
    assert count(store + "\tv_pk_mul_f32 v[52:53], v[48:49], v[50:51]\n") == 0          # reads them only
    assert count(store + "\tv_mfma_f32_16x16x32_bf16 v[48:51], v[4:7], v[8:11], v[48:51]\n") == 1
    assert count(store + "\tv_mfma_f32_16x16x32_bf16 a[48:51], v[4:7], v[8:11], a[48:51]\n") == 0   # AGPRs are another file
    assert count(store + "\tv_permlane16_swap_b32_e32 v7, v49\n") == 1
    # a store at the end of a loop body against the head of the loop (and the fall-through path)
    loop = ".LBB0_1:\n" + mul50 + "\tv_add_f32_e32 v1, v2, v3\n\tv_add_f32_e32 v1, v2, v3\n" + store
    assert count(loop + "\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n") == 1
    assert count(loop + "\ts_nop 0\n\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n") == 0   # nop + branch: two wait states
    assert count(loop + "\ts_cbranch_scc1 .LBB0_1\n" + mul50) == 1                      # ... the fall-through path
    assert count(loop + "\ts_branch .LBB0_2\n" + mul50 + ".LBB0_2:\n\ts_endpgm\n") == 0  # an unconditional branch does not fall through
