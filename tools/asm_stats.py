#!/usr/bin/env python3
"""Per-kernel instruction statistics of a hipcc -S listing: MFMAs, scratch traffic, SGPR spill lanes, barriers, and where
(between which barriers) the scratch / lane instructions sit.  usage: asm_stats.py file.s [name-filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r'\n(?=_Z[^\n:]*:[^\n]*\n)', s)
for fn in parts[1:]:
    name = fn.split(':')[0]
    if flt not in name:
        continue
    body = fn.split('.Lfunc_end')[0].split('\n')
    cnt = lambda k: sum(k in l for l in body)
    print(f"{name}: {len(body)} lines, mfma {cnt('v_mfma')}, scratch st/ld {cnt('scratch_store')}/{cnt('scratch_load')}, "
          f"writelane/readlane {cnt('v_writelane')}/{cnt('v_readlane')}, barriers {cnt('s_barrier')}, ds_read {cnt('ds_read')}, "
          f"buffer_store {cnt('buffer_store')}, lds-dma {sum(('lds' in l and ('buffer_load' in l or 'global_load' in l)) for l in body)}")
    # segments between barriers
    seg = []; cur = dict(mfma=0, sc=0, lane=0, lines=0, valu=0)
    for l in body:
        t = l.strip()
        if 's_barrier' in t:
            seg.append(cur); cur = dict(mfma=0, sc=0, lane=0, lines=0, valu=0)
        cur['lines'] += 1
        if 'v_mfma' in t: cur['mfma'] += 1
        elif 'scratch_' in t: cur['sc'] += 1
        elif 'v_writelane' in t or 'v_readlane' in t: cur['lane'] += 1
        elif t.startswith('v_'): cur['valu'] += 1
    seg.append(cur)
    print("   segments (mfma/valu/scratch/lane): " + " | ".join(f"{c['mfma']}/{c['valu']}/{c['sc']}/{c['lane']}" for c in seg))
