#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass) per kernel template and grid size."""
import csv, glob, sys, collections, os
root = sys.argv[1]
data = collections.defaultdict(lambda: collections.defaultdict(float))   # key -> counter -> sum
ndisp = collections.defaultdict(set)
dur = collections.defaultdict(float)
for f in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        tags = {"conv3t_kernel": "T:", "conv3r_kernel": "R:", "conv3s_kernel": "S16:", "conv3p_kernel": "P:", "conv3w_kernel": "W:", "mix16b_kernel": "MIX16B:", "mix16_kernel": "MIX16:", "conv_kernel": ""}
        tag = next((t for k, t in tags.items() if k in name), None)
        if tag is None: continue
        short = tag + name.split("_kernel<")[1].split(">")[0].replace("mz::", "")
        key = (short, int(r["Grid_Size"]) // int(r.get("Workgroup_Size", 256) or 256))
        data[key][r["Counter_Name"]] += float(r["Counter_Value"])
        did = (f, r["Dispatch_Id"])
        if did not in seen:
            seen.add(did)
            ndisp[key].add(did)
            if "sqA" in f:
                dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
keys = sorted(data, key=lambda k: -dur[k])
for k in keys[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    d = data[k]
    n = len([x for x in ndisp[k] if "sqA" in x[0]]) or 1
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k[0]:14s} wgs={k[1]:6d} n={n:3d} ms={dur[k]:8.2f} | MFMA_BUSY/BUSY_CYC={d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(1,d.get('SQ_BUSY_CYCLES',1)):.3f}"
          f" wait_any={d.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={d.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active={d.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f}"
          f" wait_lds={d.get('SQ_WAIT_INST_LDS',0)/wc:.3f} | lds_conf/idx={d.get('SQ_LDS_BANK_CONFLICT',0)/max(1,d.get('SQ_LDS_IDX_ACTIVE',1)):.3f}"
          f" | FETCH MB/disp={d.get('FETCH_SIZE',0)/1024/max(1,n):.1f} WRITE MB/disp={d.get('WRITE_SIZE',0)/1024/max(1,n):.1f}"
          f" | L2hit={d.get('TCC_HIT_sum',0)/max(1,d.get('TCC_HIT_sum',0)+d.get('TCC_MISS_sum',0)):.2f}"
          f" | GUI_ACTIVE/8/ms={d.get('GRBM_GUI_ACTIVE',0)/8/max(1e-9,dur[k])/1e3/ (len([x for x in ndisp[k] if 'grbm' in x[0]]) or 1) * n:.0f} MHz?")
    print("     raw:", {c: f"{v:.3g}" for c, v in sorted(d.items())})
