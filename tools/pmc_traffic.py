#!/usr/bin/env python3
"""Per-launch HBM-side traffic of the 3x3 conv kernels from rocprofv3 FETCH_SIZE / WRITE_SIZE passes.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced
streams -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Both counters are in KiB."""
import csv, glob, hashlib, json, sys, os, collections
root = sys.argv[1]
repo = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, repo)
from bench import kernel_source_sha256  # identity of the kernels the profile was taken with (sources, not the binary)
lib_sha = kernel_source_sha256()
tot = collections.defaultdict(float); n = collections.defaultdict(set)
for cname, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    for f in glob.glob(os.path.join(root, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            # (the 3x3 family: the persistent kernels, and conv_kernel in its CONV3 mode -- template argument MODE = 0: the image head)
            k3 = any(k in r["Kernel_Name"] for k in ("conv3s_kernel", "conv3p_kernel", "conv3w_kernel", "conv3r_kernel", "conv3t_kernel")) or \
                 ("conv_kernel<" in r["Kernel_Name"] and r["Kernel_Name"].split(">(")[0].endswith(", 0"))
            if not k3 or r["Counter_Name"] != cname: continue
            tot[cname] += float(r["Counter_Value"]); n[cname].add(r["Dispatch_Id"])
launches = len(n["FETCH_SIZE"]) or 1
fetch = 2.0 * tot["FETCH_SIZE"] * 1024 / launches
write = tot["WRITE_SIZE"] * 1024 / max(1, len(n["WRITE_SIZE"]))
out = {"kernel": "3x3 convolution kernels (every 3x3 launch of one 3-image micro-batch forward, cfg3 1080p, bf16)",
       "kernel_source_sha256": lib_sha, "images": 3,
       "launches": launches, "fetch_bytes_per_launch_corrected_x2": fetch, "write_bytes_per_launch": write,
       "traffic_bytes_per_launch": fetch + write,
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); Infinity-Cache hits are included in FETCH_SIZE"}
print(json.dumps(out, indent=1))
