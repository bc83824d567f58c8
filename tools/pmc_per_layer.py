#!/usr/bin/env python3
"""Per-layer HBM-side traffic of one 3-image forward: joins the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_collect.sh (one counter file
each: pass the two *_counter_collection.csv) with the launch order of `bench.py --dump-launches` (same command, same launch sequence).
usage: tools/pmc_per_layer.py fetch_counter_collection.csv write_counter_collection.csv per_launch.csv
FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 bytes, MI355X_MICROARCH.md); Infinity-Cache hits are included in it."""
import csv, sys, collections
KERNELS = ("conv3s_kernel", "conv3p_kernel", "conv3w_kernel", "conv3r_kernel", "conv3t_kernel")
def counters(path, name):
    def is3(n):  # the persistent 3x3 kernels, and conv_kernel in its CONV3 mode (template argument MODE = 0: the image head)
        return any(k in n for k in KERNELS) or ("conv_kernel<" in n and n.split(">(")[0].endswith(", 0"))
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == name and is3(r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) * 1024 for r in rows]
fetch, write = counters(sys.argv[1], "FETCH_SIZE"), counters(sys.argv[2], "WRITE_SIZE")
launches = [r for r in csv.DictReader(open(sys.argv[3])) if r["kind"] == "conv3" and r["B"] == "3"]
n = len(fetch)
launches = launches[:n]  # the first 3-image micro-batch of the dump
assert len(write) == n and len(launches) == n, (len(fetch), len(write), len(launches))
g = collections.OrderedDict()
for f, w, r in zip(fetch, write, launches):
    k = (r["H"], r["W"], r["cin"], r["cout"])
    sz = 2.0
    px = 3.0 * int(r["H"]) * int(r["W"])
    cin, cout = int(r["cin"]), int(r["cout"])
    alg_r = px * cin * sz + 9.0 * cin * cout * sz
    alg_w = px * cout * sz
    e = g.setdefault(k, [0, 0.0, 0.0, 0.0, 0.0])
    e[0] += 1; e[1] += 2 * f; e[2] += w; e[3] += alg_r; e[4] += alg_w
print(f"{'layer (H W cin cout)':28s} {'n':>3s} {'fetch x2 MB':>12s} {'alg read MB':>12s} {'ratio':>6s} {'write MB':>9s} {'alg write':>9s}")
T = [0.0] * 4
for k, e in g.items():
    print(f"{' '.join(k):28s} {e[0]:3d} {e[1] / e[0] / 1e6:12.1f} {e[3] / e[0] / 1e6:12.1f} {e[1] / e[3]:6.2f} {e[2] / e[0] / 1e6:9.1f} {e[4] / e[0] / 1e6:9.1f}")
    T[0] += e[1]; T[1] += e[3]; T[2] += e[2]; T[3] += e[4]
print(f"{'all 3x3 launches':28s} {n:3d} {T[0] / n / 1e6:12.1f} {T[1] / n / 1e6:12.1f} {T[0] / T[1]:6.2f} {T[2] / n / 1e6:9.1f} {T[3] / n / 1e6:9.1f}")
print("(fused launches also read the block input x -- C more channels per pixel -- which the 'alg read' column of this table does not count)")
