"""Aggregate the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_traffic.sh into HBM bytes per launch per kernel.

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests of wide coalesced reads at
64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Both counters are in KiB.
"""
import collections
import csv
import json
import re
import sys

tag = sys.argv[1]
KERNELS = ["conv_fwd2_kernel_256", "conv_fwd2_kernel_192", "conv_fwd2_kernel_128", "conv_fwd2_kernel_64", "conv_stream_kernel", "conv_wgrad2s_kernel",
           "conv_wgrad2_kernel", "conv_wgrad3_kernel", "conv_fwd_kernel",
           "conv_wgrad_kernel", "slab_reduce_kernel", "rowstream_kernel", "film_softmax", "merge_taps", "conv_pack_kernel"]


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    m = re.search(r"([A-Za-z_0-9]+)(<|\()", name)
    return "other:" + (m.group(1) if m else name[:40])


def load(which):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"gpurun_out/{tag}_{which}/r_counter_collection.csv")):
        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


fetch, write = load("fetch"), load("write")
out = {"note": "KiB counters; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 averaged over the launches of one "
               "eager bench step + one warm-up step (config B, 64 windows, bf16); FETCH_SIZE doubled per the gfx950 correction",
       "kernels": {}}
for k in sorted(set(fetch) | set(write), key=lambda k: -sum(fetch.get(k, [0])) - sum(write.get(k, [0]))):
    f, w = fetch.get(k, []), write.get(k, [])
    n = max(len(f), len(w))
    fa = sum(f) / len(f) if f else 0.0
    wa = sum(w) / len(w) if w else 0.0
    out["kernels"][k] = {"launches": n, "fetch_size_kib_avg": fa, "write_size_kib_avg": wa,
                         "hbm_bytes_per_launch": (2 * fa + wa) * 1024.0}
json.dump(out, open(f"gpurun_out/{tag}_traffic.json", "w"), indent=1)
for k, v in list(out["kernels"].items())[:12]:
    print(f"{k:32s} n={v['launches']:4d} fetch {v['fetch_size_kib_avg']/1024:9.1f} MiB (x2) write {v['write_size_kib_avg']/1024:9.1f} MiB  -> {v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch")
