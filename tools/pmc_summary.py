#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as the MI355X guide
prescribes) into per-kernel HBM-side traffic per launch.

Units and correction (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): both counters are in KiB;
on gfx950 FETCH_SIZE tallies 128-byte requests of wide coalesced reads at 64 bytes, i.e. reports half
of the bytes -> doubled here.  Counted at the L2's memory side, so Infinity-Cache hits are included
(weights re-fetched by each of the 8 XCD L2s show up 8x).

Usage: python tools/pmc_summary.py <dir with pmc_r1_FETCH_SIZE/ and pmc_r1_WRITE_SIZE/> out.json
"""
import collections
import csv
import json
import os
import sys


def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def per_launch(values, kernel):
    """Mean over the launches; for the tick pipeline's launch the mean over FULL ticks only (the upper half of the
    sorted values: a run of K steps has n_stages - 1 partly filled ticks at either end)."""
    if "table_kernel" in kernel:
        v = sorted(values)
        v = v[len(v) // 2:]
        return sum(v) / len(v)
    return sum(values) / len(values)


def main(root, out):
    f = load(os.path.join(root, "pmc_r1_FETCH_SIZE", "pmc_counter_collection.csv"))
    w = load(os.path.join(root, "pmc_r1_WRITE_SIZE", "pmc_counter_collection.csv"))
    mfma_path = os.path.join(root, "pmc_r1_SQ_VALU_MFMA_BUSY_CYCLES", "pmc_counter_collection.csv")
    mf = load(mfma_path) if os.path.exists(mfma_path) else {}
    res = {}
    for k, v in f.items():
        if k.startswith("__amd_rocclr"):
            continue
        fetch_kb = per_launch(v, k)
        wv = w.get(k, [0.0])
        write_kb = per_launch(wv, k)
        res[k] = {"launches": len(v), "fetch_size_kib_raw": round(fetch_kb, 1), "write_size_kib": round(write_kb, 1),
                  "hbm_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024))}
        if k in mf:  # summed over the SIMDs that ran the kernel; one v_mfma_f32_16x16x4_f32 keeps a SIMD's pipe busy 32 cycles
            res[k]["mfma_busy_cycles_per_launch"] = round(per_launch(mf[k], k), 1)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench   # csrc_sha1: the sources these passes ran on; bench.py quotes the summary only while they are unchanged
    commit = os.popen("git -C %s rev-parse --short HEAD 2>/dev/null" % os.path.dirname(os.path.abspath(bench.__file__))).read().strip()
    json.dump({"csrc_sha1": bench.csrc_sha1(), "commit": commit or "(no .git on the GPU box: see csrc_sha1)",
               "note": "bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; bench.py --no-extras, B=256; tick launch: mean over full ticks",
               "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    for k, r in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print("%8.2f MB  %s" % (r["hbm_bytes_per_launch"] / 1e6, k[:120]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
