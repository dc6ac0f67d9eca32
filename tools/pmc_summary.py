#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per launch.

Usage: python tools/pmc_summary.py <counter_collection.csv> [--json out.json] [--match substring]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide
coalesced reads (16 B/lane, global_load and buffer_load...lds alike) at 64 B, so the HBM-read estimate is
2 x FETCH_SIZE (MI355X_MICROARCH.md, section HBM).  `hbm_bytes_per_launch` = 2 x FETCH_SIZE + WRITE_SIZE, in bytes."""
import csv
import json
import re
import sys


def short(n):
    m = re.search(r"idb_gemm_kernel_lwI(DF16b|DF16_)Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)E", n)
    if m:                                       # loader-wave variants: the display names of bench.TILE_NAMES
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        mf, nf, ns, wm, lw = (int(m.group(i)) for i in range(2, 7))
        for big in (False, True):
            for shape, cfg in bench._KTILES.items():
                if cfg == (mf // 2 if big else mf, nf, wm) and (not big or mf % 2 == 0) and shape in ((8, 9) if big else (4, 6, 7, 8, 9)):
                    v = 8 if big else {(3, 4): 5, (3, 8): 6, (4, 4): 7}[(ns, lw)]
                    return bench.TILE_NAMES[10 * v + shape]
    m = re.search(r"idb_conv_patch_kernelI(DF16b|DF16_)Li(\d)ELi(\d)ELi(\d)ELb(\d)E", n)
    if m:                                       # patch-resident conv: the display names of bench.TILE_NAMES (tile ids 9x / 10x, + 1000 with a fused GroupNorm)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        mf, nf, gn = int(m.group(2)), int(m.group(3)), m.group(5) == "1"
        shape = {(1, 2): 4, (1, 5): 6, (1, 4): 7, (2, 5): 8, (2, 4): 9, (4, 5): 8, (4, 4): 9}[(mf, nf)]
        return bench.TILE_NAMES[(90 if mf == 4 else 100) + shape + (1000 if gn else 0)]
    m = re.search(r"idb_gemm_kernel(_rs|_pl)?I(DF16b|DF16_)Li(\d)ELi(\d)E(?:Li(\d)E)?(?:Li(\d)E)?", n)
    if m:
        wm = int(m.group(6) or 2)
        return (f"idb_gemm_kernel{m.group(1) or ''}<{16 * wm * int(m.group(3))}x{32 * int(m.group(4))}"
                + (",8w" if wm == 4 else "") + (f",ring{m.group(5)}>" if m.group(5) else ">"))
    m = re.search(r"\d+([a-z_0-9]+_kernel)", n)
    return m.group(1) if m else n[:60]


def main():
    path = sys.argv[1]
    out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else None
    agg = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if match and match not in k:
                continue
            a = agg.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    res = {}
    for k, cs in agg.items():
        e = {c: v[0] / v[1] for c, v in cs.items()}
        e["launches"] = max(v[1] for v in cs.values())
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
        res[k] = e
    for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0) * kv[1]["launches"]):
        print(f"{k:40s} launches {e['launches']:6d}  " + "  ".join(f"{c}={v:.4g}" for c, v in e.items() if c != "launches"))
    if out:
        json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
