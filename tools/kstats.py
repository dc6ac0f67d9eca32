#!/usr/bin/env python3
"""Print a compact table from a rocprofv3 *_kernel_stats.csv (name, calls, total ms, avg us, %)."""
import csv, sys, re
for f in sys.argv[1:]:
    print("==", f)
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:16]:
        n = r["Name"]
        m = re.search(r"idb_gemm_kernel(_rs|_pl)?I(DF16b|DF16_)Li(\d)ELi(\d)E(?:Li(\d)E)?(?:Li(\d)E)?", n)
        if m:
            wm = int(m.group(6) or 2)
            n = (f"idb_gemm_kernel{m.group(1) or ''}<{'bf16' if m.group(2) == 'DF16b' else 'f16'},{16 * wm * int(m.group(3))}x{32 * int(m.group(4))}"
                 f",{2 * wm}w" + (f",ring{m.group(5)}>" if m.group(5) else ">"))
        else:
            m = re.search(r"\d+([a-z_0-9]+_kernel)", n)
            n = m.group(1) if m else n[:48]
        print(f'{n:44s} calls {int(r["Calls"]):6d} tot_ms {float(r["TotalDurationNs"])/1e6:9.2f} avg_us {float(r["AverageNs"])/1e3:8.1f} {float(r["Percentage"]):5.1f}%')
    print(f"total kernel ms {tot/1e6:.1f}")
