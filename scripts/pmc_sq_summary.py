"""Aggregate a rocprofv3 --pmc pass of SQ counters (+ GRBM_GUI_ACTIVE) per kernel family -> profiles/r02_pmc_sq_counters_summary.csv.
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs,
MI355X_MICROARCH.md 'DVFS give-back'); the other SQ counters are in quad-cycles summed over all waves.
usage: python scripts/pmc_sq_summary.py <counter_collection.csv> <out.csv>"""
import csv, re, sys
from collections import defaultdict

def family(name):
    m = re.search(r"magpo::(k_\w+)(<[^>]*>)?", name)
    return None if not m else m.group(1) + (m.group(2) or "").replace(" ", "")

acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    f = family(r["Kernel_Name"])
    if f:
        acc[f][r["Counter_Name"]] += float(r["Counter_Value"])
        n[f].add(r["Dispatch_Id"])
cols = sorted({c for v in acc.values() for c in v})
with open(sys.argv[2], "w") as out:
    out.write("kernel,launches," + ",".join(cols) + ",mfma_util\n")
    rows = sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))
    for f, v in rows:
        cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
        util = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * 256 * cyc) if cyc else float("nan")
        out.write(f'"{f}",{len(n[f])},' + ",".join(f"{v.get(c, 0):.6g}" for c in cols) + f",{util:.3f}\n")
        print(f"{f:28s} launches {len(n[f]):4d}  mfma_util {util:5.3f}  wait_any/wave {v.get('SQ_WAIT_ANY', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1):5.2f}")
