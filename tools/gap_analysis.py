"""Idle-gap analysis of a rocprofv3 kernel trace: python tools/gap_analysis.py <kernel_trace.csv> [skip_first_n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
small = [g for g in gaps if 0 <= g < 50000]
print(f"kernels {len(rows)} span {span/1e6:.2f} ms busy {busy/1e6:.2f} ms ({100*busy/span:.1f}%)")
print(f"gaps <50us: n={len(small)} total {sum(small)/1e6:.2f} ms mean {sum(small)/max(len(small),1)/1e3:.2f} us; large gaps total {sum(g for g in gaps if g>=50000)/1e6:.2f} ms; overlaps {sum(1 for g in gaps if g<0)}")
