"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel: pmc_summarize.py <dir>"""
import collections, csv, glob, json, re, sys
out = sys.argv[1]
res = collections.defaultdict(lambda: {"calls": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"[A-Za-z_0-9:]+(<[0-9a-z, ]+>)?", name)
    return m.group(0) if m else name[:40]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c: continue
        k = short(r["Kernel_Name"]); res[k][c] += float(r["Counter_Value"]); n[k] += 1
    for k, v in n.items(): res[k]["calls"] = v
rows = [{"kernel": k, "calls": v["calls"], "fetch_KiB_per_call": v["FETCH_SIZE"] / max(v["calls"], 1), "write_KiB_per_call": v["WRITE_SIZE"] / max(v["calls"], 1),
         "fetch_KiB_total": v["FETCH_SIZE"], "write_KiB_total": v["WRITE_SIZE"]} for k, v in sorted(res.items(), key=lambda kv: -kv[1]["FETCH_SIZE"])]
json.dump(rows, open(f"{out}/summary.json", "w"), indent=1)
for r in rows[:12]:
    print(f"{r['kernel'][:44]:44s} calls={r['calls']:5d} fetch/call={r['fetch_KiB_per_call']:10.1f} KiB write/call={r['write_KiB_per_call']:10.1f} KiB  total fetch={r['fetch_KiB_total']/1024:9.1f} MiB write={r['write_KiB_total']/1024:9.1f} MiB")
