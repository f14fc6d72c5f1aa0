#!/bin/bash
# sample sclk / power while the bench runs: tools/clock_probe.sh <math>
M=${1:-fp32}
( for i in $(seq 1 60); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|Average Graphics" | tr '\n' ' '; echo; sleep 0.25; done ) > /tmp/clk_$M.log &
SP=$!
python bench.py --no-cpu-baseline --no-split-line --steps 120 --warmup 10 --math $M | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$M fps', round(d['value'],1))"
kill $SP 2>/dev/null
python - <<PY
import re
rows=[l for l in open("/tmp/clk_$M.log") if "sclk" in l]
clk=[int(m.group(1)) for l in rows for m in [re.search(r"sclk.*?\((\d+)Mhz\)", l)] if m]
pw=[float(m.group(1)) for l in rows for m in [re.search(r"Power \(W\): ([0-9.]+)", l)] if m]
print("$M samples", len(clk), "sclk MHz min/median/max", (min(clk), sorted(clk)[len(clk)//2], max(clk)) if clk else None, "power W median/max", (sorted(pw)[len(pw)//2], max(pw)) if pw else None)
print(rows[len(rows)//2].strip()[:200] if rows else "no rows")
PY
