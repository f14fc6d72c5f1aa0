"""Peak device memory and step time of the hot path with / without activation-buffer reuse: python tools/mem_plan.py [B]"""
import os, sys, subprocess, json
B = sys.argv[1] if len(sys.argv) > 1 else "32"
for reuse in ("1", "0"):
    code = f"""
import os, sys, json, runpy, torch
sys.path.insert(0, os.getcwd())
from implicit_depth_amd import nhwc
nhwc.BUFFER_REUSE = {reuse == '1'}
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-split-line', '--no-extras', '--batch', '{B}', '--steps', '20']
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
print('PEAK_GB', torch.cuda.max_memory_allocated() / 2**30)
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    peak = [l for l in out.splitlines() if l.startswith("PEAK_GB")][-1]
    print(f"B={B} reuse={reuse}: {d['value']:.1f} frames/s  {d['ms_per_step']:.3f} ms/step  {peak}")
