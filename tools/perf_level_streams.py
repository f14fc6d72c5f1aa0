"""Small batch: do the launch units of ONE dependency level overlap when they go out on two HIP streams?
python tools/perf_level_streams.py [B]   (conv plan of the hot path = CVEncoder + decoder segment, replayed level by level)

mode 0: plan.run over the segment (one C call, one stream)              = the shipped path
mode 1: level by level from Python, one stream                           = what the Python loop itself costs
mode 2: level by level; levels with >= 2 launch classes run the first class on the main stream and the rest on a
        side stream between fork / join events
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
from implicit_depth_amd import nhwc
from bench import HotPathWorkload

a = argparse.Namespace(batch=int(sys.argv[1]) if len(sys.argv) > 1 else 4, views=7, planes=64, height=384, width=512, volume="mlp")
wl = HotPathWorkload(a, torch.device("cuda"), 0)
for _ in range(2): wl.step()
torch.cuda.synchronize()
ent = next(iter(wl.model._plans.values()))
p, n_head = ent["plan"], ent["n_head_ops"]
n = len(p.ops)
# launch classes inside a level, in the scheduler's order (Plan._launch_rank): F(2x2) group | 4-row LDS group (+ level_k members) | the rest one by one
levels = []
i = n_head
while i < n:
    j = i
    while j < n and p.levels[j] == p.levels[i]: j += 1
    cls, k = [], i
    def key(q):
        r = nhwc.Plan._launch_rank(p.ops[q])[0]
        return "w" if r == -1 else "g" if r < 3 else f"s{q}"
    while k < j:
        m = k + 1
        while m < j and key(m) == key(k): m += 1
        cls.append((k, m)); k = m
    levels.append(cls); i = j
multi = sum(1 for c in levels if len(c) > 1)
print(f"B={a.batch}: {len(levels)} levels in the conv segment, {multi} with more than one launch class")
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
ev_fork = [torch.cuda.Event() for _ in levels]
ev_join = [torch.cuda.Event() for _ in levels]


def run(mode):
    if mode == 0:
        p.run(n_head, n); return
    for li, cls in enumerate(levels):
        if mode == 1 or len(cls) == 1:
            p.run(cls[0][0], cls[-1][1]); continue
        # the biggest class stays on the main stream
        ev_fork[li].record(main)
        side.wait_event(ev_fork[li])
        with torch.cuda.stream(side):
            p.run(cls[1][0], cls[-1][1])
            ev_join[li].record(side)
        p.run(cls[0][0], cls[0][1])
        main.wait_event(ev_join[li])


def timeit(mode, reps=20):
    for _ in range(3): run(mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run(mode)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ref = None
for rnd in range(3):
    t = [timeit(m) for m in (0, 1, 2)]
    print(f"round {rnd}: one call {t[0]:.3f} ms | per level, one stream {t[1]:.3f} | two streams {t[2]:.3f}")
