"""Copy the summaries of a tools/final_profiles.sh run (gpurun_out/<tag>/...) into profiles/<round>/, refresh profiles/pmc_traffic.json and the
bench-line table of profiles/<round>/README.md:  python tools/collect_profiles.py <tag> [round = r04]"""
import glob, json, os, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
S, D = f"gpurun_out/{tag}", f"profiles/{sys.argv[2] if len(sys.argv) > 2 else 'r04'}"
os.makedirs(f"{D}/matrix", exist_ok=True)
for f in glob.glob(f"{S}/matrix/*.json"):
    shutil.copy(f, f"{D}/matrix/")
shutil.copy(f"{S}/bench_default.json", f"{D}/bench_hot_path_mlp_k7_gb32.json")
for src, dst in (("prof_hot", "hot_path_mlp_gb32_kernel_stats.csv"), ("prof_dot", "warp_match_dot_gb32_kernel_stats.csv"), ("prof_temporal", "temporal_b1_d96_kernel_stats.csv")):
    fs = glob.glob(f"{S}/{src}/**/*kernel_stats.csv", recursive=True)
    if fs:
        shutil.copy(fs[0], f"{D}/{dst}")
shutil.copy(f"gpurun_out/pmc_{tag}_hot/summary.json", f"{D}/pmc_hbm_hot_path_mlp_b32.json")
shutil.copy(f"gpurun_out/pmc_{tag}_dot/summary.json", f"{D}/pmc_hbm_warp_match_dot_b32.json")
shutil.copy(f"gpurun_out/pmc_mfma_{tag}/summary.json", f"{D}/pmc_mfma_util_hot_path_mlp_b32.json")
open(f"{D}/pmc_cv_dot_win_b32.txt", "w").write("".join(l for l in open(f"{S}/pmc_dot_win.txt") if "amdgpu.ids" not in l and l[:1].isupper()))
bench = json.loads(open(f"{D}/bench_hot_path_mlp_k7_gb32.json").read().strip().splitlines()[-1])
pt = json.load(open("profiles/pmc_traffic.json"))
rows = json.load(open(f"{D}/pmc_hbm_hot_path_mlp_b32.json"))
dom = [r for r in rows if r["kernel"] == bench["roofline"]["kernel"]]
rnd = int("".join(ch for ch in D.split("/")[-1] if ch.isdigit()) or 0)
if dom:
    # (notes are regenerated from the data: launches per step = profiled launches / the 4 steps (3 timed + 1 warm-up) x the roofline leg's replays are excluded by --no-extras? no:
    # the conv replays of bench.py's roofline leg run in the profiled process too, so only the per-launch mean is quoted)
    fam = {r["kernel"]: r["calls"] for r in rows if r["kernel"].startswith("conv3x3_wino4_k")}
    pt["hot_path/mlp/b32"].update(kernel=dom[0]["kernel"], launches_profiled=dom[0]["calls"], round=rnd,
                                  traffic_bytes_per_launch=(2 * dom[0]["fetch_KiB_per_call"] + dom[0]["write_KiB_per_call"]) * 1024,
                                  note=f"mean over the {dom[0]['calls']} profiled launches of {dom[0]['kernel']} ({bench['roofline']['launches_per_step']} per step); "
                                       f"profiled launches of the three F(4x4) instances: {fam}; rows of {D}/pmc_hbm_hot_path_mlp_b32.json")
rows = json.load(open(f"{D}/pmc_hbm_warp_match_dot_b32.json"))
w = [r for r in rows if r["kernel"].startswith("cv_dot_win_k")]
if w:
    pt["warp_match_dot/b32"].update(kernel=w[0]["kernel"], launches_profiled=w[0]["calls"], round=rnd,
                                    traffic_bytes_per_launch=(2 * w[0]["fetch_KiB_per_call"] + w[0]["write_KiB_per_call"]) * 1024,
                                    note="cv_dot_win_k alone (its run-list pre-pass cv_runs_k and the arg-max combine cv_argmax_partials_k are separate rows of the same file)")
json.dump(pt, open("profiles/pmc_traffic.json", "w"), indent=1)
lines = []
for f in sorted(glob.glob(f"{D}/matrix/*.json")):
    d = json.loads(open(f).read())
    r = d["roofline"]
    lines.append(f"| `{os.path.basename(f)}` | {d['value']:.1f} | {d['ms_per_step']:.3f} | `{r['kernel'][:48]}` | {r['achieved']:.1f} {r['unit']} | {r['frac']:.3f} |"
                 + (f" {r['achieved_algorithmic']:.1f} |" if "achieved_algorithmic" in r else " |"))
s = open(f"{D}/README.md").read()
a = s.index("| bench line | frames/s |")
s = s[:a] + "| bench line | frames/s | ms/step | roofline kernel | achieved | frac | algorithmic |\n|---|---|---|---|---|---|---|\n" + "\n".join(lines) + \
    "\n\n(B=1 rows: the slower of the two timed kernel families is the 4-row direct family — grouped and level launches — which the roofline object then names.)\n"
open(f"{D}/README.md", "w").write(s)
r = bench["roofline"]
print(f"bench {bench['value']:.1f} fps {bench['ms_per_step']:.2f} ms; {r['kernel']} x{r['launches_per_step']} {r['kernel_ms']*1e3:.1f} us frac {r['frac']:.3f} alg {r['achieved_algorithmic']:.1f}; conv {r['all_conv_kernels']['ms_per_step']:.2f} ms")
wm = bench["warp_match"]
print(f"warp_match {wm['kernel_ms']:.4f} ms frac {wm['frac']:.4f}; temporal {[round(v['value'],1) for v in bench['temporal']['sequences_per_gpu'].values()]}; k8 {bench['k8']['value']:.1f}; "
      f"fv {bench['fv_mlp']['ms']:.2f} ms; cpu {bench['cpu_baseline']['value']:.3f}; extra {[(k, round(v['value'], 1)) for k, v in bench.get('extra', {}).items()]}")
