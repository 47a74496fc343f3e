#!/usr/bin/env python3
"""Developer tool: copy the summaries of tools/profile_round.sh + tools/bench_set.sh (+ an 8 192-env kstats run) from gpurun_out/ into
profiles/ (our kernels only).  Usage: tools/publish_profiles.py <round tag> <bench_set tag> [<kstats tag>]"""
import csv, glob, json, os, shutil, sys
tag, bset = sys.argv[1], sys.argv[2]
ks = sys.argv[3] if len(sys.argv) > 3 else None
OURS = ("k_", "parcdyn", "void k_env_post")
src = f"gpurun_out/prof_{tag}/summary"
for f in os.listdir(src):
    if f.endswith(".csv"):
        rows = list(csv.reader(open(os.path.join(src, f))))
        csv.writer(open("profiles/" + f, "w")).writerows([rows[0]] + [r for r in rows[1:] if any(k in r[0] for k in ("k_", "rocclr"))])
    else:
        shutil.copy(os.path.join(src, f), "profiles/" + f)
p = f"profiles/{tag}_pmc_hbm_traffic.json"
d = json.load(open(p))
for k in ("FETCH_SIZE_KiB_avg_per_dispatch", "WRITE_SIZE_KiB_avg_per_dispatch"):
    d[k] = {q: v for q, v in d[k].items() if q.startswith(OURS)}
json.dump(d, open(p, "w"), indent=1)
p = f"profiles/{tag}_cfg5_shard.json"
d = json.load(open(p))
d["per_kernel_avg_per_dispatch"] = {q: v for q, v in d["per_kernel_avg_per_dispatch"].items() if q.startswith(OURS)}
json.dump(d, open(p, "w"), indent=1)
names = {"headline": "bench", "kinematic_65536": "bench_kinematic_65536", "cfg3": "bench_cfg3", "cfg5_shard": "bench_cfg5_shard",
         "shard_8192": "bench_shard_8192", "two_ranks_one_gpu": "bench_two_ranks_one_gpu"}
for k, v in names.items():
    json.dump(json.load(open(f"gpurun_out/bench_set_{bset}/{k}.json")), open(f"profiles/{tag}_{v}.json", "w"), indent=1)
if ks:
    f = max(glob.glob(f"gpurun_out/kstats_{ks}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)  # the directory keeps earlier runs
    rows = list(csv.reader(open(f)))
    csv.writer(open(f"profiles/{tag}_shard_8192_kernel_stats.csv", "w")).writerows([rows[0]] + [r for r in rows[1:] if any(k in r[0] for k in ("k_", "rocclr"))])
d = json.load(open(f"profiles/{tag}_pmc_dynamics.json"))
print(json.dumps(d["derived"]))
print({k: d[k] for k in ("SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB")})
h = json.load(open(f"profiles/{tag}_bench.json")); r = h["roofline"]
print(h["value"], h["ms_per_step"], r["kernel_ms"], r["achieved"], r["frac"], r["obs_kernel"]["kernel_ms"], r["obs_kernel"]["achieved"], r["obs_kernel"]["frac"],
      r["obs_kernel"]["traffic"], h["cpu_baseline"]["value"], h["cpu_baseline"]["torch_path"]["value"])
for r in list(csv.reader(open(f"profiles/{tag}_bench_kernel_stats.csv")))[1:9]:
    print(r[0][:50], r[1], round(float(r[3]) / 1e3, 1))
