"""Summaries of a tools/profile_round.sh run -> <out>/summary/*.json|csv (copy the ones to be judged into profiles/)."""
import collections, csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from parc_amd import lib as L
out, tag = sys.argv[1], sys.argv[2]
BUILD = {"csrc_sha16": L.csrc_hash(), "build_flags": L.load().parc_build_flags().decode()}  # what the counters were collected on
os.makedirs(out + "/summary", exist_ok=True)


def pmc(d):
    fs = glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in fs:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    return {k: {c: v / cnt[(k, c)] for c, v in cs.items()} for k, cs in acc.items()}


st = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
if st:
    shutil.copy(st[0], out + f"/summary/{tag}_bench_kernel_stats.csv")
if os.path.exists(out + "/bench_under_profiler.json"):
    shutil.copy(out + "/bench_under_profiler.json", out + f"/summary/{tag}_bench_under_profiler.json")
fe, wr = pmc("fetch"), pmc("write")
json.dump({**BUILD, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), bench.py --steps 20 (default config: dynamics on, 65536 envs). "
                   "Units: KiB per dispatch; gfx950 FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md HBM section): consumers double it.",
           "FETCH_SIZE_KiB_avg_per_dispatch": {k: v["FETCH_SIZE"] for k, v in fe.items() if "FETCH_SIZE" in v},
           "WRITE_SIZE_KiB_avg_per_dispatch": {k: v["WRITE_SIZE"] for k, v in wr.items() if "WRITE_SIZE" in v}},
          open(out + f"/summary/{tag}_pmc_hbm_traffic.json", "w"), indent=1)
dk = "parcdyn::k_dynamics_wave"
sq = {}
for d in ("sq1", "sq2", "sq3", "sq4"):
    sq.update(pmc(d).get(dk, {}))
if sq:
    waves = sq.get("SQ_WAVES", 0.0) or 1.0
    der = {"valu_instructions_per_wave": sq.get("SQ_INSTS_VALU", 0) / waves,
           "valu_active_fraction_of_wave_time": sq.get("SQ_ACTIVE_INST_VALU", 0) / max(sq.get("SQ_WAVE_CYCLES", 1), 1),
           "any_instruction_active_fraction": sq.get("SQ_ACTIVE_INST_ANY", 0) / max(sq.get("SQ_WAVE_CYCLES", 1), 1),
           "wait_fraction_of_wave_time_incl_barriers": sq.get("SQ_WAIT_ANY", 0) / max(sq.get("SQ_WAVE_CYCLES", 1), 1),
           "lane_utilisation_of_valu": sq.get("SQ_THREAD_CYCLES_VALU", 0) / max(64.0 * sq.get("SQ_ACTIVE_INST_VALU", 1), 1),
           "icache_miss_rate": sq.get("SQC_ICACHE_MISSES", 0) / max(sq.get("SQC_ICACHE_REQ", 1), 1)}
    rec = {"kernel": "k_dynamics_wave", "config": "bench.py default config (65536 envs, dynamics on), --steps 20",
           "note": "rocprofv3 --pmc, four counters per pass, --kernel-trace only, averages per dispatch. SQ_* cycle counters are per-wave quad-cycles summed over "
                   "waves; 4096 waves = 1024 blocks x 4 waves, 1 wave per SIMD.",
           "FETCH_SIZE_KiB": fe.get(dk, {}).get("FETCH_SIZE"), "WRITE_SIZE_KiB": wr.get(dk, {}).get("WRITE_SIZE")}
    rec.update(sq); rec["derived"] = der; rec.update(BUILD)
    json.dump(rec, open(out + f"/summary/{tag}_pmc_dynamics.json", "w"), indent=1)
c5f, c5w = pmc("c5fetch"), pmc("c5write")
if c5f:
    n = 16384
    rows = {}
    for k in c5f:
        f_ = c5f[k].get("FETCH_SIZE", 0.0); w_ = c5w.get(k, {}).get("WRITE_SIZE", 0.0)
        rows[k] = {"FETCH_SIZE_KiB": f_, "WRITE_SIZE_KiB": w_, "hbm_bytes_per_env_step": (2.0 * f_ + w_) * 1024.0 / n}
    json.dump({"config": "cfg 5 shard: bench.py --envs 16384 --motions 16384 --yaw 1 (16 384 envs = 131 072 / 8 on the 16 384-pseudo-clip library, "
                         "frame records + grid > 256 MB Infinity Cache), dynamics on", "note": "FETCH doubled per the gfx950 correction in hbm_bytes_per_env_step; "
                         "SURVEY 8(d) bound for the table-miss case: <= 11 732 B per env-step for the obs path", "per_kernel_avg_per_dispatch": rows},
              open(out + f"/summary/{tag}_cfg5_shard.json", "w"), indent=1)
print(sorted(os.listdir(out + "/summary")))
