#!/usr/bin/env python3
"""gpurun_out/<tags> (tools/round_artifacts.sh, tools/mode_artifacts.sh) -> the tracked files under profiles/.
usage: python3 tools/collect_profiles.py r03 <round_artifacts tag> [<k9 mode tag> [<l5000 mode tag>]]
Per kernel the algorithmic bytes of one launch (every byte the kernel's job requires, once) next to the counter traffic
(FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md) and the kernel trace's average duration."""
import csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix, art = sys.argv[1], sys.argv[2]
modes = dict(zip(("k9", "l5000"), sys.argv[3:5]))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def algorithmic(kernel, d):
    """bytes per launch by the kernel's job: n_ops, kept events / samples, reads of the bench line d"""
    c = d["config"]
    n_ops, ke, ks, reads, smp = c["ss_ops_per_gpu"], c["kept_events_rank0"], c["kept_samples_rank0"], c["reads_per_gpu"], c["samples_per_gpu"]
    k = kernel.replace("void ", "")
    if k.startswith("k_read_stats") and "rare" not in k: return 2 * smp + 56 * reads          # the signal once + per-read scalars in, median / MAD out
    if k.startswith("k_gather"): return 10 * ks + 56 * ke                                        # 2 B in + 8 B out per sample; record 16 + offset 8 + calibration 32 per event
    if k.startswith("k_events"): return 10 * n_ops                                               # op_n 4 + op_t 1 + base 1 in, slot word 4 out
    if k.startswith("k_part_scatter"): return 8 * n_ops + 18 * ke                                # slot word + op_n per op in, 16 B element + 2 B low digit per accepted event out (~ kept at this limit)
    if k.startswith("k_region_place"): return 32 * ke                                            # element in, record out
    if k.startswith("k_region_count"): return 2 * ke
    if k.startswith("k_rank_emit2"): return 4 * n_ops + 16 * ke                                  # slot word per op in, record per kept event out
    return None


def traffic_table(tag, name):
    src = os.path.join(G, tag)
    d = last_json(os.path.join(src, "bench.json"))
    pmc = json.load(open(os.path.join(src, "pmc_fetch_write_kb.json")))
    avg = {}
    for row in csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))):
        avg[row["Name"].split("(")[0].strip()] = float(row["AverageNs"])
    kernels = {}
    for kn in sorted(pmc["FETCH_SIZE"], key=lambda k: -avg.get(k, 0)):
        f, w = pmc["FETCH_SIZE"][kn] * 1024, pmc["WRITE_SIZE"].get(kn, 0) * 1024
        us = avg.get(kn, 0) / 1e3
        e = {"fetch_bytes_x2": 2 * f, "write_bytes": w, "avg_us": us, "GBs_on_counter_traffic": ((2 * f + w) / (us * 1e-6) / 1e9) if us else None}
        a = algorithmic(kn, d)
        if a:
            e["algorithmic_bytes"] = a; e["traffic_over_algorithmic"] = (2 * f + w) / a
            e["frac_of_8TBs_on_algorithmic_bytes"] = (a / (us * 1e-6) / 8e12) if us else None
        kernels[kn] = e
    out = {"workload": d["config"]["workload"], "ms_per_step": d["ms_per_step"], "whole_step_frac": d["whole_step_frac"],
           "kept_events": d["config"]["kept_events_rank0"], "kept_samples": d["config"]["kept_samples_rank0"],
           "note": "per kernel: FETCH_SIZE x 2 (gfx950 correction for wide streaming reads: an upper bound for narrow or scattered ones) + WRITE_SIZE per launch, "
                   "next to the kernel trace's average duration (tools/mode_artifacts.sh; the profiled passes run on one stream, bench.py --one-stream). 'void' = the "
                   "k_slot_model kernels of bench.py's model pass, k_unpack_recs / reduce_kernel / copyBuffer = its result download: not part of the step",
           "kernels": kernels}
    json.dump(out, open(os.path.join(P, f"{prefix}_{name}_pmc_traffic.json"), "w"), indent=1)
    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(P, f"{prefix}_{name}_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(P, f"{prefix}_{name}_bench.json"))
    print(name, "ms/step %.4f frac %.3f" % (d["ms_per_step"], d["whole_step_frac"]))


src = os.path.join(G, art)
d = last_json(os.path.join(src, "bench.json"))
pmc = json.load(open(os.path.join(src, "pmc_fetch_write_kb.json")))
f_kb, w_kb = pmc["FETCH_SIZE"]["k_read_stats"], pmc["WRITE_SIZE"]["k_read_stats"]
alg = d["roofline"]["bytes_per_launch"]
tr = (2 * f_kb + w_kb) * 1024
json.dump({"kernel": "k_read_stats", "FETCH_SIZE_KB_avg": f_kb, "WRITE_SIZE_KB_avg": w_kb, "gfx950_fetch_correction": 2.0, "traffic_bytes_per_launch": tr,
           "algorithmic_bytes_per_launch": alg, "ratio": tr / alg,
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/round_artifacts.sh, {prefix} build (profiles/{prefix}_kernel_stats.csv is the "
                     "kernel trace of the same build), " + d["config"]["workload"] + "; the profiled passes run bench.py --one-stream",
           "all_kernels_KB": pmc}, open(os.path.join(P, f"{prefix}_pmc_traffic.json"), "w"), indent=1)
for a, b in (("kernel_stats.csv", "kernel_stats.csv"), ("kernel_stats_two_streams.csv", "kernel_stats_two_streams.csv"), ("bench.json", "bench.json"),
             ("trace_bench.json", "trace_bench.json"), ("pytest_gpu.txt", "pytest_gpu.txt")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(P, f"{prefix}_{b}"))
print("headline ms/step %.4f frac %.3f, k_read_stats %.1f us = %.3f, counter traffic x%.3f" % (d["ms_per_step"], d["whole_step_frac"], d["roofline"]["avg_launch_ms"] * 1e3, d["roofline"]["frac"], tr / alg))
for name, tag in modes.items():
    traffic_table(tag, name)
