#!/usr/bin/env python3
"""bench.py -- the gmove hot path on N MI355X GPUs (BASELINE.json metric: signal samples/s into k-mer buckets).

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
walk/filter events -> deterministic per-k-mer ranking -> (N>1: RCCL all-gather of per-k-mer counts) ->
sample_limit cut -> med-MAD statistics of every read -> gather of the kept windows. The headline workload at N=1 is
BASELINE.json configs[1]: synthetic RNA004, 50 000 reads x 4 000 samples, k=5, --rna --scaling 1,
min/max_dur 20/40, sample_limit 100, all 1024 k-mers. At N>1 every rank holds its own 50 000-read shard of
one PAF-ordered job (weak scaling); value = samples of all ranks / max-over-ranks time. `config.workload` names the
BASELINE config the chosen flags are (or says "custom").

`python bench.py --gpus N` with N > 1 and no torch.distributed environment starts its own
`python -m torch.distributed.run --nproc-per-node N` child (before anything touches a GPU) and relays its JSON line.

Prints ONE JSON line (rank 0). Objects besides the contract's fields:
  roofline      k_read_stats, the kernel that streams the signal (dispatch time stamps on the library's stream)
  whole_step    SURVEY 8(d)'s algorithmic bytes of the COMPLETE step over the step time
  useful        which of the streamed samples the reference would have read at all: it stops at the read that completes
                the last k-mer (src/gmove.cpp:733-735); per rank `samples` next to `useful_samples`
  config3_mode  (N=1) BASELINE configs[3]: 50 000 DNA reads, k = 9 (262 144 k-mers), sample_limit 1000: ms/step, whole-step
                fraction of the HBM roofline, per-kernel times
  all_kept_mode (N=1) the headline reads at sample_limit 5000 (configs[2]'s limit: every accepted event is kept)
  ragged_mode   (N=1) DNA reads of lognormal length 2 000 .. 200 000 samples + 0.1 % of 10^6, the same 2 x 10^8 samples, k = 9 and k = 5:
                k_read_stats' share of the roofline with the long reads split across waves, and with one wave per read
  ms_per_step_blocks  three timed blocks (min / median / max) for value, one_stream_mode, config3_mode, all_kept_mode
  config2_mode  (N>1) BASELINE configs[2]: the same N x 50 000 reads as ONE job at sample_limit 5000, same weak-scaling step
  job_layer     (N>1) the step through pg_job_* (one process, N host threads, ncclCommInitAll), the path `poregen gmove
                --devices` uses; fed from host memory, so PCIe-inclusive and never `value`
  hbm_not_mall, pcie_inclusive, end_to_end, lazy_statistics_mode, one_stream_mode, kmer_model_once_per_job, cpu_baseline (N=1)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def relaunch_under_torchrun(args):
    """--gpus N > 1 without a torch.distributed environment: one rank per GPU as a CHILD process group (never an exec of
    this process, and nothing here has touched a GPU yet); the ranks' single JSON line is relayed."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line)
    raise SystemExit(r.returncode if r.returncode else (0 if line else 1))


def workload_label(kind, reads, read_len, k, limit, world):
    """Which BASELINE.json config the flags are."""
    shape = f"synthetic {kind} SLOW5+PAF, {reads} reads x {read_len} samples per GPU, k={k}, scaling med-MAD, sample_limit={limit}"
    shape += ", min/max_dur 20/40, --rna" if kind == "rna004" else ""
    std = reads == 50000 and read_len == 4000
    if std and kind == "rna004" and k == 5 and limit == 100:
        name = "BASELINE configs[1]" + (f" as a weak-scaling shard per GPU (x{world})" if world > 1 else "")
    elif std and kind == "rna004" and k == 5 and limit == 5000:
        name = "BASELINE configs[2]" + (f" ({world} of its 8 shards)" if world != 8 else "") if world > 1 else "BASELINE configs[2]'s per-GPU shard (one of its 8)"
    elif std and kind == "dna_r10" and k == 9 and limit == 1000 and world == 1:
        name = "BASELINE configs[3]"
    else:
        name = "custom (no BASELINE config)"
    return f"{name}: {shape}"


def b_alg(n_samples, n_reads, n_ops, n_bases, kept_samples, kept_events, n_slots):
    """every input byte once, every output byte once."""
    return 2 * n_samples + 24 * n_reads + 5 * n_ops + n_bases + 8 * kept_samples + 12 * kept_events + 8 * n_slots


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pre-warm", type=int, default=0, help="extra untimed settling steps in front of --warmup (round 5 ran 300 by default; now opt-in, and counted in the reported `warmup`)")
    ap.add_argument("--reads", type=int, default=50000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=4000)
    ap.add_argument("--kind", default="rna004")
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--sample-limit", type=int, default=100)
    ap.add_argument("--lazy", action="store_true", help="statistics only for reads that own a kept event")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed profile instead of two rocprofv3 --pmc child passes of this command (~1 min)")
    ap.add_argument("--no-lazy-extra", action="store_true", help="skip the extra lazy-statistics / two-stream measurements (profiling runs)")
    ap.add_argument("--no-extras", action="store_true", help="skip config3_mode / all_kept_mode / config2_mode / job_layer / hbm_not_mall / pcie_inclusive / end_to_end (profiling and A/B runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-dist", action="store_true", help="N=1 only: run the multi-GPU step (count, RCCL all_gather, collect) on a one-rank group, to price its overhead")
    ap.add_argument("--defer", action="store_true", help="multi-GPU step: the statistics on the chain's stream, queued behind the ISSUE of the all_gather (PG_FLAG_DEFER_STATS), where they hide "
                    "the collective; default since round 5: on the second stream as at N=1 (one-rank RCCL step 0.169 -> 0.152-0.156 ms; an exposed collective of up to ~70 us costs less than the stream does)")
    ap.add_argument("--no-defer", action="store_true", help="(the default since round 5; kept for old command lines)")
    ap.add_argument("--lib", default=None, help="measurement builds: path of another libpgmove build to load instead of poregen_amd/libpgmove.so")
    ap.add_argument("--overlap-tail", action="store_true", help="statistics on a second stream next to the cut/emit/scan launches (PG_FLAG_OVERLAP_TAIL)")
    ap.add_argument("--split-walk", action="store_true", help="ss walk and event filter as two launches (PG_FLAG_DEBUG_SPLIT_WALK), for comparison")
    ap.add_argument("--overlap", action="store_true", help="statistics of a batch on a second stream (PG_FLAG_OVERLAP; the library's default since round 3 wherever it applies)")
    ap.add_argument("--one-stream", action="store_true", help="every kernel of a batch on one stream (PG_FLAG_ONE_STREAM): round 2's default")
    ap.add_argument("--homopolymer-frac", type=float, default=None, help="share of reads with a 40-base homopolymer run (default: 0.1 for dna_r10 = SURVEY 8d cfg 3, else 0)")
    ap.add_argument("--job-layer-child", type=int, default=0, help=argparse.SUPPRESS)  # internal: run the pg_job_* step over this many devices and print its JSON object
    args = ap.parse_args()
    if args.job_layer_child:
        return job_layer_child(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # libraries print to stdout too (RCCL's version banner at the first collective): until the JSON line is due, file descriptor 1
    # is the process's stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    if args.lib:
        from poregen_amd import _abi
        _abi.LIB_PATH = os.path.abspath(args.lib)
    from poregen_amd import dist as pgdist
    from poregen_amd import synth
    from poregen_amd.engine import GmoveEngine, GmoveParams, generate_kmers

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("PG_BENCH_BACKEND", "nccl")  # "gloo": rehearse N>1 on fewer GPUs (ranks share devices)
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_step = world > 1 or args.force_dist
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    elif args.force_dist:
        import tempfile
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="file://" + tempfile.mktemp(prefix="pg_rdv_"), rank=0, world_size=1, device_id=dev)

    rna = args.kind == "rna004"
    p = dict(kmer_size=args.k, rna=rna, scaling=1, sample_limit=args.sample_limit, device=local_rank, lazy_stats=args.lazy, overlap=True if args.overlap else (False if args.one_stream else None), split_walk=args.split_walk, overlap_tail=args.overlap_tail)
    if rna:
        p.update(min_dur=20, max_dur=40)
    kmers = generate_kmers(args.k, rna=rna)

    t0 = time.time()
    hp_frac = args.homopolymer_frac if args.homopolymer_frac is not None else (0.1 if args.kind == "dna_r10" else 0.0)
    host = synth.make_batch_fast(args.reads, read_len=args.read_len, kind=args.kind, seed=20251003 + 1 + 1000 * rank, homopolymer_frac=hp_frac)
    gen_s = time.time() - t0
    shard = host.to_device(dev)
    torch.cuda.synchronize()
    shard.resident = True  # uploaded once, complete: the statistics of a step need not wait for what the step's stream still holds (PG_BATCH_RESIDENT)
    n_samples = host.n_samples
    n_ops = int(host.op_off[-1])
    n_bases = int(host.seq_off[-1])

    stream_ordered = dist_step and backend == "nccl"
    side = None
    if stream_ordered:  # count -> RCCL all_gather -> collect ordered on one stream, no host sync inside a step
        side = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(side)
    # multi-GPU step: the statistics are queued between the issue of the all_gather and the wait for it (dist.sharded_step)
    # (the gloo rehearsal synchronises the host between count and collect: statistics queued by pg_count would be waited for there)
    defer = dist_step and (args.defer or backend == "gloo") and not args.no_defer and not args.lazy and not args.overlap

    def timed(params, steps, warmup, extra_blocks=0):
        """warmup untimed steps, then `steps` timed ones bracketed by barrier + synchronize on both sides, MAX over ranks."""
        eng = GmoveEngine(GmoveParams(kmers=kmers, defer_stats=defer, **params))
        if stream_ordered:
            eng.use_torch_stream(side)
        gather_buf = torch.empty(world * len(kmers), dtype=torch.int64, device=dev)  # receive buffer of the per-step all_gather
        # pg_count writes this rank's counts straight into its row of the receive buffer: the all_gather runs in place
        counts_buf = gather_buf[rank * len(kmers):(rank + 1) * len(kmers)] if backend == "nccl" else torch.empty(len(kmers), dtype=torch.int64, device=dev)
        state = {}

        def step():
            eng.reset()
            if dist_step:
                total = pgdist.sharded_step(eng, shard, counts_buf=counts_buf, stream_ordered=stream_ordered, gather_buf=gather_buf)
                state["freq"] = pgdist.merged_freq(total, params["sample_limit"], engine=eng)
            else:
                eng.submit(shard)

        def fence():
            eng.sync()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()

        # The contract: W untimed steps, then exactly K timed ones. (Round 5 put 300 settling steps in front by default -- the first block of
        # a run reads ~5 % slower than the blocks behind it, profiles/r05_bench_blocks.txt; that is now opt-in, --pre-warm N, and the
        # reported `warmup` is the number of untimed steps actually run. The settled figure is what the two blocks behind the contract's
        # block show: ms_per_step_blocks.)
        for _ in range(args.pre_warm):
            step()
        if args.pre_warm:
            fence()
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        enqueue_ms = (time.perf_counter() - t0) / steps * 1e3  # host time to queue a step: a step cannot be faster than this
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # two more blocks of the same K steps, reported beside the contract's block (`value` stays the first block: EXACTLY K timed steps):
        # boxes, and minutes on one box, differ by several per cent
        extra = []
        for _ in range(extra_blocks):
            tb = time.perf_counter()
            for _ in range(steps):
                step()
            fence()
            d2 = time.perf_counter() - tb
            if world > 1:
                t = torch.tensor([d2], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                d2 = float(t.item())
            extra.append(d2 / steps * 1e3)
        state["extra_blocks_ms"] = extra
        return eng, dt / steps * 1e3, enqueue_ms, state

    def useful_of(eng, state, limit):
        """What the reference would have read: it stops at the read that completes the last k-mer of the WHOLE list (gmove.cpp:733-735).
        Per rank: samples streamed, samples of the reads up to and including the job's completing read."""
        counts, ev_len, _, ev_read = eng.kept_tensors(device=dev, with_reads=True)
        n_kept = int(ev_len.numel())
        last_read = int(ev_read.max().item()) if n_kept else -1
        if dist_step:
            freq = state["freq"]
            full = bool((freq >= limit).all().item()) if limit > 0 else False
        else:
            full = eng.all_slots_full()
        mine = [n_kept, last_read, host.n_reads, int(n_samples)]
        if world > 1:
            allr = [None] * world
            dist.all_gather_object(allr, mine)
        else:
            allr = [mine]
        per_rank = []
        if full:
            r_star = max(g for g in range(world) if allr[g][0] > 0)  # the last shard that still placed an event
        for g in range(world):
            kept_g, last_g, reads_g, samples_g = allr[g]
            if not full or g < r_star:
                u = samples_g
            elif g == r_star:
                u = int(host.sig_off[last_g + 1]) if g == rank else (last_g + 1) * args.read_len
            else:
                u = 0
            per_rank.append({"rank": g, "samples": samples_g, "useful_samples": u, "kept_events": kept_g})
        tot, use = sum(x["samples"] for x in per_rank), sum(x["useful_samples"] for x in per_rank)
        return {"all_kmers_complete": full, "samples": tot, "useful_samples": use, "useful_fraction": use / tot if tot else None, "per_rank": per_rank}

    eng, ms_per_step, enqueue_ms, state = timed(p, args.steps, args.warmup, extra_blocks=2)
    value = world * n_samples / (ms_per_step * 1e-3)
    all_blocks = sorted([ms_per_step] + state["extra_blocks_ms"])
    useful = useful_of(eng, state, args.sample_limit)

    # ---- per-kernel times: the dispatches' own time stamps (PG_FLAG_PROFILE), separate untimed passes ----------
    prof = GmoveEngine(GmoveParams(kmers=kmers, profile=True, **p))
    for _ in range(2):
        prof.reset(); prof.submit(shard)
    prof.sync(); prof.kernel_stats_reset()
    n_prof = 20
    for _ in range(n_prof):
        prof.reset(); prof.submit(shard)
    prof.sync()
    ks = prof.kernel_stats()
    res = prof.finish()
    kept_events = int(res.counts.sum()); kept_samples = int(res.samples.size)
    # the k-mer model reduction (pg_model) over this batch's kept samples: once per JOB, not part of the step
    try:
        for _ in range(2):
            prof.model()                       # first launch loads the code object
        prof.kernel_stats_reset()
        for _ in range(5):
            mdl = prof.model()
        km = prof.kernel_stats()["k_slot_model"]
        model_ms = km[1] / km[0]
        model_info = {"kernel": "k_slot_model", "avg_launch_ms": model_ms, "values": int(mdl.n_values.sum()),
                      "algorithmic_bytes": 8 * kept_samples + 4 * kept_events, "GB/s": (8 * kept_samples + 4 * kept_events) / (model_ms * 1e-3) / 1e9}
    except Exception as e:  # (a timing-probe build of the library produces garbage values: the model refuses them)
        if not args.lib:
            raise
        model_info = {"error": str(e)}
    prof.close()
    stats_ms = ks["k_read_stats"][1] / ks["k_read_stats"][0]
    # algorithmic bytes of the statistics kernel per launch: every int16 sample once, the three
    # doubles + two offsets of each read in, median + MAD out
    stats_bytes = 2 * n_samples + (24 + 16 + 16) * host.n_reads
    # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes, FETCH_SIZE doubled per the gfx950
    # note of MI355X_MICROARCH.md): measured offline on the headline workload and committed under profiles/
    traffic = None; trace_ms = None; traffic_source = None
    headline = args.reads == 50000 and args.read_len == 4000 and args.kind == "rna004" and args.k == 5
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            if headline:
                traffic = pmc["traffic_bytes_per_launch"]
                traffic_source = "profiles/" + name + " (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes of this command; a committed constant, NOT measured in this run)"
            break
        except Exception:
            pass
    # the committed rocprofv3 kernel trace of this command (profiles/): its average duration of the same kernel, for comparison
    trace_name = None
    for name in ("r06_kernel_stats.csv", "r05_kernel_stats.csv", "r04_kernel_stats.csv", "r03_kernel_stats.csv", "r02_kernel_stats.csv"):
        try:
            import csv
            for row in csv.DictReader(open(os.path.join(ROOT, "profiles", name))):
                if row["Name"].startswith("k_read_stats(") and headline:
                    trace_ms = float(row["AverageNs"]) * 1e-6
            trace_name = "profiles/" + name
            break
        except Exception:
            pass
    trace2_ms = None  # the same kernel in the committed trace of the DEFAULT command (two streams: it shares the chip there by design)
    try:
        import csv
        two = next(n for n in ("r06_kernel_stats_two_streams.csv", "r05_kernel_stats_two_streams.csv", "r04_kernel_stats_two_streams.csv", "r03_kernel_stats_two_streams.csv") if os.path.exists(os.path.join(ROOT, "profiles", n)))
        for row in csv.DictReader(open(os.path.join(ROOT, "profiles", two))):
            if row["Name"].startswith("k_read_stats(") and headline:
                trace2_ms = float(row["AverageNs"]) * 1e-6
    except Exception:
        pass
    roofline = {
        "bound": "hbm", "kernel": "k_read_stats", "achieved": stats_bytes / (stats_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": stats_bytes / (stats_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
        "bytes_per_launch": stats_bytes, "avg_launch_ms": stats_ms,
        "committed_trace_avg_launch_ms": trace_ms, "committed_trace_frac": (stats_bytes / (trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if trace_ms else None,
        "measured": "dispatch time stamps (HIP events attached to the launches) in a profile-mode pass of this run, kernels on one stream; "
                    f"the committed trace ({trace_name}) is `bench.py --one-stream` for the same reason: in the default two-stream "
                    "step the statistics share the chip with the ranking kernels by design",
        "committed_trace_two_streams_avg_launch_ms": trace2_ms,
    }
    if world == 1 and rank == 0 and not args.no_extras and not args.no_live_traffic:
        lt = live_traffic(args)
        if lt and "bytes" in lt:
            roofline["traffic_committed"] = {"bytes": traffic, "source": traffic_source}
            roofline["traffic"] = lt["bytes"]
            roofline["traffic_source"] = ("MEASURED beside this run: two child passes of this command under rocprofv3 (--pmc FETCH_SIZE, --pmc WRITE_SIZE, separate, one stream), "
                                          "averaged over k_read_stats' launches: FETCH_SIZE %.0f KB x 2 (gfx950) + WRITE_SIZE %.0f KB" % (lt["fetch_kb"], lt["write_kb"]))
            roofline["traffic_over_algorithmic"] = lt["bytes"] / stats_bytes
        elif lt:
            roofline["traffic_live_error"] = lt["error"]
    kernels_ms = {k: v[1] / n_prof for k, v in ks.items()}
    balg = b_alg(n_samples, host.n_reads, n_ops, n_bases, kept_samples, kept_events, len(kmers))
    whole_step = {"algorithmic_bytes": balg, "bytes_per_sample": balg / n_samples, "ms_per_step": ms_per_step,
                  "achieved_GBs": balg / (ms_per_step * 1e-3) / 1e9, "frac": balg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "kernels_sum_ms": sum(kernels_ms.values())}

    # the same job with statistics only for the reads that own a kept event (legal: acceptance is signal-independent in
    # the PAF path); reported next to `value`, never as `value`
    lazy_info = None
    if world == 1 and not dist_step and not args.lazy and not args.no_lazy_extra:
        lz, lms, _, _ = timed(dict(p, lazy_stats=True), args.steps, args.warmup)
        lres = lz.finish()
        touched = int(np.unique(lres.ev_read).size)
        lazy_info = {"value": n_samples / (lms * 1e-3), "ms_per_step": lms, "reads_with_statistics": touched,
                     "samples_touched": int(touched * args.read_len)}
        lz.close()

    # `value` is the library's default: two streams (PG_FLAG_OVERLAP) -- the statistics of batch i+1 next to the chain of batch i, on a
    # stream that may use three quarters of every XCD's CUs. Kernels then share the chip, so the per-kernel times and the roofline
    # object come from the profiled ONE-stream passes above; here the same job with every kernel on one stream (round 2's default).
    one_stream = None
    two_streams_on = world == 1 and not dist_step and not args.lazy and not args.one_stream
    if two_streams_on and not args.no_lazy_extra:
        ts, tms, _, tst = timed(dict(p, overlap=False), args.steps, args.warmup, extra_blocks=2)
        tblocks = sorted([tms] + tst["extra_blocks_ms"]); tms = tblocks[1]
        tres = ts.finish()
        same = (np.array_equal(tres.samples, res.samples) and np.array_equal(tres.ev_read, res.ev_read)
                and np.array_equal(tres.samp_off, res.samp_off))
        one_stream = {"value": n_samples / (tms * 1e-3), "ms_per_step": tms, "ms_per_step_blocks": {"min": tblocks[0], "median": tms, "max": tblocks[-1]},
                      "whole_step_frac": balg / (tms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "results_equal_two_streams": bool(same)}
        ts.close()

    out = {
        "metric": "signal samples/sec aggregated into k-mer buckets", "value": value, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup + args.pre_warm, "pre_warm_steps": args.pre_warm, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16 in / f64 arithmetic",
        "data": "synthetic",
        "config": {
            "workload": workload_label(args.kind, args.reads, args.read_len, args.k, args.sample_limit, world),
            "reads_per_gpu": args.reads, "samples_per_gpu": n_samples, "ss_ops_per_gpu": n_ops, "n_slots": len(kmers),
            "stats_mode": "lazy" if args.lazy else "every read (as the reference)", "parallelism": f"read-shard x{world}",
            "collective": ("all_gather of u64[n_slots] accepted counts per step over " + ("RCCL/xGMI" if backend == "nccl" else backend)) if dist_step else None,
            "statistics_placement": ("behind the issue of the all_gather, on the chain's stream (pg_stats, --defer)" if defer else "queued by pg_count on the second stream, as at N=1") if dist_step else None,
            "kept_events_rank0": kept_events, "kept_samples_rank0": kept_samples, "homopolymer_frac": hp_frac,
        },
        "ms_per_step_blocks": {"contract_block": ms_per_step, "min": all_blocks[0], "median": all_blocks[len(all_blocks) // 2], "max": all_blocks[-1], "steps_per_block": args.steps,
                               "note": "value / ms_per_step are the first block (exactly --steps timed steps behind exactly `warmup` untimed ones); two more blocks of the same length follow it: the settled figure"},
        "roofline": roofline,
        "whole_step": whole_step,
        "whole_step_frac": whole_step["frac"],
        "useful": useful,
        "kernels_ms_per_step": kernels_ms,
        "host_enqueue_ms_per_step": enqueue_ms,
        "lazy_statistics_mode": lazy_info,
        "streams": ("two: statistics of the next batch on a stream of their own (3/4 of every XCD's CUs)" if two_streams_on else "one"),
        "one_stream_mode": one_stream,
        "kmer_model_once_per_job": model_info,
        "gen_seconds": gen_s,
    }
    eng.close()

    extras = not args.no_extras and not args.force_dist
    if extras and world > 1:
        # BASELINE configs[2]: the same shards as ONE job at sample_limit 5000 (at N = 8: 400 000 reads sharded 8 ways)
        e2, ms2, _, st2 = timed(dict(p, sample_limit=5000), max(5, args.steps // 4), 3)
        u2 = useful_of(e2, st2, 5000)
        out["config2_mode"] = {"workload": workload_label(args.kind, args.reads, args.read_len, args.k, 5000, world), "ms_per_step": ms2,
                               "value": world * n_samples / (ms2 * 1e-3), "unit": "samples/s", "useful": u2}
        e2.close()
        out["job_layer"] = job_layer(args, world, rank, dist)
    if extras and rank == 0 and world == 1:
        t0 = time.time()
        out["all_kept_mode"] = mode_run(shard, host, kmers, dict(p, sample_limit=5000),
                                        "BASELINE configs[2]'s limit on the headline reads: sample_limit 5000, every accepted event is kept")
        out["hbm_not_mall"] = hbm_not_mall(shard, kmers, p, dev)
        out["pcie_inclusive"] = pcie_inclusive(host, kmers, p)
        out["early_stop_mode"] = early_stop_mode(host, kmers, p, dev)
        del shard
        torch.cuda.empty_cache()
        out["config3_mode"] = config3_mode(dev)
        out["ragged_mode"] = ragged_mode(dev)
        out["end_to_end"] = end_to_end(host, args)
        out["extras_seconds"] = time.time() - t0
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(host, kmers, p, args.cpu_seconds)
    if rank == 0:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)  # the ONE line of the contract on the real stdout
        print(json.dumps(out), flush=True)
    if dist_step:
        dist.destroy_process_group()


def mode_run(shard, host, kmers, q, what, steps=10):
    """One more workload on a device-resident batch: ms per step (plain pg_submit: the library's default, two streams), the whole step
    against the HBM roofline, per-kernel times from a separate profiled (one-stream) pass."""
    import torch
    from poregen_amd.engine import GmoveEngine, GmoveParams
    e = GmoveEngine(GmoveParams(kmers=kmers, **q))
    MODE_WARMUP = 30
    for _ in range(MODE_WARMUP):
        e.reset(); e.submit(shard)
    e.sync()
    blocks = []  # three timed blocks: boxes (and minutes) differ by several per cent, one number says little
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            e.reset(); e.submit(shard)
        e.sync(); torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / steps * 1e3)
    ms = sorted(blocks)[1]
    v = e.device_view()
    ke, ksm = int(v.n_events), int(v.n_samples)
    e.close()
    pe = GmoveEngine(GmoveParams(kmers=kmers, profile=True, **q))
    for _ in range(2):
        pe.reset(); pe.submit(shard)
    pe.sync(); pe.kernel_stats_reset()
    for _ in range(5):
        pe.reset(); pe.submit(shard)
    pe.sync()
    raw = pe.kernel_stats()
    counters = {k: v2[0] for k, v2 in raw.items() if v2[1] == 0.0 and k in ("long_reads_split", "long_helpers_short_batches", "stats_cancelled_on_device")}
    ks = {k: v2[1] / 5 for k, v2 in raw.items() if k not in counters}
    pe.close()
    n_ops, n_bases = int(host.op_off[-1]), int(host.seq_off[-1])
    balg = b_alg(host.n_samples, host.n_reads, n_ops, n_bases, ksm, ke, len(kmers))
    return {"workload": what, "sample_limit": q["sample_limit"], "ms_per_step": ms, "ms_per_step_blocks": {"min": min(blocks), "median": ms, "max": max(blocks), "steps_per_block": steps, "warmup_steps": MODE_WARMUP},
            "value": host.n_samples / (ms * 1e-3), "unit": "samples/s",
            "kept_events": ke, "kept_samples": ksm, "algorithmic_bytes": balg, "whole_step_frac": balg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "whole_step_frac_blocks": {"min": balg / (max(blocks) * 1e-3) / 1e9 / HBM_PEAK_GBS, "max": balg / (min(blocks) * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "kernels_ms_per_step": ks, "counters": counters}


def config3_mode(dev):
    """BASELINE configs[3]: synthetic DNA (R10 dwell), 50 000 reads x 4 000 samples, k = 9: 262 144 k-mers, sample_limit 1000."""
    from poregen_amd import synth
    from poregen_amd.engine import generate_kmers
    import numpy as np
    from poregen_amd.engine import GmoveEngine, GmoveParams
    host = synth.make_batch_fast(50000, read_len=4000, kind="dna_r10", seed=20251003 + 3, homopolymer_frac=0.1)
    shard = host.to_device(dev)
    kmers = generate_kmers(9, rna=False)
    q = dict(kmer_size=9, rna=False, scaling=1, sample_limit=1000, device=dev.index or 0)
    r = mode_run(shard, host, kmers, q, workload_label("dna_r10", 50000, 4000, 9, 1000, 1) + ", 10 % homopolymer-rich reads (SURVEY 8d cfg 3)")
    r["n_slots"] = len(kmers)
    # the skew the homopolymer reads add: slots whose accepted events exceed the cap, and the largest partition region's share
    e = GmoveEngine(GmoveParams(kmers=kmers, **q))
    acc = e.count(shard).astype(np.int64)
    e.close()
    reg = acc.reshape(512, -1).sum(axis=1)
    r["skew"] = {"accepted_events": int(acc.sum()), "slots_at_cap": int((acc >= 1000).sum()), "events_cut_by_cap": int(np.maximum(acc - 1000, 0).sum()),
                 "largest_slot": int(acc.max()), "largest_region_share_of_512": float(reg.max() / max(1, reg.sum())), "mean_region_share": 1.0 / 512}
    return r


def ragged_mode(dev):
    """The regime no fixed-length workload shows (SURVEY section 5 "long-context", section 7 "hard parts"): DNA reads of lognormal length
    2 000 .. 200 000 samples plus 0.1 % reads of 10^6 samples, 2 x 10^8 samples in all -- the headline's total -- at k = 9 (configs[3]'s
    flags) and k = 5. A read above 32 768 samples has its statistics cut into slices for several waves (pg_internal.h: PgLongState);
    `one_wave_per_read` is the same step with that switched off (PGMOVE_NO_LONG_SPLIT=1): one wave then streams a 2 MB read alone."""
    import numpy as np
    from poregen_amd import synth
    from poregen_amd.engine import generate_kmers
    L = synth.ragged_lengths(total_samples=200_000_000, seed=20251005)
    host = synth.make_ragged_fast(L, kind="dna_r10", seed=20251005 + 7)
    shard = host.to_device(dev)
    out = {"reads": int(host.n_reads), "samples": int(host.n_samples), "ss_ops": int(host.op_off[-1]),
           "read_length": {"min": int(L.min()), "median": int(np.median(L)), "mean": float(L.mean()), "max": int(L.max()), "reads_of_1e6": int((L == 1_000_000).sum()),
                           "reads_above_split_threshold_32768": int((L > 32768).sum()), "share_of_samples_in_them": float(L[L > 32768].sum() / L.sum())}}
    stats_bytes = 2 * host.n_samples + 56 * host.n_reads
    for k, limit in ((9, 1000), (5, 100)):
        kmers = generate_kmers(k, rna=False)
        q = dict(kmer_size=k, rna=False, scaling=1, sample_limit=limit, device=dev.index or 0)
        r = mode_run(shard, host, kmers, q, f"ragged DNA reads, k={k}, scaling med-MAD, sample_limit={limit}", steps=5)
        ms = r["kernels_ms_per_step"].get("k_read_stats")
        r["k_read_stats"] = {"avg_launch_ms": ms, "bytes_per_launch": stats_bytes, "GB/s": stats_bytes / (ms * 1e-3) / 1e9, "frac": stats_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        os.environ["PGMOVE_NO_LONG_SPLIT"] = "1"
        try:
            r1 = mode_run(shard, host, kmers, q, "the same, one wave per read", steps=3)
        finally:
            del os.environ["PGMOVE_NO_LONG_SPLIT"]
        ms1 = r1["kernels_ms_per_step"].get("k_read_stats")
        r["one_wave_per_read"] = {"ms_per_step": r1["ms_per_step"], "whole_step_frac": r1["whole_step_frac"], "k_read_stats_ms": ms1,
                                  "k_read_stats_frac": stats_bytes / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        out[f"k{k}"] = r
    return out


def job_layer(args, world, rank, dist):
    """The step through pg_job_* : ONE process, `world` devices, a host thread and a context per device, the exchange over
    ncclCommInitAll, the shards' kept samples concatenated on the first device -- what `poregen gmove --devices 0,1,...` runs. It runs in
    a CHILD process of rank 0 (this file with --job-layer-child) under a time limit, while the ranks idle at the barrier: the path has
    never seen more than one GPU before the driver's first multi-GPU run, and a fault or a hang in it must not take the headline line
    with it. The job layer stages its batch from host memory (rank 0's shard, cut `world` ways): a PCIe-inclusive figure."""
    import subprocess
    import torch
    info = None
    # The other ranks wait on the HOST (a gloo group: no RCCL kernel spins on their GPUs while the child times its steps on the same
    # devices), with their GPU work drained first.
    torch.cuda.synchronize()
    try:
        cpu_group = dist.new_group(backend="gloo")
    except Exception:
        cpu_group = None
    if rank == 0:
        cmd = [sys.executable, os.path.abspath(__file__), "--job-layer-child", str(world), "--reads", str(args.reads), "--read-len", str(args.read_len),
               "--kind", args.kind, "--k", str(args.k), "--sample-limit", str(args.sample_limit)]
        if args.lib:
            cmd += ["--lib", args.lib]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            info = json.loads(line[-1]) if line else {"error": "child exited %d: %s" % (r.returncode, r.stderr[-300:])}
        except subprocess.TimeoutExpired:
            info = {"error": "the pg_job child did not finish within 240 s (killed)"}
        except Exception as ex:  # the headline must not die with the extra
            info = {"error": repr(ex)[:300]}
        if isinstance(info, dict):
            info["other_ranks_during_the_child"] = "idle on a host-side (gloo) barrier, GPU work drained" if cpu_group is not None else "inside an RCCL barrier"
    if cpu_group is not None:
        dist.barrier(group=cpu_group)
    else:
        dist.barrier()
    return info


def job_layer_child(args):
    """see job_layer: one process, args.job_layer_child devices; prints ONE JSON object"""
    world = args.job_layer_child
    try:
        if args.lib:
            from poregen_amd import _abi as abi0
            abi0.LIB_PATH = os.path.abspath(args.lib)
        from poregen_amd import _abi, synth
        from poregen_amd.engine import GmoveJob, GmoveParams, generate_kmers
        rna = args.kind == "rna004"
        p = dict(kmer_size=args.k, rna=rna, scaling=1, sample_limit=args.sample_limit)
        if rna:
            p.update(min_dur=20, max_dur=40)
        kmers = generate_kmers(args.k, rna=rna)
        host = synth.make_batch_fast(args.reads, read_len=args.read_len, kind=args.kind, seed=20251003 + 1)
        job = GmoveJob(GmoveParams(kmers=kmers, **p), list(range(world)), _abi.PG_JOB_EXCHANGE_AUTO)
        job.submit(host); job.sync()      # contexts, communicators, first buffers
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            job.submit(host)
        job.sync()
        ms = (time.perf_counter() - t0) / n * 1e3
        t1 = time.perf_counter()
        r = job.finish_deferred()         # small arrays merged on the host, samples concatenated on the first device (peer copies) and fetched
        fin_ms = (time.perf_counter() - t1) * 1e3
        # the same step with every rank's shard RESIDENT on its device (pg_job_submit_shards): nothing crosses PCIe inside the step, the
        # C++ host only queues -- the device-resident N-GPU step of the job layer
        import torch
        cuts = [host.n_reads * g // world for g in range(world + 1)]
        shards = [host.slice_reads(cuts[g], cuts[g + 1]).to_device(torch.device("cuda", g)) for g in range(world)]
        torch.cuda.synchronize()
        job2 = GmoveJob(GmoveParams(kmers=kmers, **p), list(range(world)), _abi.PG_JOB_EXCHANGE_AUTO)
        for _ in range(3):
            job2.reset(); job2.submit_shards(shards)
        job2.sync()
        steps2 = 20
        t2 = time.perf_counter()
        for _ in range(steps2):
            job2.reset(); job2.submit_shards(shards)   # a step = a new job over the resident shards (pg_job_reset synchronises the ranks: a host round trip per step)
        job2.sync()
        res_ms = (time.perf_counter() - t2) / steps2 * 1e3
        resident = {"ms_per_step": res_ms, "value": int(host.n_samples) / (res_ms * 1e-3), "unit": "samples/s",
                    "note": "pg_job_reset + pg_job_submit_shards on device-resident shards, host-synchronous per step (the job layer settles a batch before the next)"}
        job2.close()
        info = {"devices": world, "exchange": "rccl" if job.uses_rccl else "host", "rccl_ranks_seen": world if job.uses_rccl else 0,
                "ms_per_step_pcie_inclusive": ms, "reads_per_step": host.n_reads, "samples_per_step": int(host.n_samples),
                "kept_events": int(r.counts.sum()), "kept_samples": int(r.samples.size), "finish_deferred_and_fetch_ms": fin_ms,
                "all_kmers_complete": job.all_slots_full(), "process": "child of rank 0", "device_resident_shards": resident}
        job.close()
    except Exception as ex:
        info = {"error": repr(ex)[:300]}
    sys.stdout.flush()
    print(json.dumps(info), flush=True)
    return 0


def hbm_not_mall(shard, kmers, p, dev, copies=8):
    """k_read_stats over ONE batch of copies x 50 000 reads (3.2 GB of signal, 12x the 256 MiB Infinity Cache): the same rate as on
    the replayed 400 MB buffer means the headline rate is an HBM rate, not a MALL rate. The batch is the shard replicated on the
    device (offsets shifted), so its content repeats but no address does."""
    import torch
    from poregen_amd.engine import Batch, GmoveEngine, GmoveParams

    def rep_off(o):
        body = o[:-1]
        return torch.cat([body + i * o[-1] for i in range(copies)] + [(copies * o[-1]).reshape(1)])
    big = Batch(n_reads=shard.n_reads * copies, on_device=True, n_ops=shard.n_ops * copies, all_matches=shard.all_matches,
                sig=torch.cat([shard.sig[:-8].repeat(copies), torch.zeros(8, dtype=shard.sig.dtype, device=dev)]), sig_off=rep_off(shard.sig_off),
                digitisation=shard.digitisation.repeat(copies), offset=shard.offset.repeat(copies), range=shard.range.repeat(copies),
                query_start=shard.query_start.repeat(copies), target_start=shard.target_start.repeat(copies), target_end=shard.target_end.repeat(copies),
                seq=shard.seq.repeat(copies), seq_off=rep_off(shard.seq_off), op_n=shard.op_n.repeat(copies), op_t=shard.op_t.repeat(copies),
                op_off=rep_off(shard.op_off))
    torch.cuda.synchronize()
    e = GmoveEngine(GmoveParams(kmers=kmers, profile=True, **p))
    e.reset(); e.submit(big); e.sync(); e.kernel_stats_reset()
    n = 4
    for _ in range(n):
        e.reset(); e.submit(big)
    e.sync()
    ks = e.kernel_stats()
    e.close()
    ms = ks["k_read_stats"][1] / ks["k_read_stats"][0]
    n_samples = int(big.sig.numel()) - 8
    nbytes = 2 * n_samples + 56 * big.n_reads
    step_ms = sum(v[1] for v in ks.values()) / n
    del big
    return {"reads": shard.n_reads * copies, "signal_bytes": 2 * n_samples, "k_read_stats_ms": ms, "GB/s": nbytes / (ms * 1e-3) / 1e9,
            "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernels_sum_ms_per_step": step_ms}


def early_stop_mode(host, kmers, p, dev, jobs=20):
    """The same reads as the reference works through them: in PAF order until every k-mer is complete, then no further (gmove.cpp:733-735).
    The headline step streams the whole 50 000-read batch, of which the reference would have read the first few thousand (`useful`); here the
    device-resident reads are submitted in the CLI's ramp (2 048 reads, 4 096, 8 192, ...; PG_FLAG_STOP_WHEN_FULL) and the job ends with the
    batch that completes the list. value = samples of the reads the JOB submitted / its time: what a caller who feeds a device sees per job."""
    import torch
    from poregen_amd.engine import GmoveEngine, GmoveParams
    parts, lo, size = [], 0, 2048
    while lo < host.n_reads:
        hi = min(host.n_reads, lo + size)
        parts.append(host.slice_reads(lo, hi).to_device(dev)); lo = hi; size *= 2
    e = GmoveEngine(GmoveParams(kmers=kmers, stop_when_full=True, **p))

    def job():
        e.reset()
        n = 0
        for b in parts:
            e.submit(b); n += b.n_reads
            if e.all_slots_full():   # (waits for the batch: the decision the CLI takes from pg_poll one batch late)
                break
        return n
    for _ in range(3):
        n_read = job()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(jobs):
        n_read = job()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / jobs * 1e3
    complete = e.all_slots_full()
    e.close()
    samples = int(host.sig_off[n_read])
    return {"batches": [int(b.n_reads) for b in parts], "reads_submitted": int(n_read), "samples_submitted": samples, "all_kmers_complete": bool(complete),
            "ms_per_job": ms, "value": samples / (ms * 1e-3), "unit": "samples/s",
            "note": "one job = reset, ramped device-resident batches until every k-mer is complete (a host wait per batch); compare `useful` of the headline step"}


def pcie_inclusive(host, kmers, p, steps=4):
    """The step fed from HOST memory: pg_submit with host pointers stages the batch over PCIe first (never `value`)."""
    from poregen_amd.engine import GmoveEngine, GmoveParams
    e = GmoveEngine(GmoveParams(kmers=kmers, **p))
    e.reset(); e.submit(host); e.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        e.reset(); e.submit(host)
    e.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    e.close()
    nbytes = sum(getattr(host, f).nbytes for f in ("sig", "sig_off", "digitisation", "offset", "range", "query_start", "target_start", "target_end", "seq", "seq_off", "op_n", "op_t", "op_off"))
    return {"ms_per_step": ms, "value": host.n_samples / (ms * 1e-3), "unit": "samples/s", "staged_bytes": nbytes, "staging_GB/s": nbytes / (ms * 1e-3) / 1e9,
            "source": "pageable host memory (numpy arrays), hipMemcpyAsync per array"}


def end_to_end(host, args):
    """SURVEY 8(d) metric (i): `bin/poregen gmove` as a child process on the workload written as BLOW5 + PAF + FASTQ, input files in the
    page cache, wall clock from process start until the dump files are closed and the process has exited. `value` divides the samples
    the run actually READ (like the reference it stops reading once every k-mer is complete) by the wall time; the figure over all
    samples of the input files stands next to it as value_over_all_input."""
    import re
    import shutil
    import tempfile
    from poregen_amd import synth
    d = tempfile.mkdtemp(prefix="pg_e2e_", dir="/tmp")
    try:
        synth.write_blow5(host, d + "/r.blow5"); synth.write_paf_fastq(host, d + "/r")
        for f in ("r.blow5", "r.paf", "r.fastq"):
            with open(os.path.join(d, f), "rb") as fh:      # page cache
                while fh.read(1 << 24):
                    pass
        threads = min(16, os.cpu_count() or 1)
        out = {"host_threads": threads, "files": "uncompressed BLOW5 + PAF(ss) + FASTQ in the page cache", "runs": {}}
        for lim in (args.sample_limit, 5000):
            best = None
            for rep in range(3):
                o = f"{d}/out_{lim}_{rep}"
                cmd = [os.path.join(ROOT, "bin", "poregen"), "gmove", "-k", str(args.k), "--scaling", "1", "--file_limit", str(4 ** args.k), "--sample_limit", str(lim),
                       d + "/r.blow5", d + "/r.paf", "--fastq", d + "/r.fastq", o] + (["--rna", "--min_dur", "20", "--max_dur", "40"] if args.kind == "rna004" else [])
                t0 = time.perf_counter()
                r = subprocess.run(cmd, capture_output=True, text=True)
                wall = time.perf_counter() - t0
                if r.returncode != 0:
                    return {"error": r.stderr[-400:]}
                stages = [ln[len("[gmove] "):] for ln in r.stderr.splitlines() if ln.startswith("[gmove] ")]
                m = re.search(r"\[gmove\] (\d+) reads, (\d+) samples", r.stderr)
                read_samples = int(m.group(2)) if m else int(host.n_samples)
                if best is None or wall < best["wall_s"]:
                    best = {"wall_s": wall, "reads_read": int(m.group(1)) if m else host.n_reads, "samples_read": read_samples,
                            "value": read_samples / wall, "value_over_all_input": host.n_samples / wall, "unit": "samples/s", "stages": stages}
                shutil.rmtree(o, ignore_errors=True)
            out["runs"][f"sample_limit_{lim}"] = best
        first = out["runs"][f"sample_limit_{args.sample_limit}"]
        out.update(value=first["value"], value_over_all_input=first["value_over_all_input"], wall_s=first["wall_s"], unit="samples/s", stages=first["stages"])
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def live_traffic(args, timeout_s=90):
    """HBM bytes per launch of the dominant kernel, MEASURED beside this run: two child passes of this very command under rocprofv3
    (--pmc FETCH_SIZE and --pmc WRITE_SIZE, separately -- MI355X_MICROARCH.md, HBM / rocprofv3 section), on one stream like the per-kernel
    times, the counters averaged over k_read_stats' launches, FETCH_SIZE doubled (the gfx950 correction for wide streaming reads).
    Children, not exec: this process has initialised the GPU. None if rocprofv3 is absent or a pass fails (the committed figure stays)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None
    d = tempfile.mkdtemp(prefix="pg_pmc_", dir="/tmp")
    try:
        kb = {}
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", name, "--output-format", "csv", "-d", os.path.join(d, name), "--", sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1",
                   "--one-stream", "--no-cpu-baseline", "--no-lazy-extra", "--no-extras", "--reads", str(args.reads), "--read-len", str(args.read_len), "--kind", args.kind,
                   "--k", str(args.k), "--sample-limit", str(args.sample_limit)] + (["--lib", args.lib] if args.lib else [])
            env = dict(os.environ, TMPDIR="/tmp")
            for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(v, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return {"error": f"rocprofv3 --pmc {name}: rc {r.returncode}: " + r.stderr[-200:]}
            vals = []
            for f in glob.glob(os.path.join(d, name, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == name and row.get("Kernel_Name", "").split("(")[0].strip() == "k_read_stats":
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return {"error": f"no {name} rows for k_read_stats"}
            kb[name] = sum(vals) / len(vals)
        return {"fetch_kb": kb["FETCH_SIZE"], "write_kb": kb["WRITE_SIZE"], "bytes": (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0}
    except Exception as e:  # (a profiler that is missing a library, a timeout: the committed figure stays)
        return {"error": repr(e)[:200]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(host, kmers, p, budget_s):
    """The CPU oracle (oracle/, a single-threaded port of the reference algorithm; the reference itself cannot be
    built in this image) on successive 4000-read slices of the same workload until ~budget_s of CPU work."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    from poregen_amd import synth
    spent = 0.0; samples = 0; reads = 0; runs = 0
    lo = 0
    while spent < budget_s and lo < host.n_reads:
        hi = min(host.n_reads, lo + 4000)
        o = orc.Oracle(kmers, kmer_size=p["kmer_size"], scaling=1, sample_limit=p["sample_limit"], flag_rna=int(p["rna"]),
                       min_dur=p.get("min_dur", 5), max_dur=p.get("max_dur", 70))
        prepared = []
        for r in range(lo, hi):
            ts, te = int(host.target_start[r]), int(host.target_end[r])
            prepared.append((host.sig[int(host.sig_off[r]):int(host.sig_off[r + 1])], float(host.digitisation[r]), float(host.offset[r]),
                             float(host.range[r]), int(host.query_start[r]), ts, te, synth.seq_string(host, r), synth.ss_string(host, r)))
        t0 = time.perf_counter()
        for a in prepared:
            rc = o.paf_read(*a)
            if rc == orc.ORC_STOPPED:
                break
        spent += time.perf_counter() - t0
        samples += o.total_samples(); reads += int(o.L.orc_reads_seen(o.h))
        runs += 1; lo = hi
    return {"value": samples / spent if spent > 0 else 0.0, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"{runs} oracle runs over successive 4000-read slices of the workload (each run stops, like the reference, once all "
                      f"k-mers are complete); {reads} reads / {samples} samples actually processed in {spent:.2f} s"}


if __name__ == "__main__":
    main()
