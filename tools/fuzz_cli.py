"""Randomised whole-directory comparison of `bin/poregen gmove` with the CPU oracle's CLI on synthetic files: PAF, move-
table and SAM front-ends, random options. Not part of the test suite. usage: python3 tools/fuzz_cli.py [n_cases] [seed]"""
import filecmp, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import orc
from poregen_amd import synth

BIN = os.path.join(ROOT, "bin", "poregen")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = skipped = 0
for case in range(n_cases):
    d = tempfile.mkdtemp(prefix="pgfz_")
    try:
        use_bam = blow5 = False
        front = str(rng.choice(["paf", "paf", "table", "sam", "sam"]))
        k = int(rng.choice([3, 5, 6]))
        n_reads = int(rng.choice([1, 5, 40, 120]))
        pre = os.path.join(d, "syn")
        opts = ["-k", str(k), "--sample_limit", str(int(rng.choice([1, 4, 30, 100]))), "--file_limit", str(4 ** k)]
        if rng.random() < 0.6: opts += ["--scaling", "1"]
        if rng.random() < 0.3: opts += ["-d"]
        if rng.random() < 0.3: opts += ["--min_dur", str(int(rng.choice([1, 10, 20]))), "--max_dur", str(int(rng.choice([25, 40, 100])))]
        if rng.random() < 0.2: opts += ["--pa_min", str(float(rng.choice([-20.0, 60.0, 100.0]))), "--pa_max", str(float(rng.choice([120.0, 150.0, 400.0])))]
        if rng.random() < 0.3:
            a = int(rng.integers(1, 4 ** k + 1)); b2 = int(rng.integers(a, 4 ** k + 1)); opts += ["--index_start", str(a), "--index_end", str(b2)]
        if front == "paf":
            rna = bool(rng.integers(0, 2))
            b = synth.make_batch(n_reads, read_len=int(rng.choice([600, 4000])), kind="rna004" if rna else "dna_r10", seed=int(rng.integers(1 << 30)),
                                 indel_rate=float(rng.choice([0.0, 0.03])))
            synth.write_files(b, pre)
            args = [pre + ".slow5", pre + ".paf", "--fastq", pre + ".fastq"] + opts + (["--rna"] if rna else [])
            args += ["--kmer_pick_margin", str(int(rng.integers(0, 4)))]
            blow5 = rng.random() < 0.5  # the product reads a BLOW5 of the same reads (zlib + svb-zd half of the time), the oracle the ASCII SLOW5
            if blow5:
                comp = [False, True, "zstd"][int(rng.integers(0, 3))]  # none / zlib + svb-zd / zstd + svb-zd (round 6)
                if comp == "zstd" and synth.zstd_compress(b"x") is None: comp = True
                synth.write_blow5(b, pre + ".blow5", compress=comp)
            if rng.random() < 0.2: args += ["--margin", str(int(rng.integers(1, 4)))]
        else:
            b = synth.make_batch(n_reads, read_len=int(rng.choice([600, 4000])), kind="dna_r10", seed=int(rng.integers(1 << 30)))
            trim = int(rng.choice([0, 0, 23]))
            synth.write_table_files(b, pre, trim=trim)
            args = [pre + ".slow5", pre + (".table" if front == "table" else ".sam")] + opts
            use_bam = front == "sam" and rng.random() < 0.6  # the product reads the BAM, the oracle CLI the SAM text of the same records
            if use_bam:
                synth.write_bam(b, pre + ".bam", trim=trim, block_bytes=int(rng.choice([300, 5000, 65000])))
            if front == "table":
                args += ["-m", str(int(rng.integers(0, min(k, 3)))), "-s", str(int(rng.integers(0, 3)))]
            if rng.random() < 0.2: args += ["--margin", str(int(rng.integers(1, 4)))]
        o = subprocess.run([orc.CLI] + args + [os.path.join(d, "cpu")], capture_output=True, text=True)
        if o.returncode == 70:  # the oracle flags an input on which the reference has undefined behaviour
            skipped += 1
            continue
        gargs = [a[:-4] + ".bam" if (front == "sam" and use_bam and a.endswith(".sam")) else a for a in args] if front != "paf" else \
                [a[:-6] + ".blow5" if (blow5 and a.endswith(".slow5")) else a for a in args]
        extra = ["--batch_reads", str(int(rng.choice([1, 7, 64, 20000])))] if rng.random() < 0.7 else []  # (no --batch_reads: the ramp of a whole-list job)
        if rng.random() < 0.25:  # the job layer: several shards on the one GPU of the box (host exchange), or one rank over RCCL
            extra += ["--devices", str(rng.choice(["0,0", "0,0,0,0", "0"]))]
        env = dict(os.environ)
        if not extra and rng.random() < 0.7: env["POREGEN_BATCH_RAMP"] = str(int(rng.choice([1, 3, 16])))  # tiny first batches: 1, 2, 4, ... reads
        if rng.random() < 0.3: env["PGMOVE_HOLD_MIN_BYTES"] = "1"  # small batches' samples stay on the device too (device merge, device text)
        g = subprocess.run([BIN, "gmove"] + gargs + [os.path.join(d, "gpu")] + extra, capture_output=True, text=True, env=env)
        ok = (o.returncode == 0) == (g.returncode == 0)
        if ok and o.returncode == 0:
            ok = open(os.path.join(d, "gpu", "freq.txt")).read() == open(os.path.join(d, "cpu", "freq.txt")).read()
            names = sorted(os.listdir(os.path.join(d, "cpu", "dump")))
            ok = ok and names == sorted(os.listdir(os.path.join(d, "gpu", "dump")))
            if ok:
                _, mism, errs = filecmp.cmpfiles(os.path.join(d, "gpu", "dump"), os.path.join(d, "cpu", "dump"), names, shallow=False)
                ok = not mism and not errs
        if not ok:
            bad += 1
            keep = os.path.join(ROOT, "gpurun_out", f"fuzz_cli_case{case}")
            os.makedirs(os.path.dirname(keep), exist_ok=True); shutil.copytree(d, keep, dirs_exist_ok=True)
            print("CASE", case, "DIFFERS:", front, " ".join(args + extra), "| oracle rc", o.returncode, "gpu rc", g.returncode, g.stderr[-300:].replace("\n", " | "), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    if case % 20 == 19:
        print("case", case + 1, "ok so far" if not bad else f"{bad} differences", flush=True)
print("cli fuzz done:", n_cases, "cases,", skipped, "outside the reference's defined behaviour (skipped),", bad, "differences")
sys.exit(1 if bad else 0)
