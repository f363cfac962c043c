"""End-to-end `poregen gmove` on BASELINE config 1 as files (BLOW5, uncompressed or zlib + svb-zd, + PAF + FASTQ): wall time of the whole
process (host parsing + PCIe + GPU + %.8f formatting + writing 1024 files), next to the CPU oracle CLI on a prefix."""
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poregen_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
compress = len(sys.argv) > 2 and sys.argv[2] == "zlib"   # "zlib": zlib records + svb-zd signals (slow5tools' default)
d = "/tmp/pg_e2e"; shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
t0 = time.time(); b = synth.make_batch_fast(n, kind="rna004", seed=20251004); print("generated", n, "reads in %.1f s" % (time.time() - t0))
t0 = time.time(); synth.write_blow5(b, d + "/r.blow5", compress=compress); synth.write_paf_fastq(b, d + "/r"); print("wrote files in %.1f s" % (time.time() - t0))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lim in (100, 5000):
    for rep in range(2):
        out = f"{d}/out_{lim}_{rep}"
        t0 = time.time()
        r = subprocess.run([root + "/bin/poregen", "gmove", "-k", "5", "--rna", "--scaling", "1", "--min_dur", "20", "--max_dur", "40", "--file_limit", "1024",
                            "--sample_limit", str(lim), d + "/r.blow5", d + "/r.paf", "--fastq", d + "/r.fastq", out, "--batch_reads", "50000"], capture_output=True, text=True)
        dt = time.time() - t0
        print(f"sample_limit {lim} run {rep}: exit {r.returncode}, wall {dt:.2f} s -> {b.n_samples / dt / 1e6:.1f} M samples/s end to end;", "; ".join(l for l in r.stderr.splitlines() if l.startswith("[gmove]"))[:900])
