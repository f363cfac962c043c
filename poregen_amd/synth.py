"""Deterministic synthetic SLOW5+PAF+FASTQ workloads of the shapes BASELINE.json names.

The generator produces the parsed, structure-of-arrays form directly (what the host parsers would hand to
libpgmove) and can also write the three files for CLI runs. Shapes:
  rna004  : reads x 4000 samples, ~31 samples/base, RNA-oriented PAF (target_start > target_end)
  dna_r10 : reads x 4000 samples, stride-5 dwells with mean ~12.5, DNA-oriented PAF
Optional indels put `nD` / `nI` ops into the ss strings (config 4 of BASELINE.json).
"""
import numpy as np

from .engine import Batch

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def _splitmix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def make_batch(n_reads=1000, read_len=4000, kind="rna004", seed=20251003, indel_rate=0.0, spike_rate=0.005,
               chunk_reads=4000, homopolymer_frac=0.0):
    """Returns (Batch, meta) where meta holds per-read base strings (for FASTQ / the oracle)."""
    rng = np.random.default_rng(seed)
    rna = kind == "rna004"
    mean_extra = 5.2 if rna else 1.5          # dwell = 5*(1+Geom) -> mean ~31 (RNA004) / ~12.5 (DNA stride 5)
    p_geom = 1.0 / (1.0 + mean_extra)
    cand = int(read_len / (5 * (1 + mean_extra)) * 3) + 64

    sig_parts, op_n_parts, op_t_parts, seq_parts = [], [], [], []
    sig_off = [0]; seq_off = [0]; op_off = [0]
    dig = np.full(n_reads, 2048.0)
    offset = rng.integers(-260, -229, n_reads).astype(np.float64)
    rng_range = rng.uniform(280.0, 285.0, n_reads)
    qstart = np.zeros(n_reads, np.int32); tstart = np.zeros(n_reads, np.int32); tend = np.zeros(n_reads, np.int32)

    for c0 in range(0, n_reads, chunk_reads):
        c1 = min(n_reads, c0 + chunk_reads)
        nr = c1 - c0
        dw = 5 * (1 + (rng.geometric(p_geom, size=(nr, cand)) - 1))
        dw = np.clip(dw, 5, 200).astype(np.int64)
        cs = np.cumsum(dw, axis=1)
        nb = (cs < read_len).sum(axis=1) + 1                      # bases per read; the last dwell absorbs the remainder
        nb = np.minimum(nb, cand)
        for i in range(nr):
            r = c0 + i
            n = int(nb[i])
            d = dw[i, :n].copy()
            d[n - 1] = read_len - (cs[i, n - 2] if n > 1 else 0)
            if d[n - 1] <= 0:                                      # cannot happen with cand large enough; keep dwells >= 1
                d[n - 1] = 1
            bases = rng.integers(0, 4, n)
            if homopolymer_frac > 0 and rng.random() < homopolymer_frac:
                run = rng.integers(0, 4); a = rng.integers(0, max(1, n - 40)); bases[a:a + 40] = run
            # 5-mer context level: 70 + 60 * hash / 2^64 pA
            code = np.zeros(n, np.uint64)
            for t in range(5):
                sh = np.roll(bases, t).astype(np.uint64); sh[:t] = 0
                code |= sh << np.uint64(2 * t)
            level = 70.0 + 60.0 * (_splitmix(code).astype(np.float64) / 2.0 ** 64)
            # ops: matches, optionally with insertions (skip samples) / deletions (skip bases)
            if indel_rate > 0:
                ops_n, ops_t, seq_b, lv, dd = [], [], [], [], []
                for j in range(n):
                    u = rng.random()
                    if u < indel_rate and j > 8 and j < n - 8:     # deletion: bases present in the sequence, no samples
                        nd = int(rng.integers(1, 4)); ops_n.append(nd); ops_t.append(2)
                        seq_b.extend(rng.integers(0, 4, nd).tolist())
                    elif u < 2 * indel_rate and j > 8 and j < n - 8:  # insertion: samples that belong to no base
                        ni = int(rng.integers(5, 41)); ops_n.append(ni); ops_t.append(1)
                        lv.append(100.0); dd.append(ni)
                    ops_n.append(int(d[j])); ops_t.append(0); seq_b.append(int(bases[j])); lv.append(level[j]); dd.append(int(d[j]))
                lv = np.asarray(lv); dd = np.asarray(dd, np.int64)
                seq_codes = np.asarray(seq_b, np.int64)
                ops_n = np.asarray(ops_n, np.uint32); ops_t = np.asarray(ops_t, np.uint8)
            else:
                lv, dd, seq_codes = level, d, bases
                ops_n = d.astype(np.uint32); ops_t = np.zeros(n, np.uint8)
            L = int(dd.sum())
            pa = np.repeat(lv, dd) + rng.normal(0.0, 3.0, L)
            sp = rng.random(L) < spike_rate
            nsp = int(sp.sum())
            if nsp:
                pa[sp] = np.where(rng.random(nsp) < 0.5, rng.uniform(5.0, 38.0, nsp), rng.uniform(182.0, 250.0, nsp))
            scale = rng_range[r] / dig[r]
            raw = np.clip(np.rint(pa / scale - offset[r]), -32768, 32767).astype(np.int16)
            sig_parts.append(raw); sig_off.append(sig_off[-1] + L)
            ns = len(seq_codes)
            # PAF orientation. The sequence in the FASTQ is in basecall (5'->3') order; for RNA the signal is
            # 3'->5', i.e. the first matched base of the walk is the LAST base of the fetched sequence.
            if rna:
                seq_parts.append(BASES[seq_codes[::-1]]); tstart[r] = ns; tend[r] = 0
            else:
                seq_parts.append(BASES[seq_codes]); tstart[r] = 0; tend[r] = ns
            seq_off.append(seq_off[-1] + ns)
            op_n_parts.append(ops_n); op_t_parts.append(ops_t); op_off.append(op_off[-1] + len(ops_n))

    b = Batch(
        n_reads=n_reads,
        sig=np.concatenate(sig_parts) if sig_parts else np.zeros(0, np.int16), sig_off=np.asarray(sig_off, np.uint64),
        digitisation=dig, offset=offset, range=rng_range, query_start=qstart, target_start=tstart, target_end=tend,
        seq=np.concatenate(seq_parts) if seq_parts else np.zeros(0, np.uint8), seq_off=np.asarray(seq_off, np.uint64),
        op_n=np.concatenate(op_n_parts) if op_n_parts else np.zeros(0, np.uint32),
        op_t=np.concatenate(op_t_parts) if op_t_parts else np.zeros(0, np.uint8), op_off=np.asarray(op_off, np.uint64))
    return b.validate_host()


def make_batch_fast(n_reads=50000, read_len=4000, kind="rna004", seed=20251003, spike_rate=0.005, chunk_reads=5000, homopolymer_frac=0.0):
    """Vectorised generator for throughput-sized batches (matches only, no indels): same distributions as
    make_batch but without the per-read Python loop. homopolymer_frac: that share of the reads carries one 40-base run of a single
    base (SURVEY 8d cfg 3's "10 % homopolymer-rich reads": a handful of low-complexity k-mers far above sample_limit); 0 draws
    nothing extra, so the other workloads keep their streams."""
    rng = np.random.default_rng(seed)
    rna = kind == "rna004"
    mean_extra = 5.2 if rna else 1.5
    p_geom = 1.0 / (1.0 + mean_extra)
    cand = int(read_len / (5 * (1 + mean_extra)) * 3) + 64
    dig = np.full(n_reads, 2048.0)
    offset = rng.integers(-260, -229, n_reads).astype(np.float64)
    rng_range = rng.uniform(280.0, 285.0, n_reads)
    sig = np.empty(n_reads * read_len, np.int16)
    seq_parts, opn_parts, nbs = [], [], []
    for c0 in range(0, n_reads, chunk_reads):
        c1 = min(n_reads, c0 + chunk_reads); nr = c1 - c0
        dw = np.clip(5 * rng.geometric(p_geom, size=(nr, cand)), 5, 200).astype(np.int64)
        cs = np.cumsum(dw, axis=1)
        nb = np.minimum((cs < read_len).sum(axis=1) + 1, cand)
        col = np.arange(cand)[None, :]
        valid = col < nb[:, None]
        last = col == (nb[:, None] - 1)
        prev = np.where(nb > 1, cs[np.arange(nr), np.maximum(nb - 2, 0)], 0)
        dw = np.where(last, (read_len - prev)[:, None], dw)
        dw = np.where(valid, dw, 0)
        bases = rng.integers(0, 4, size=(nr, cand))
        if homopolymer_frac > 0:
            hp = np.flatnonzero(rng.random(nr) < homopolymer_frac)
            a = (rng.random(hp.size) * np.maximum(1, nb[hp] - 40)).astype(np.int64)
            run = rng.integers(0, 4, hp.size)
            cols = a[:, None] + np.arange(40)[None, :]
            bases[hp[:, None], np.minimum(cols, cand - 1)] = run[:, None]
        code = np.zeros((nr, cand), np.uint64)
        for t in range(5):
            sh = np.roll(bases, t, axis=1).astype(np.uint64); sh[:, :t] = 0
            code |= sh << np.uint64(2 * t)
        level = 70.0 + 60.0 * (_splitmix(code).astype(np.float64) / 2.0 ** 64)
        flat_d = dw[valid]; flat_l = level[valid]
        pa = np.repeat(flat_l, flat_d)
        assert pa.size == nr * read_len
        pa += rng.standard_normal(pa.size, dtype=np.float32) * np.float32(3.0)
        sp = np.flatnonzero(rng.random(pa.size, dtype=np.float32) < spike_rate)
        nsp = sp.size
        pa[sp] = np.where(rng.random(nsp) < 0.5, rng.uniform(5.0, 38.0, nsp), rng.uniform(182.0, 250.0, nsp))
        pa = pa.reshape(nr, read_len)
        pa /= (rng_range[c0:c1] / dig[c0:c1])[:, None]
        pa -= offset[c0:c1][:, None]
        np.rint(pa, out=pa); np.clip(pa, -32768, 32767, out=pa)
        sig[c0 * read_len:c1 * read_len] = pa.reshape(-1).astype(np.int16)
        opn_parts.append(flat_d.astype(np.uint32)); nbs.append(nb)
        if rna:  # fetched sequence is in basecall order: reverse of the walk order
            rev = np.where(valid, bases, -1)[:, ::-1]
            seq_parts.append(BASES[rev[rev >= 0]])
        else:
            seq_parts.append(BASES[bases[valid]])
    nb = np.concatenate(nbs).astype(np.int64)
    off_b = np.concatenate([[0], np.cumsum(nb)]).astype(np.uint64)
    op_n = np.concatenate(opn_parts)
    b = Batch(
        n_reads=n_reads, sig=sig, sig_off=(np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)),
        digitisation=dig, offset=offset, range=rng_range, query_start=np.zeros(n_reads, np.int32),
        target_start=(nb if rna else np.zeros(n_reads)).astype(np.int32), target_end=(np.zeros(n_reads) if rna else nb).astype(np.int32),
        seq=np.concatenate(seq_parts), seq_off=off_b, op_n=op_n, op_t=np.zeros(op_n.size, np.uint8), op_off=off_b.copy())
    return b.validate_host()


def ragged_lengths(total_samples=200_000_000, seed=20251005, lo=2000, hi=200_000, median=12_000, sigma=1.0, huge=1_000_000, huge_frac=0.001):
    """Read lengths of a ragged DNA run (SURVEY section 5 "long-context", section 7 "hard parts"): lognormal between lo and hi, `huge_frac` of
    the reads `huge` samples long, cut so that the lengths add up to total_samples exactly."""
    rng = np.random.default_rng(seed)
    out = []; tot = 0
    while tot < total_samples:
        L = np.clip(np.rint(median * np.exp(sigma * rng.standard_normal(4096))), lo, hi).astype(np.int64)
        L[rng.random(4096) < huge_frac] = huge
        out.append(L); tot += int(L.sum())
    L = np.concatenate(out)
    cs = np.cumsum(L)
    n = int(np.searchsorted(cs, total_samples, side="left")) + 1
    L = L[:n].copy()
    L[-1] -= int(cs[n - 1]) - total_samples
    if L[-1] < lo:  # fold a short remainder into the read in front of it
        L[-2] += L[-1]; L = L[:-1]
    assert int(L.sum()) == total_samples
    return L


def make_ragged_fast(lengths, kind="dna_r10", seed=20251005, spike_rate=0.005, chunk_samples=20_000_000) -> Batch:
    """make_batch_fast for reads of DIFFERENT lengths (matches only): one stream of dwells per chunk of consecutive reads, cut at the
    read boundaries (a dwell that straddles a boundary becomes two ops), same levels / noise / spikes / calibrations as make_batch_fast."""
    lengths = np.asarray(lengths, np.int64)
    n_reads = lengths.size
    rng = np.random.default_rng(seed)
    rna = kind == "rna004"
    mean_extra = 5.2 if rna else 1.5
    p_geom = 1.0 / (1.0 + mean_extra)
    dig = np.full(n_reads, 2048.0)
    offset = rng.integers(-260, -229, n_reads).astype(np.float64)
    rng_range = rng.uniform(280.0, 285.0, n_reads)
    total = int(lengths.sum())
    sig = np.empty(total, np.int16)
    sig_off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    opn_parts, base_parts, nops = [], [], []
    r0 = 0
    while r0 < n_reads:
        r1 = r0 + 1; acc = int(lengths[r0])
        while r1 < n_reads and acc + int(lengths[r1]) <= chunk_samples:
            acc += int(lengths[r1]); r1 += 1
        n_dw = int(acc / (5 * (1 + mean_extra)) * 1.3) + 1024
        cs = np.cumsum(np.clip(5 * rng.geometric(p_geom, size=n_dw), 5, 200).astype(np.int64))
        while cs[-1] < acc:  # (never with the 1.3 margin; keep the stream long enough anyway)
            cs = np.concatenate([cs, cs[-1] + np.cumsum(np.clip(5 * rng.geometric(p_geom, size=n_dw), 5, 200).astype(np.int64))])
        bounds = np.cumsum(lengths[r0:r1])                       # read ends inside the chunk's stream
        edges = np.unique(np.concatenate([cs[cs < acc], bounds]))  # op ends: dwell ends and read ends, the last one == acc
        d = np.diff(np.concatenate([[0], edges]))
        op_read = np.searchsorted(bounds, edges - d, side="right")  # read (inside the chunk) of every op
        bases = rng.integers(0, 4, size=d.size)
        code = np.zeros(d.size, np.uint64)
        for t in range(5):
            sh = np.roll(bases, t).astype(np.uint64); sh[:t] = 0
            code |= sh << np.uint64(2 * t)
        level = 70.0 + 60.0 * (_splitmix(code).astype(np.float64) / 2.0 ** 64)
        pa = np.repeat(level, d)
        pa += rng.standard_normal(pa.size, dtype=np.float32) * np.float32(3.0)
        sp = np.flatnonzero(rng.random(pa.size, dtype=np.float32) < spike_rate)
        pa[sp] = np.where(rng.random(sp.size) < 0.5, rng.uniform(5.0, 38.0, sp.size), rng.uniform(182.0, 250.0, sp.size))
        rr = np.repeat(np.arange(r0, r1), lengths[r0:r1])
        pa /= (rng_range / dig)[rr]
        pa -= offset[rr]
        np.rint(pa, out=pa); np.clip(pa, -32768, 32767, out=pa)
        sig[int(sig_off[r0]):int(sig_off[r1])] = pa.astype(np.int16)
        cnt = np.bincount(op_read, minlength=r1 - r0)
        if rna:  # the fetched sequence is in basecall order: the reverse of the walk order, read by read
            start = np.concatenate([[0], np.cumsum(cnt)])[:-1]
            idx = (start + cnt)[op_read] - 1 - (np.arange(d.size) - start[op_read])
            bases = bases[idx]
        opn_parts.append(d.astype(np.uint32)); base_parts.append(BASES[bases]); nops.append(cnt)
        r0 = r1
    nb = np.concatenate(nops).astype(np.int64)
    off_b = np.concatenate([[0], np.cumsum(nb)]).astype(np.uint64)
    op_n = np.concatenate(opn_parts)
    b = Batch(
        n_reads=n_reads, sig=sig, sig_off=sig_off, digitisation=dig, offset=offset, range=rng_range, query_start=np.zeros(n_reads, np.int32),
        target_start=(nb if rna else np.zeros(n_reads)).astype(np.int32), target_end=(np.zeros(n_reads) if rna else nb).astype(np.int32),
        seq=np.concatenate(base_parts), seq_off=off_b, op_n=op_n, op_t=np.zeros(op_n.size, np.uint8), op_off=off_b.copy())
    return b.validate_host()


def add_indels_fast(b: Batch, del_rate=0.02, ins_rate=0.02, seed=1, rna=True) -> Batch:
    """BASELINE configs[4]'s ss strings at throughput size: a matches-only batch (make_batch_fast) gets, vectorised, `del_rate` of its
    matches a deletion op `nD` in front (n in 1..3: bases of the fetched sequence without samples -- they are inserted into the
    sequence) and `ins_rate` of its matches (of 15 samples or more) an insertion op `nI` in front (n in 5..min(40, dwell - 5): samples
    without a base, taken from the match's own dwell, so every read still covers exactly its signal). The signal is shared with `b`."""
    rng = np.random.default_rng(seed)
    n_ops = int(b.op_off[-1])
    opn = b.op_n.astype(np.int64)
    has_d = rng.random(n_ops) < del_rate
    has_i = (rng.random(n_ops) < ins_rate) & (opn >= 15)
    n_d = np.where(has_d, rng.integers(1, 4, n_ops), 0)
    n_i = np.where(has_i, np.minimum(40, rng.integers(5, np.maximum(opn - 5, 6))), 0)
    # ops: per original match [D?] [I?] M
    cand_n = np.stack([n_d, n_i, opn - n_i], axis=1).reshape(-1)
    cand_t = np.tile(np.array([2, 1, 0], np.uint8), n_ops)
    keep = np.stack([has_d, has_i, np.ones(n_ops, bool)], axis=1).reshape(-1)
    op_n = cand_n[keep].astype(np.uint32); op_t = cand_t[keep]
    per_op = 1 + has_d.astype(np.int64) + has_i.astype(np.int64)
    ob = b.op_off.astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(per_op)])
    op_off = csum[ob].astype(np.uint64)
    # bases in WALK order: the deleted bases in front of the match's own base
    seq_off_old = b.seq_off.astype(np.int64)
    nb_old = np.diff(seq_off_old)
    read_of_op = np.repeat(np.arange(b.n_reads), np.diff(ob))
    pos_in_read = np.arange(n_ops) - ob[read_of_op]
    walk_old = b.seq[(seq_off_old[read_of_op] + (nb_old[read_of_op] - 1 - pos_in_read)) if rna else (seq_off_old[read_of_op] + pos_in_read)]
    per_base = 1 + n_d
    bsum = np.concatenate([[0], np.cumsum(per_base)])
    walk = BASES[rng.integers(0, 4, int(bsum[-1]))]
    walk[bsum[1:] - 1] = walk_old
    seq_off = bsum[ob].astype(np.uint64)
    nb = np.diff(seq_off.astype(np.int64))
    if rna:  # fetched sequence = reverse of the walk order, per read
        so = seq_off.astype(np.int64)
        r_of_b = np.repeat(np.arange(b.n_reads), nb)
        p = np.arange(int(bsum[-1])) - so[r_of_b]
        seq = np.empty_like(walk)
        seq[so[r_of_b] + (nb[r_of_b] - 1 - p)] = walk
    else:
        seq = walk
    out = Batch(n_reads=b.n_reads, sig=b.sig, sig_off=b.sig_off, digitisation=b.digitisation, offset=b.offset, range=b.range,
                query_start=b.query_start, target_start=(nb if rna else np.zeros(b.n_reads)).astype(np.int32),
                target_end=(np.zeros(b.n_reads) if rna else nb).astype(np.int32), seq=np.ascontiguousarray(seq), seq_off=seq_off,
                op_n=op_n, op_t=op_t, op_off=op_off)
    return out.validate_host()


OP_CHARS = ",ID"


def ss_string(b: Batch, r: int) -> str:
    a, e = int(b.op_off[r]), int(b.op_off[r + 1])
    return "".join(f"{int(n)}{OP_CHARS[int(t)]}" for n, t in zip(b.op_n[a:e], b.op_t[a:e]))


def seq_string(b: Batch, r: int) -> str:
    return b.seq[int(b.seq_off[r]):int(b.seq_off[r + 1])].tobytes().decode()


def write_files(b: Batch, prefix: str):
    """ASCII SLOW5 + PAF(ss) + FASTQ for the batch (formats: SURVEY.md Appendix B). Read ids are r<index>."""
    with open(prefix + ".slow5", "w") as f:
        f.write("#slow5_version\t0.2.0\n#num_read_groups\t1\n@asic_id\tsynthetic\n")
        f.write("#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n")
        f.write("#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n")
        for r in range(b.n_reads):
            s = b.sig[int(b.sig_off[r]):int(b.sig_off[r + 1])]
            f.write(f"r{r}\t0\t{b.digitisation[r]:.17g}\t{b.offset[r]:.17g}\t{b.range[r]:.17g}\t4000\t{len(s)}\t" + ",".join(map(str, s.tolist())) + "\n")
    with open(prefix + ".fastq", "w") as f:
        for r in range(b.n_reads):
            s = seq_string(b, r)
            f.write(f"@r{r} synthetic\n{s}\n+\n{'I' * len(s)}\n")
    with open(prefix + ".paf", "w") as f:
        for r in range(b.n_reads):
            L = int(b.sig_off[r + 1] - b.sig_off[r]); ns = int(b.seq_off[r + 1] - b.seq_off[r])
            f.write(f"r{r}\t{L}\t{int(b.query_start[r])}\t{L}\t+\tr{r}\t{ns}\t{int(b.target_start[r])}\t{int(b.target_end[r])}\t{ns}\t{ns}\t255\tss:Z:{ss_string(b, r)}\n")


def write_table_files(b: Batch, prefix: str, stride: int = 5, trim: int = 0, seed: int = 1):
    """ASCII SLOW5 + 7-column move table (read_id, fastq_len, seq, stride, moves, signal_len, trim_offset) for a
    DNA-oriented, match-only batch whose dwells are multiples of `stride` (format: src/gmove.cpp:570-577). With
    trim > 0 every read gets `trim` extra leading samples that the table tells gmove to ignore."""
    rng = np.random.default_rng(seed)
    assert np.all(b.op_t == 0) and np.all(b.target_start == 0)
    with open(prefix + ".slow5", "w") as f, open(prefix + ".table", "w") as t:
        f.write("#slow5_version\t0.2.0\n#num_read_groups\t1\n")
        f.write("#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n")
        for r in range(b.n_reads):
            s = b.sig[int(b.sig_off[r]):int(b.sig_off[r + 1])]
            lead = rng.integers(300, 1500, trim).astype(np.int16)
            full = np.concatenate([lead, s])
            f.write(f"r{r}\t0\t{b.digitisation[r]:.17g}\t{b.offset[r]:.17g}\t{b.range[r]:.17g}\t4000\t{len(full)}\t" + ",".join(map(str, full.tolist())) + "\n")
            d = b.op_n[int(b.op_off[r]):int(b.op_off[r + 1])]
            assert np.all(d % stride == 0) and np.all(d > 0)
            moves = "".join("1" + "0" * (int(x) // stride - 1) for x in d)
            seq = seq_string(b, r)
            t.write(f"r{r}\t{len(seq)}\t{seq}\t{stride}\t{moves}\t{len(full)}\t{trim}\n")
    # the same records as unaligned SAM with the basecaller's move tags (mv:B:c,<stride>,<moves>; ns; ts)
    with open(prefix + ".sam", "w") as f:
        f.write("@HD\tVN:1.6\tSO:unknown\n@PG\tID:synthetic\n")
        for r in range(b.n_reads):
            d = b.op_n[int(b.op_off[r]):int(b.op_off[r + 1])]
            mv = ",".join("1" + ",0" * (int(x) // stride - 1) for x in d)
            seq = seq_string(b, r)
            L = int(b.sig_off[r + 1] - b.sig_off[r]) + trim
            f.write(f"r{r}\t4\t*\t0\t0\t*\t*\t0\t0\t{seq}\t*\tmv:B:c,{stride},{mv}\tqs:i:10\tns:i:{L}\tts:i:{trim}\n")


def write_bam(b: Batch, path: str, stride: int = 5, trim: int = 0, block_bytes: int = 20000):
    """The records of write_table_files' SAM as a BAM file (BGZF: gzip members with the BC extra field; layout: SAM/BAM
    specification 4.2) -- unaligned reads, SEQ 4-bit packed, tags mv:B:c, qs:C, ns:I, ts:I. block_bytes < 64 KiB makes
    records straddle BGZF blocks, which is what a reader has to get right."""
    import struct
    import zlib
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    text = b"@HD\tVN:1.6\tSO:unknown\n@PG\tID:synthetic\n"
    raw = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 0))
    for r in range(b.n_reads):
        d = b.op_n[int(b.op_off[r]):int(b.op_off[r + 1])]
        mv = [stride]
        for x in d:
            mv += [1] + [0] * (int(x) // stride - 1)
        seq = seq_string(b, r)
        L = int(b.sig_off[r + 1] - b.sig_off[r]) + trim
        name = f"r{r}".encode() + b"\x00"
        packed = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i >> 1] |= code.get(ch, 15) << (4 if i % 2 == 0 else 0)
        tags = (b"mvBc" + struct.pack("<i", len(mv)) + struct.pack(f"<{len(mv)}b", *mv) + b"qsC" + struct.pack("<B", 10)
                + b"nsI" + struct.pack("<I", L) + b"tsI" + struct.pack("<I", trim))
        body = (struct.pack("<iiBBHHHiiii", -1, -1, len(name), 0, 4680, 0, 4, len(seq), -1, -1, 0) + name + bytes(packed)
                + b"\xff" * len(seq) + tags)
        raw += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        def block(data: bytes):
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = c.compress(data) + c.flush()
            bsize = 18 + len(comp) + 8 - 1
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp
                    + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))
        for o in range(0, len(raw), block_bytes):
            block(bytes(raw[o:o + block_bytes]))
        block(b"")  # the BGZF end-of-file marker


def _svb_zd(sig: np.ndarray) -> bytes:
    """svb-zd block of a signal: u32 count, then streamvbyte (Lemire: ceil(n/4) control bytes of 2 bits per value =
    byte length - 1, then the values' little-endian bytes) of the zig-zag-coded deltas of the samples."""
    import struct
    x = sig.astype(np.int64)
    d = np.diff(x, prepend=0)
    zz = ((d << 1) ^ (d >> 63)).astype(np.uint32)
    nb = np.where(zz < (1 << 8), 1, np.where(zz < (1 << 16), 2, np.where(zz < (1 << 24), 3, 4))).astype(np.uint8)
    n = zz.size
    codes = np.zeros((n + 3) // 4 * 4, np.uint8); codes[:n] = nb - 1
    ctrl = (codes[0::4] | (codes[1::4] << 2) | (codes[2::4] << 4) | (codes[3::4] << 6)).astype(np.uint8)
    le = zz.astype("<u4").view(np.uint8).reshape(-1, 4)
    data = le[np.arange(4)[None, :] < nb[:, None]]
    return struct.pack("<I", n) + ctrl.tobytes() + data.tobytes()


def zstd_compress(data: bytes, level: int = 3, streamed: bool = False):
    """One-shot zstd frame through the system's libzstd.so.1 (ctypes; there is no zstd module in this image). None if the library is absent.
    streamed: through ZSTD_compressStream instead -- a frame WITHOUT its content size in the header (what a writer that streams records produces)."""
    import ctypes as C
    global _zstd
    try:
        _zstd
    except NameError:
        try:
            _zstd = C.CDLL("libzstd.so.1")
            _zstd.ZSTD_compressBound.restype = C.c_size_t; _zstd.ZSTD_compressBound.argtypes = [C.c_size_t]
            _zstd.ZSTD_compress.restype = C.c_size_t
            _zstd.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
            _zstd.ZSTD_isError.restype = C.c_uint; _zstd.ZSTD_isError.argtypes = [C.c_size_t]
        except OSError:
            _zstd = None
    if _zstd is None:
        return None
    if streamed:
        class _B(C.Structure):
            _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]
        _zstd.ZSTD_createCStream.restype = C.c_void_p
        _zstd.ZSTD_initCStream.argtypes = [C.c_void_p, C.c_int]; _zstd.ZSTD_initCStream.restype = C.c_size_t
        _zstd.ZSTD_compressStream.argtypes = [C.c_void_p, C.POINTER(_B), C.POINTER(_B)]; _zstd.ZSTD_compressStream.restype = C.c_size_t
        _zstd.ZSTD_endStream.argtypes = [C.c_void_p, C.POINTER(_B)]; _zstd.ZSTD_endStream.restype = C.c_size_t
        _zstd.ZSTD_freeCStream.argtypes = [C.c_void_p]
        cs = _zstd.ZSTD_createCStream(); _zstd.ZSTD_initCStream(cs, level)
        src = C.create_string_buffer(data, len(data)); dst = C.create_string_buffer(_zstd.ZSTD_compressBound(len(data)) + 64)
        i = _B(C.cast(src, C.c_void_p), len(data), 0); o = _B(C.cast(dst, C.c_void_p), len(dst), 0)
        assert not _zstd.ZSTD_isError(_zstd.ZSTD_compressStream(cs, C.byref(o), C.byref(i))) and i.pos == len(data)
        assert _zstd.ZSTD_endStream(cs, C.byref(o)) == 0
        _zstd.ZSTD_freeCStream(cs)
        return dst.raw[:o.pos]
    cap = _zstd.ZSTD_compressBound(len(data))
    buf = C.create_string_buffer(cap)
    n = _zstd.ZSTD_compress(buf, cap, data, len(data), level)
    assert not _zstd.ZSTD_isError(n)
    return buf.raw[:n]


def write_blow5(b: Batch, path: str, compress=False):
    """BLOW5 for the batch (layout: SURVEY.md 8f-1); read ids are r<index>. compress=False: record compression none,
    signal compression none -- meant for throughput-sized CLI runs where ASCII SLOW5 parsing would dominate.
    compress=True: zlib records + svb-zd signals, what slow5tools writes by default (as test/example.blow5).
    compress="zstd": zstd records (record compression 2; the reference's `make zstd=1` build) + svb-zd signals; "zstd-stream": the same with
    frames that do not carry their content size."""
    import struct
    import zlib
    if compress:
        hdr = (b"#slow5_version\t0.2.0\n#num_read_groups\t1\n@asic_id\tsynthetic\n"
               b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n"
               b"#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n")
        with open(path, "wb") as f:
            f.write(b"BLOW5\x01" + bytes([0, 2, 0]) + bytes([2 if compress in ("zstd", "zstd-stream") else 1]) + struct.pack("<I", 1) + bytes([1]) + bytes(64 - 15))
            f.write(struct.pack("<I", len(hdr)) + hdr)
            for r in range(b.n_reads):
                rid = f"r{r}".encode()
                blk = _svb_zd(b.sig[int(b.sig_off[r]):int(b.sig_off[r + 1])])
                body = (struct.pack("<H", len(rid)) + rid + struct.pack("<I", 0)
                        + struct.pack("<dddd", b.digitisation[r], b.offset[r], b.range[r], 4000.0) + struct.pack("<Q", len(blk)) + blk)
                z = zstd_compress(body, streamed=(compress == "zstd-stream")) if compress in ("zstd", "zstd-stream") else zlib.compress(body)
                f.write(struct.pack("<Q", len(z)) + z)
            f.write(b"5WOLB")
        return
    hdr = (b"#slow5_version\t0.2.0\n#num_read_groups\t1\n@asic_id\tsynthetic\n"
           b"#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*\n"
           b"#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal\n")
    with open(path, "wb") as f:
        f.write(b"BLOW5\x01" + bytes([0, 2, 0]) + bytes([0]) + struct.pack("<I", 1) + bytes([0]) + bytes(64 - 15))
        f.write(struct.pack("<I", len(hdr)) + hdr)
        for r in range(b.n_reads):
            rid = f"r{r}".encode()
            sig = b.sig[int(b.sig_off[r]):int(b.sig_off[r + 1])]
            body = (struct.pack("<H", len(rid)) + rid + struct.pack("<I", 0)
                    + struct.pack("<dddd", b.digitisation[r], b.offset[r], b.range[r], 4000.0) + struct.pack("<Q", sig.size))
            f.write(struct.pack("<Q", len(body) + sig.size * 2) + body)
            f.write(sig.tobytes())
        f.write(b"5WOLB")


def write_paf_fastq(b: Batch, prefix: str):
    """PAF(ss) + FASTQ of the batch without the (slow) ASCII SLOW5."""
    with open(prefix + ".fastq", "w") as f:
        for r in range(b.n_reads):
            s = seq_string(b, r)
            f.write(f"@r{r} synthetic\n{s}\n+\n{'I' * len(s)}\n")
    with open(prefix + ".paf", "w") as f:
        for r in range(b.n_reads):
            L = int(b.sig_off[r + 1] - b.sig_off[r]); ns = int(b.seq_off[r + 1] - b.seq_off[r])
            f.write(f"r{r}\t{L}\t{int(b.query_start[r])}\t{L}\t+\tr{r}\t{ns}\t{int(b.target_start[r])}\t{int(b.target_end[r])}\t{ns}\t{ns}\t255\tss:Z:{ss_string(b, r)}\n")
