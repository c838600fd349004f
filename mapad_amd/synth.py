"""Synthetic genomes / reads for benchmarks and parity tests.

Genome: i.i.d. uniform ACGT from a splitmix64 stream.  Reads: 90 % endogenous (uniform position and strand,
per-base substitutions), 10 % exogenous (i.i.d. random); optional ss-library C->T damage, variable lengths, indels.
Deterministic for a given seed; pure numpy (host-side plumbing, not part of the timed path).
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = np.arange(256, dtype=np.uint8)
for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[a] = b


def splitmix64(seed, n):
    """n uint64 values of the splitmix64 sequence started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def genome(n_bp, seed=1234):
    out = np.empty(n_bp, dtype=np.uint8)
    chunk = 1 << 24
    for s in range(0, n_bp, chunk):
        m = min(chunk, n_bp - s)
        r = splitmix64(seed + 0x1000003 * (s // chunk), (m + 31) // 32)
        shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
        codes = ((r[:, None] >> shifts) & np.uint64(3)).astype(np.uint8).reshape(-1)[:m]
        out[s:s + m] = _ACGT[codes]
    return out


def revcomp(a):
    return _COMP[a[::-1]]


def reads(genome_arr, n_reads, length=50, seed=4321, subst_rate=0.02, exo_frac=0.10, qual=40, qual_range=None,
          damage=None, len_range=None, indel_frac=0.0):
    """Returns (seqs uint8[total], quals uint8[total], offsets uint64[n+1]).

    damage: None or dict(f=, t=, d=, s=) for the single-stranded C->T model (README example).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    G = len(genome_arr)
    lens = np.full(n_reads, length, dtype=np.int64) if len_range is None else rng.integers(len_range[0], len_range[1] + 1, n_reads)
    offsets = np.zeros(n_reads + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    total = int(offsets[-1])
    seqs = np.empty(total, dtype=np.uint8)
    exo = rng.random(n_reads) < exo_frac
    pos = rng.integers(0, G - int(lens.max()) - 4, n_reads)
    strand = rng.random(n_reads) < 0.5
    if len_range is None and indel_frac == 0.0:
        idx = pos[:, None] + np.arange(length)[None, :]
        m = genome_arr[idx]
        rc = _COMP[m[:, ::-1]]
        m = np.where(strand[:, None], rc, m)
        sub = rng.random((n_reads, length)) < subst_rate
        shift = rng.integers(1, 4, (n_reads, length)).astype(np.uint8)
        code = np.searchsorted(_ACGT, m).astype(np.uint8)
        m = np.where(sub, _ACGT[(code + shift) & 3], m)
        if damage is not None:
            i = np.arange(length)
            p_fwd = damage["f"] ** (i + 1) + damage["t"] ** (length - i) - damage["f"] ** (i + 1) * damage["t"] ** (length - i)
            p_ct = damage["s"] * p_fwd + damage["d"] * (1 - p_fwd)
            deam = (m == ord("C")) & (rng.random((n_reads, length)) < p_ct[None, :])
            m = np.where(deam, np.uint8(ord("T")), m)
        rnd = _ACGT[rng.integers(0, 4, (n_reads, length))]
        m = np.where(exo[:, None], rnd, m)
        seqs[:] = m.reshape(-1)
    else:
        for r in range(n_reads):
            L = int(lens[r])
            if exo[r]:
                s = _ACGT[rng.integers(0, 4, L)]
            else:
                s = genome_arr[pos[r]:pos[r] + L + 2].copy()
                if rng.random() < indel_frac and L > 14:
                    at = int(rng.integers(5, L - 7))
                    k = int(rng.integers(1, 3))
                    if rng.random() < 0.5:
                        s = np.concatenate([s[:at], s[at + k:]])  # deletion from the read
                    else:
                        s = np.concatenate([s[:at], _ACGT[rng.integers(0, 4, k)], s[at:]])
                s = s[:L]
                if strand[r]:
                    s = revcomp(s)
                sub = rng.random(L) < subst_rate
                code = np.searchsorted(_ACGT, s)
                s = np.where(sub, _ACGT[(code + rng.integers(1, 4, L)) & 3], s)
                if damage is not None:
                    i = np.arange(L)
                    p_fwd = damage["f"] ** (i + 1) + damage["t"] ** (L - i) - damage["f"] ** (i + 1) * damage["t"] ** (L - i)
                    p_ct = damage["s"] * p_fwd + damage["d"] * (1 - p_fwd)
                    s = np.where((s == ord("C")) & (rng.random(L) < p_ct), np.uint8(ord("T")), s)
            seqs[int(offsets[r]):int(offsets[r + 1])] = s
    if qual_range is None:
        quals = np.full(total, qual, dtype=np.uint8)
    else:
        quals = rng.integers(qual_range[0], qual_range[1] + 1, total).astype(np.uint8)
    return seqs, quals, offsets
