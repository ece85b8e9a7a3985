"""
hic-synth-v1: deterministic synthetic Hi-C contact maps for the KR-normalise + Louvain hot path.

The reference has no sample data and its ingest (``mkmap``: pysam/BioPython, contact_map.py:488-809)
is out of scope, so every test and bench input comes from here.  The OUTPUT CONTRACT is the one the
reference's ingest produces (sparse_utils.py:246-266, ``Sparse2DAccumulator.get_coo``): a square
symmetric uint32 matrix without duplicate coordinates whose diagonal holds self-contacts, plus
per-contig ``length`` and ``sites`` (contact_map.py:19 ``SeqInfo``).

Model (SURVEY.md section 8(d) -- the CONTRACT law, the default since round 6): ``G = max(2, n // 200)`` genomes whose contig
counts follow Zipf(1.5) (one genome holds 38 % of the contigs, most hold a handful); ``genome_law='sqrt'`` gives the variant
rounds 1-4 were tuned on (``G = max(4, n // 500)``, sizes ~ rank**-0.5).  Contig ids randomly permuted (no id locality), contig length
~ lognormal(ln 8000, 1) clipped >= 500 (about 2 % fall under the default min_len of 1000), sites = max(1, Poisson(length / 256)); ~90 % of edges join
two contigs of one genome, ~10 % are uniform noise; pair counts Geometric(0.2) x 5 (intra) or x 1
(noise), duplicates summed; diagonal Geometric(0.02) with 5 % forced zeros (exercises the
zero-diagonal substitution of sparse_utils.py:110-115).
"""
import numpy as np

__all__ = ['SynthMap', 'make_edges', 'make_synth', 'CONFIGS', 'DEFAULT_GENOME_LAW']

# genome-size law of every generator call that does not name one: SURVEY.md 8(d)'s (the contract workload)
DEFAULT_GENOME_LAW = 'survey'

# BASELINE.json configs -> (n contigs, E undirected edges, seed, degree law)
CONFIGS = {
    'C1': (1000, 50000, 1, 'flat'),
    'C2': (100000, 5000000, 2, 'flat'),
    'C3': (1000000, 100000000, 3, 'flat'),
    'C5': (5000000, 1000000000, 5, 'powerlaw'),
}


class SynthMap(object):
    """Plain container: symmetric CSR (int64 indptr, int32 indices, uint32 counts) + contig tables."""

    def __init__(self, n, indptr, indices, counts, lengths, sites, genome):
        self.n = n
        self.indptr = indptr
        self.indices = indices
        self.counts = counts
        self.lengths = lengths
        self.sites = sites
        self.genome = genome

    @property
    def nnz(self):
        return int(self.indptr[-1])

    @property
    def n_edges(self):
        """Undirected off-diagonal edges E (stored nnz = 2E + number of non-zero diagonals)."""
        rows = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.indptr))
        return int(np.count_nonzero(rows < self.indices))

    def to_scipy(self, dtype=np.uint32):
        import scipy.sparse as sp
        return sp.csr_matrix((self.counts.astype(dtype), self.indices, self.indptr), shape=(self.n, self.n))


def _genome_sizes(n, n_genomes, exponent=0.5):
    w = np.arange(1, n_genomes + 1, dtype=np.float64) ** -exponent
    sizes = np.maximum(2, np.floor(w / w.sum() * n).astype(np.int64))
    # fix rounding drift on the largest genome
    sizes[0] += n - sizes.sum()
    assert sizes[0] >= 2 and sizes.sum() == n
    return sizes


def _solve_draws(target_unique, pair_cap, share):
    """Number of draws K such that sum_g cap_g * (1 - exp(-K share_g / cap_g)) ~= target_unique."""
    lo, hi = float(target_unique), float(target_unique) * 64.0
    if np.sum(pair_cap) <= target_unique:
        return hi
    for _ in range(80):
        mid = 0.5 * (lo + hi)
        got = np.sum(pair_cap * -np.expm1(-mid * share / pair_cap))
        if got < target_unique:
            lo = mid
        else:
            hi = mid
    return hi


def genome_layout(n, genomes=None, genome_law=None):
    """Genome sizes of a map.  'survey' (SURVEY.md 8(d) to the letter; the default, DEFAULT_GENOME_LAW): G = max(2, n // 200),
    sizes ~ Zipf(1.5) -- 38 % of the contigs in the largest genome, most genomes of a handful.  'sqrt' (the variant rounds 1-4
    benchmarked): G = max(4, n // 500) genomes, sizes ~ rank**-0.5.  `genomes` overrides G (tests: thousands of small genomes)."""
    if genome_law is None:
        genome_law = DEFAULT_GENOME_LAW
    if genome_law == 'survey':
        g = max(2, n // 200) if genomes is None else int(genomes)
        return _genome_sizes(n, min(g, n // 2), 1.5)
    if genome_law != 'sqrt':
        raise ValueError('unknown genome_law {}'.format(genome_law))
    g = max(4, n // 500) if genomes is None else int(genomes)
    return _genome_sizes(n, min(g, n // 2), 0.5)


def make_edges(n, n_edges, seed, degree_law='flat', noise=0.1, intra_boost=5, genomes=None, genome_law=None):
    """
    Draw the contact pair stream.  `noise`: share of the edges that join two random contigs (inter-genome); `intra_boost`:
    multiplier of the intra-genome counts.  The defaults are hic-synth-v1; larger noise / boost 1 give a HARD clustering
    problem (modularity ~0.4-0.6, several Louvain levels) for the tests.  `genomes` / `genome_law`: see genome_layout.

    :return: (i, j, count) with i < j, possibly holding duplicates (the consumer sums them),
             diag counts (uint32[n]), lengths (int32[n]), sites (float64[n]), genome id (int32[n])
    """
    rs = np.random.RandomState(seed)
    sizes = genome_layout(n, genomes, genome_law)
    n_genomes = sizes.shape[0]
    starts = np.concatenate(([0], np.cumsum(sizes)[:-1]))
    perm = rs.permutation(n).astype(np.int64)       # position-in-genome-order -> contig id
    genome = np.empty(n, dtype=np.int32)
    genome[perm] = np.repeat(np.arange(n_genomes, dtype=np.int32), sizes)

    lengths = np.maximum(500, rs.lognormal(np.log(8000.0), 1.0, n)).astype(np.int32)
    sites = np.maximum(1, rs.poisson(lengths / 256.0)).astype(np.float64)

    n_intra = int(round((1.0 - noise) * n_edges))
    n_noise = n_edges - n_intra
    cap = sizes.astype(np.float64) * (sizes - 1) / 2.0
    share = sizes / float(sizes.sum())
    k_intra = _solve_draws(n_intra, cap, share)
    per_genome = rs.multinomial(int(k_intra), share)

    g_of = np.repeat(np.arange(n_genomes, dtype=np.int64), per_genome)
    m = g_of.shape[0]
    if degree_law == 'flat':
        u = rs.random_sample(m)
        v = rs.random_sample(m)
    elif degree_law == 'powerlaw':
        # rank**-0.8 endpoint preference inside a genome: inverse CDF of a continuous power law
        u = rs.random_sample(m) ** 5.0
        v = rs.random_sample(m) ** 5.0
    else:
        raise ValueError('unknown degree_law {}'.format(degree_law))
    sz = sizes[g_of]
    a = np.minimum((u * sz).astype(np.int64), sz - 1)
    b = np.minimum((v * sz).astype(np.int64), sz - 1)
    del u, v
    keep = a != b
    a = perm[starts[g_of] + a][keep]
    b = perm[starts[g_of] + b][keep]
    del g_of, sz
    c_intra = (rs.geometric(0.2, a.shape[0]) * intra_boost).astype(np.uint32)

    na = rs.randint(0, n, int(n_noise * 1.01) + 8).astype(np.int64)
    nb = rs.randint(0, n, na.shape[0]).astype(np.int64)
    keep = na != nb
    na, nb = na[keep], nb[keep]
    c_noise = rs.geometric(0.2, na.shape[0]).astype(np.uint32)

    i = np.concatenate((a, na))
    j = np.concatenate((b, nb))
    c = np.concatenate((c_intra, c_noise))
    lo = np.minimum(i, j)
    hi = np.maximum(i, j)

    diag = rs.geometric(0.02, n).astype(np.uint32)
    diag[rs.random_sample(n) < 0.05] = 0
    return lo, hi, c, diag, lengths, sites, genome


def symmetrise_to_csr(n, lo, hi, c, diag):
    """
    Sum duplicate (lo, hi) pairs, mirror to a full symmetric matrix and append the non-zero diagonal:
    the host (numpy) twin of what ``Sparse2DAccumulator.get_coo`` (sparse_utils.py:246-266) hands on.
    """
    key = lo * np.int64(n) + hi
    order = np.argsort(key, kind='stable')
    key = key[order]
    c = c[order].astype(np.uint64)
    first = np.concatenate(([True], key[1:] != key[:-1]))
    ukey = key[first]
    seg = np.flatnonzero(first)
    uc = np.add.reduceat(c, seg) if c.shape[0] else c
    uc = np.minimum(uc, np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ui = ukey // n
    uj = ukey % n
    dnz = np.flatnonzero(diag)
    rows = np.concatenate((ui, uj, dnz))
    cols = np.concatenate((uj, ui, dnz))
    vals = np.concatenate((uc, uc, diag[dnz]))
    order = np.argsort(rows * np.int64(n) + cols, kind='stable')
    rows = rows[order]
    indices = cols[order].astype(np.int32)
    counts = vals[order]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=indptr[1:])
    return indptr, indices, counts


def make_synth(n, n_edges, seed, degree_law='flat', noise=0.1, intra_boost=5, genomes=None, genome_law=None):
    """Full host-side generation -> :class:`SynthMap`."""
    lo, hi, c, diag, lengths, sites, genome = make_edges(n, n_edges, seed, degree_law, noise, intra_boost, genomes, genome_law)
    indptr, indices, counts = symmetrise_to_csr(n, lo, hi, c, diag)
    return SynthMap(n, indptr, indices, counts, lengths, sites, genome)
