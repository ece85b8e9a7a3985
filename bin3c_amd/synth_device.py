"""
hic-synth-v1 on the device, for maps the host generator cannot reach: the pair stream is drawn, de-duplicated and summed
in HBM (torch: input synthesis, not the product path) and handed to b3c_coo_symmetrise.

Why it exists.  `synth.make_edges` solves the number of draws for a UNIFORM choice of endpoints inside a genome.  Under
the power law (endpoint preference ~ rank**-0.8, BASELINE.json configs[4]) most draws land on hub pairs that were drawn
already: 1e9 draws at 5M contigs leave 4.6e8 unique edges, and the stated 1e9 unique edges need about six times as many
draws -- minutes of numpy and >100 GB of host memory.  Here batches of draws are merged into a sorted (pair, count) set
until the number of UNIQUE intra-genome edges hits its target (within `tol`), each batch sized from the yield of the one
before.  The model is the host generator's: same genome layout, contig tables (numpy RandomState(seed): permutation,
lengths, sites), 90 % intra-genome / 10 % uniform noise, counts Geometric(0.2) x 5 | x 1 -- one per unique pair by default, or
with duplicates summed like the host generator (`dup`, see make_counts_device; saturating at 2^32 - 1), diagonal Geometric(0.02) with 5 % zeros.  The draws themselves come from torch's Philox
generator on the device (seeded), so a map is reproducible on this software stack but is not the host generator's map.
"""
import numpy as np

from .synth import genome_layout

__all__ = ['make_counts_device']


def _merge(keys, cnts, k2, c2):
    """two (sorted unique keys, int64 counts) sets -> their union with counts of equal keys summed"""
    import torch
    if keys is None:
        return k2, c2
    k = torch.cat((keys, k2))
    c = torch.cat((cnts, c2))
    del keys, cnts, k2, c2
    k, order = torch.sort(k)
    c = c[order]
    del order
    first = torch.ones_like(k, dtype=torch.bool)
    first[1:] = k[1:] != k[:-1]
    seg = torch.cumsum(first, 0) - 1
    uk = k[first]
    uc = torch.zeros(uk.shape[0], dtype=torch.int64, device=k.device)
    uc.index_add_(0, seg, c)                  # integer sums: exact in any order
    return uk, uc


def _reduce_batch(key, c):
    import torch
    key, order = torch.sort(key)
    c = c[order]
    del order
    first = torch.ones_like(key, dtype=torch.bool)
    first[1:] = key[1:] != key[:-1]
    seg = torch.cumsum(first, 0) - 1
    uk = key[first]
    uc = torch.zeros(uk.shape[0], dtype=torch.int64, device=key.device)
    uc.index_add_(0, seg, c)
    return uk, uc


def make_counts_device(ctx, n, n_edges, seed, degree_law='powerlaw', noise=0.1, intra_boost=5, genomes=None,
                       genome_law=None, batch=1 << 28, tol=0.005, max_batches=200, verbose=False, extra=None, dup='first', draws_only=False):
    """
    :param dup: what a pair drawn more than once keeps.  'sum': the sum of its draws' counts, as the host generator does -- under
                the power law a hub pair is then drawn thousands of times and carries a count of 10^5..10^6 beside tail pairs of 5,
                and Knight-Ruiz needs TENS OF THOUSANDS of iterations at the stated edge counts (measured: 400k contigs / 1.6e7
                edges > 5,000; 4M / 3e8 > 20,000 -- the reference's limit is 1,000).  'first' (default): ONE count per unique
                pair, Geometric(0.2) x boost, drawn after de-duplication: the DEGREES follow the power law (hub contigs with
                10^5..10^6 partners), the counts do not explode, and the balancing stays in the hundreds of iterations.
    :param draws_only: stop after as many intra-genome DRAWS as intra-genome edges were asked for, however few unique pairs they
                       leave (the host generator's behaviour; with dup='sum' this is round 2's C5: 1e9 draws -> 4.6e8 edges)
    :param extra: optional (lo, hi, count) numpy arrays of further pairs (lo < hi) merged into the stream (tests: planted hubs)
    :return: (counts Mat in HBM, E = unique undirected off-diagonal edges, lengths int32[n], sites float64[n],
              genome int32[n], info dict with the draws it took)
    """
    import torch
    dev = torch.device('cuda', ctx.device)
    rs = np.random.RandomState(seed)
    sizes = genome_layout(n, genomes, genome_law)
    n_genomes = sizes.shape[0]
    starts = np.concatenate(([0], np.cumsum(sizes)[:-1]))
    perm = rs.permutation(n).astype(np.int64)
    genome = np.empty(n, dtype=np.int32)
    genome[perm] = np.repeat(np.arange(n_genomes, dtype=np.int32), sizes)
    lengths = np.maximum(500, rs.lognormal(np.log(8000.0), 1.0, n)).astype(np.int32)
    sites = np.maximum(1, rs.poisson(lengths / 256.0)).astype(np.float64)
    diag = rs.geometric(0.02, n).astype(np.uint32)
    diag[rs.random_sample(n) < 0.05] = 0

    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed) * 1000003 + 17)
    sizes_t = torch.as_tensor(sizes, device=dev)
    starts_t = torch.as_tensor(starts, device=dev)
    perm_t = torch.as_tensor(perm, device=dev)
    cum = torch.cumsum(sizes_t.double() / float(sizes.sum()), 0)
    cum[-1] = 1.0

    def rand(m):
        return torch.rand(m, dtype=torch.float64, device=dev, generator=gen)

    def geometric(m, p):
        # inverse CDF: floor(log(U) / log(1 - p)) + 1, support 1, 2, ...
        u = rand(m).clamp_(min=1e-300)
        return (torch.log(u) / np.log1p(-p)).floor_().long() + 1

    n_intra = int(round((1.0 - noise) * n_edges))
    n_noise = n_edges - n_intra
    cap = float(np.sum(sizes.astype(np.float64) * (sizes - 1) / 2.0))
    if n_intra > 0.9 * cap:
        raise ValueError('asked for {} intra-genome edges, the genomes hold {:.3g} pairs'.format(n_intra, cap))
    keys = cnts = None
    draws = 0
    nu = 0
    m = int(min(batch, max(1 << 16, n_intra)))
    info = {'batches': 0}
    while n_intra > 0:
        g = torch.searchsorted(cum, rand(m)).clamp_(max=n_genomes - 1)
        sz = sizes_t[g]
        u, v = rand(m), rand(m)
        if degree_law == 'powerlaw':
            u, v = u ** 5.0, v ** 5.0            # rank**-0.8 endpoint preference: inverse CDF of a continuous power law
        elif degree_law != 'flat':
            raise ValueError('unknown degree_law {}'.format(degree_law))
        a = torch.minimum((u * sz).long(), sz - 1)
        b = torch.minimum((v * sz).long(), sz - 1)
        del u, v
        keep = a != b
        base = starts_t[g]
        a = perm_t[(base + a)[keep]]
        b = perm_t[(base + b)[keep]]
        del g, sz, base, keep
        key = torch.minimum(a, b) * n + torch.maximum(a, b)
        del a, b
        if dup == 'sum':
            c = geometric(key.shape[0], 0.2) * intra_boost
        else:
            c = torch.zeros(key.shape[0], dtype=torch.int64, device=dev)       # counts are drawn per unique pair below
        k2, c2 = _reduce_batch(key, c)
        del key, c
        keys, cnts = _merge(keys, cnts, k2, c2)
        draws += m
        gained = keys.shape[0] - nu
        nu = keys.shape[0]
        info['batches'] += 1
        if verbose:
            print('[synth] batch {}: {} draws -> {} unique intra edges (target {})'.format(info['batches'], m, nu, n_intra), flush=True)
        if nu >= n_intra * (1.0 - tol) or info['batches'] >= max_batches or (draws_only and draws >= n_intra):
            break
        # the next batch from this one's marginal yield (it only falls: aim slightly short, the loop tops up)
        per_draw = max(gained / float(m), 1e-6)
        m = int(min(batch, max(1 << 16, (n_intra - nu) / per_draw)))
        if draws_only:
            m = int(min(batch, n_intra - draws))
    info['intra_draws'] = draws
    info['intra_unique'] = nu
    if dup != 'sum' and keys is not None:
        cnts = geometric(keys.shape[0], 0.2) * intra_boost                       # one count per unique intra-genome pair
    if n_noise > 0:
        na = torch.randint(0, n, (int(n_noise * 1.01) + 8,), device=dev, generator=gen)
        nb = torch.randint(0, n, (na.shape[0],), device=dev, generator=gen)
        keep = na != nb
        na, nb = na[keep], nb[keep]
        key = torch.minimum(na, nb) * n + torch.maximum(na, nb)
        del na, nb, keep
        if dup == 'sum':
            k2, c2 = _reduce_batch(key, geometric(key.shape[0], 0.2))
        else:
            k2 = torch.unique(key)
            c2 = geometric(k2.shape[0], 0.2)
        del key
        keys, cnts = _merge(keys, cnts, k2, c2)
    if extra is not None:
        xl = torch.as_tensor(np.asarray(extra[0], dtype=np.int64), device=dev)
        xh = torch.as_tensor(np.asarray(extra[1], dtype=np.int64), device=dev)
        xc = torch.as_tensor(np.asarray(extra[2], dtype=np.int64), device=dev)
        k2, c2 = _reduce_batch(xl * n + xh, xc)
        keys, cnts = _merge(keys, cnts, k2, c2)
    lo = keys // n
    hi = keys - lo * n
    del keys
    cnt32 = cnts.clamp_(max=0xFFFFFFFF)
    cnt32 = torch.where(cnt32 >= (1 << 31), cnt32 - (1 << 32), cnt32).to(torch.int32)      # uint32 bit pattern
    del cnts
    torch.cuda.synchronize(dev)
    counts, e_true = ctx.coo_symmetrise(n, lo, hi, cnt32, diag)
    del lo, hi, cnt32
    torch.cuda.empty_cache()
    info['edges'] = e_true
    return counts, e_true, lengths, sites, genome, info
