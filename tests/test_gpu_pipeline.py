"""
GPU suite: the device-resident cluster hot path (bin3c_amd/pipeline.py) against the reference's golden
pipeline fixture and against the oracle, with and without the locality-ordered layout.
"""
import os

import numpy as np
import pytest

import graphs
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_pipeline_matches_reference_fixture(ctx, oracle_lib):
    from bin3c_amd.pipeline import run_hot_path, processed_to_host
    g = np.load(os.path.join(GOLDEN, 'pipeline.npz'))
    counts = ctx.upload(g['indptr'], g['indices'], g['counts'])
    res = run_hot_path(ctx, counts, g['sites'], g['lengths'], int(g['min_len']), int(g['min_sig']), keep=True)
    assert np.array_equal(res.signal, g['signal'])            # integer work: bit-exact
    assert np.array_equal(res.mask, g['mask'])
    np.testing.assert_allclose(res.x, g['x'], rtol=1e-9)      # contract: 1e-6
    ip, ix, data = processed_to_host(ctx, res)
    assert np.array_equal(ip, g['proc_indptr']) and np.array_equal(ix, g['proc_indices'])
    np.testing.assert_allclose(data, g['proc_data'], rtol=1e-9)
    sip, six, sdata = res.sub.download()
    assert np.array_equal(sip, g['sub_indptr']) and np.array_equal(six, g['sub_indices'])
    np.testing.assert_allclose(sdata, g['sub_data'], rtol=1e-9)
    assert abs(res.scl - float(g['scl'])) <= 1e-9 * float(g['scl'])
    # labels: the oracle's schedule on the device-produced sub-map
    w = oracle_lib.graph_weights(sip, six, sdata, res.scl)
    ref = oracle_lib.louvain_bsp(sip, six, oracle_lib.quantise(w, oracle_lib.quant_shift(sip, six, w)), n_sub=4)
    assert np.array_equal(res.labels, ref['labels'])
    res.free()


@pytest.mark.parametrize('law', graphs.LAWS)
@pytest.mark.parametrize('n,e,seed', [(3000, 90000, 5), (40000, 1300000, 21)])
def test_pipeline_ordered_equals_unordered(ctx, oracle_lib, n, e, seed, law):
    """The locality-ordered layout is a pure performance device: every output is unchanged."""
    from bin3c_amd.pipeline import run_hot_path
    from bin3c_amd.synth import make_synth
    sm = make_synth(n, e, seed, genome_law=law)
    counts = ctx.upload(sm.indptr, sm.indices, sm.counts)
    r1 = run_hot_path(ctx, counts, sm.sites, sm.lengths, 1000, 5, reorder=True)
    r2 = run_hot_path(ctx, counts, sm.sites, sm.lengths, 1000, 5, reorder=False)
    assert np.array_equal(r1.mask, r2.mask) and np.array_equal(r1.signal, r2.signal)
    assert r1.kr_info.n_iter == r2.kr_info.n_iter
    np.testing.assert_allclose(r1.x, r2.x, rtol=1e-10)
    # x differs in the last bits (summation order), so compare partitions, not raw labels
    if law == 'sqrt':
        assert graphs.same_partition(r1.labels, r2.labels)
        assert graphs.same_partition(r1.labels, sm.genome[r1.mask])
    else:
        # (the contract law's giant genome is a flat landscape: last-bit differences of x may tip single contigs)
        assert graphs.adjusted_rand(r1.labels, r2.labels) > 0.98
        assert graphs.adjusted_rand(r1.labels, sm.genome[r1.mask]) > 0.98
    # against the oracle end to end
    sig = oracle_lib.max_offdiag_u32(sm.indptr, sm.indices, sm.counts)
    assert np.array_equal(r1.signal, sig)
    nd = oracle_lib.site_norm(sm.indptr, sm.indices, sm.counts.astype(np.float64), sm.sites)
    rc, xo, st = oracle_lib.kr_balance(sm.indptr, sm.indices, nd)
    assert st.n_iter == r1.kr_info.n_iter
    np.testing.assert_allclose(r1.x, xo, rtol=1e-9)


@pytest.mark.parametrize('drop', [0.0, 0.03, 0.2, 0.6])
def test_fused_sub_map_in_graph_form_counts_from_the_dropped_rows(ctx, oracle_lib, drop):
    """b3c_normalise_scale_compress in graph form (the call of the cluster step).  With few rows dropped its first pass finds the kept
    entries of every row from the DROPPED rows' side (round 6: row j dropped and (j, i) stored <=> (i, j) stored -- the map is
    structurally symmetric); with many, by the mask gather.  Either way: structure == compress of the oracle, the lower triangle bit
    for bit, the upper one mirroring it."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(20000, 600000, 13)
    nd = oracle_lib.site_norm(sm.indptr, sm.indices, sm.counts.astype(np.float64), sm.sites)
    rc, xo, st = oracle_lib.kr_balance(sm.indptr, sm.indices, nd)
    rs = np.random.RandomState(int(drop * 100) + 1)
    mask = rs.rand(sm.n) >= drop
    proc = oracle_lib.scale_sym(sm.indptr, sm.indices, nd, xo)
    m, sip, six, sa = oracle_lib.compress(sm.indptr, sm.indices, proc, mask)
    counts = ctx.upload(sm.indptr, sm.indices, sm.counts)
    sub = ctx.normalise_scale_compress(counts, sm.sites, xo, mask, graph_form=True)
    dip, dix, da = sub.download()
    assert sub.n == m and np.array_equal(dip, sip) and np.array_equal(dix, six)
    lower = np.repeat(np.arange(m), np.diff(sip)) >= six
    assert np.array_equal(da[lower], sa[lower])
    import scipy.sparse as sp
    g = sp.csr_matrix((da, dix, dip), shape=(m, m))
    assert (g != g.T).nnz == 0                       # both triangles carry the lower one's value
    sub.free()
    counts.free()
