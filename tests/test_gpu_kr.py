"""
GPU suite: the HIP Knight-Ruiz path, called through the C ABI, against the reference's golden
vectors and against the CPU oracle.  Bar (BASELINE.json north_star): scale vectors within 1e-6
relative of the reference; we also require identical iteration counts and 1e-9 on the fixtures.
"""
import glob
import logging
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

KR_CASES = sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(GOLDEN, 'kr_*.npz')))
RTOL_X = 1e-9   # fixtures; the contract is 1e-6


@pytest.mark.parametrize('case', KR_CASES)
def test_kr_matches_reference_golden(ctx, case):
    g = np.load(os.path.join(GOLDEN, 'kr_{}.npz'.format(case)))
    mat = ctx.upload(g['in_indptr'], g['in_indices'], g['in_data'])
    rc, x, info = ctx.kr_balance(mat, tol=float(g['tol']), max_iter=int(g['max_iter']), check=False)
    err = str(g['error'])
    assert (info.n_zero_diag > 0) == bool(int(g['warn_zero_diag']))
    assert bool(info.is_symmetric) == (not int(g['warn_asym']))
    if err:
        from bin3c_amd import _native
        assert rc == _native.E_NOCONV
        assert ctx.lib.b3c_last_error(ctx.h).decode() == err
        return
    assert rc == 0
    assert info.n_iter == int(g['n_iter'])
    assert bool(info.hit_max_iter) == bool(int(g['warn_maxiter']))
    np.testing.assert_allclose(x, g['x'], rtol=RTOL_X, atol=0)
    ctx.scale_sym(mat, g['x'])
    _, _, data = mat.download(structure=False)
    if data.shape == g['out_data'].shape:
        assert np.array_equal(data, g['out_data'])   # x_i * (a_ij * x_j): two multiplies, bit-exact
    mat.free()


def test_kr_bound_branches_taken_on_device(ctx):
    g = np.load(os.path.join(GOLDEN, 'kr_delta_bound.npz'))
    _, _, info = ctx.kr_balance(ctx.upload(g['in_indptr'], g['in_indices'], g['in_data']))
    assert info.hit_delta > 0
    g = np.load(os.path.join(GOLDEN, 'kr_bigDelta_bound.npz'))
    _, _, info = ctx.kr_balance(ctx.upload(g['in_indptr'], g['in_indices'], g['in_data']))
    assert info.hit_Delta > 0


def test_kr_python_surface_like_reference(ctx, caplog):
    """mzd.sparse_utils.kr_biostochastic: same signature, return types, log lines and exceptions."""
    from bin3c_amd.mzd import sparse_utils
    g = np.load(os.path.join(GOLDEN, 'kr_zero_diag.npz'))
    n = g['in_indptr'].shape[0] - 1
    m = sp.csr_matrix((g['in_data'], g['in_indices'], g['in_indptr']), shape=(n, n))
    with caplog.at_level(logging.DEBUG, logger='bin3c_amd.mzd.sparse_utils'):
        b, x = sparse_utils.kr_biostochastic(m.tocoo(), ctx=ctx)
    assert sp.isspmatrix_csr(b) and b.shape == (n, n)
    np.testing.assert_allclose(x, g['x'], rtol=RTOL_X)
    msgs = [r.getMessage() for r in caplog.records]
    assert 'treating 40 zeros on diagonal as ones' in msgs
    assert 'It took {} iterations to achieve bistochasticity'.format(int(g['n_iter'])) in msgs
    # rows of the balanced WORKING matrix sum to one; the returned one scales the original (zero diag stays zero)
    np.testing.assert_allclose(np.asarray(b.sum(axis=1)).ravel() + (m.diagonal() == 0) * x * x, 1.0, atol=5e-6)
    assert (m != m).nnz == 0 and np.array_equal(m.data, g['in_data'])   # input untouched (:106)
    with pytest.raises(AssertionError):
        sparse_utils.kr_biostochastic(np.eye(3), ctx=ctx)
    with pytest.raises(AssertionError):
        sparse_utils.kr_biostochastic(sp.csr_matrix(np.ones((2, 3))), ctx=ctx)
    g = np.load(os.path.join(GOLDEN, 'kr_maxiter_raise.npz'))
    n = g['in_indptr'].shape[0] - 1
    m = sp.csr_matrix((g['in_data'], g['in_indices'], g['in_indptr']), shape=(n, n))
    with pytest.raises(RuntimeError, match=str(g['error'])):
        sparse_utils.kr_biostochastic(m, max_iter=int(g['max_iter']), ctx=ctx)


def test_kr_vs_oracle_synthetic_and_deterministic(ctx, oracle_lib):
    """Seeded synthetic contact map: device == oracle within 1e-9, and two device runs are bit-identical."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(20000, 600000, 11)
    nd = oracle_lib.site_norm(sm.indptr, sm.indices, sm.counts.astype(np.float64), sm.sites)
    rc, xo, st = oracle_lib.kr_balance(sm.indptr, sm.indices, nd)
    mat = ctx.upload(sm.indptr, sm.indices, nd)
    _, x1, info = ctx.kr_balance(mat)
    _, x2, _ = ctx.kr_balance(mat)
    assert info.n_iter == st.n_iter
    np.testing.assert_allclose(x1, xo, rtol=1e-9)
    assert np.array_equal(x1, x2)
    # size-independent property: X A' X is doubly stochastic (A' = A with unit diagonal where A's is zero)
    a = sp.csr_matrix((nd, sm.indices, sm.indptr), shape=(sm.n, sm.n))
    d = (a.diagonal() == 0).astype(np.float64)
    rows = x1 * (a @ x1) + d * x1 * x1
    assert np.abs(rows - 1.0).max() < 1e-5


def test_kr_locality_reordered_path_matches_oracle(ctx, oracle_lib, monkeypatch):
    """Large enough for the internal locality ordering (reorder.hip) to kick in: x must not depend on it."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(40000, 1300000, 21)
    assert sm.nnz >= (1 << 21)
    nd = oracle_lib.site_norm(sm.indptr, sm.indices, sm.counts.astype(np.float64), sm.sites)
    rc, xo, st = oracle_lib.kr_balance(sm.indptr, sm.indices, nd)
    mat = ctx.upload(sm.indptr, sm.indices, nd)
    _, x1, info = ctx.kr_balance(mat)
    assert info.n_iter == st.n_iter and info.n_zero_diag == st.n_zero_diag
    np.testing.assert_allclose(x1, xo, rtol=1e-9)
    monkeypatch.setenv('B3C_NO_REORDER', '1')
    _, x2, info2 = ctx.kr_balance(mat)
    assert info2.n_iter == info.n_iter
    np.testing.assert_allclose(x2, x1, rtol=1e-11)
    # count form of the same map through the three SpMV forms: dictionary tiles (default at this size), dictionary tiles with
    # the tails summed in fixed point by their own pass (automatic beyond 2^19 contigs, forced here), CSR row groups
    cm = ctx.upload(sm.indptr, sm.indices, sm.counts)
    monkeypatch.delenv('B3C_NO_REORDER')
    _, xc, ic = ctx.kr_balance_counts(cm, sm.sites)
    assert ic.n_iter == st.n_iter and ic.tcd_tiles > 0
    np.testing.assert_allclose(xc, xo, rtol=1e-9)
    monkeypatch.setenv('B3C_KR_TCD_XT', '1')
    _, xt, it = ctx.kr_balance_counts(cm, sm.sites)
    assert it.n_iter == st.n_iter and it.tcd_tiles > 0
    np.testing.assert_allclose(xt, xc, rtol=1e-11)
    _, xt2, _ = ctx.kr_balance_counts(cm, sm.sites)
    assert np.array_equal(xt, xt2)                          # integer accumulation: bit-identical repeats
    monkeypatch.delenv('B3C_KR_TCD_XT')
    monkeypatch.setenv('B3C_KR_NO_TCD', '1')
    _, xv, iv = ctx.kr_balance_counts(cm, sm.sites)
    assert iv.n_iter == st.n_iter and iv.tcd_tiles == 0
    np.testing.assert_allclose(xv, xc, rtol=1e-11)


@pytest.mark.parametrize('scale', [1, 100])
def test_dictionary_words_of_24_bits(ctx, oracle_lib, monkeypatch, scale):
    """With the tails in their own pass the dictionary kernel streams 24-bit words (13-bit dictionary index, 11-bit count;
    kr_tcd24_pack): same products in the same order, so x is BIT FOR BIT that of the 32-bit words (B3C_KR_TCD_P24=0).  scale = 100
    pushes counts beyond 2047: those entries are handed to the tail pass by the dictionary build (kr_tcd_build, `cfront`), so the
    24-bit words stay in use; without that rule (B3C_KR_TCD_DEMOTE=0) the pack kernel reports the overflow and the 32-bit words stay
    -- same x up to the order of the sums."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(40000, 1300000, 41, genome_law='sqrt')      # (the variant law: its dictionaries hold most entries and its counts stay below 656)
    cnt = (sm.counts.astype(np.uint64) * scale).astype(np.uint32)
    assert cnt.max() < 65536 and (cnt.max() > 2047) == (scale > 1)
    nd = oracle_lib.site_norm(sm.indptr, sm.indices, cnt.astype(np.float64), sm.sites)
    rc, xo, st = oracle_lib.kr_balance(sm.indptr, sm.indices, nd)
    assert rc == oracle_lib.OK
    cm = ctx.upload(sm.indptr, sm.indices, cnt)
    monkeypatch.setenv('B3C_KR_TCD_XT', '1')
    _, x24, i24 = ctx.kr_balance_counts(cm, sm.sites)
    assert i24.tcd_xt_tails > 0 and i24.n_iter == st.n_iter
    np.testing.assert_allclose(x24, xo, rtol=1e-9)
    monkeypatch.setenv('B3C_KR_TCD_P24', '0')
    _, x32, i32 = ctx.kr_balance_counts(cm, sm.sites)
    assert i32.n_iter == i24.n_iter and np.array_equal(x24, x32)
    monkeypatch.delenv('B3C_KR_TCD_P24')
    monkeypatch.setenv('B3C_KR_TCD_DEMOTE', '0')
    _, xk, ik = ctx.kr_balance_counts(cm, sm.sites)
    assert ik.n_iter == i24.n_iter
    np.testing.assert_allclose(xk, x24, rtol=1e-12)
    n_big = int((cnt >= 2048).sum())
    assert ik.tcd_xt_tails <= i24.tcd_xt_tails <= ik.tcd_xt_tails + n_big
    if scale > 1:
        assert i24.tcd_xt_tails > ik.tcd_xt_tails          # (entries with large counts did move)
    else:
        assert np.array_equal(xk, x24)
    cm.free()


def test_dictionaries_not_worth_their_staging_are_dropped(ctx, monkeypatch):
    """kr_tcd_build leaves a pass of a tile whose dictionary columns are used fewer than B3C_KR_TCD_MINUSE8 / 8 times on average to
    the tail pass (round 4; default 3.5).  Forced for every tile (a huge threshold) and switched off (0): the same balance up to the
    order of the sums, more tails with the rule than without, bit-identical repeats."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(40000, 1300000, 43)
    cm = ctx.upload(sm.indptr, sm.indices, sm.counts)
    monkeypatch.setenv('B3C_KR_TCD_XT', '1')
    res = {}
    for thr in ('0', '28', '1000000'):
        monkeypatch.setenv('B3C_KR_TCD_MINUSE8', thr)
        _, x, info = ctx.kr_balance_counts(cm, sm.sites)
        _, x2, _ = ctx.kr_balance_counts(cm, sm.sites)
        assert np.array_equal(x, x2)
        res[thr] = (x, info)
    x0, i0 = res['0']
    for thr in ('28', '1000000'):
        x, info = res[thr]
        assert info.n_iter == i0.n_iter
        np.testing.assert_allclose(x, x0, rtol=1e-11)
        assert info.tcd_xt_tails >= i0.tcd_xt_tails
    assert res['1000000'][1].tcd_xt_tails > i0.tcd_xt_tails
    assert res['1000000'][1].tcd_dict_total == 0          # no dictionary column anywhere
    cm.free()


def test_thin_dictionary_share_goes_to_the_tail_pass_altogether(ctx, monkeypatch):
    """When the dictionaries that survive the build-time rule hold under a fifth of the entries, every entry becomes a tail word (cut
    straight from the CSR arrays) and all tiles get their epilogue from kr_kb_epi (round 5; B3C_KR_TCD_ALLTAILS=0 keeps the mixed
    layout).  Same iteration count, x to 1e-11 (the tail sums are exact integers, the dictionary sums fp64), bit-identical repeats."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(40000, 1300000, 47, genome_law='survey')
    cm = ctx.upload(sm.indptr, sm.indices, sm.counts)
    monkeypatch.setenv('B3C_KR_TCD_XT', '1')
    monkeypatch.setenv('B3C_KR_TCD_MINUSE8', '96')          # 12 uses per column: only the densest tiles keep a dictionary
    res = {}
    for sw in ('0', '1'):
        monkeypatch.setenv('B3C_KR_TCD_ALLTAILS', sw)
        _, x, info = ctx.kr_balance_counts(cm, sm.sites)
        _, x2, _ = ctx.kr_balance_counts(cm, sm.sites)
        assert np.array_equal(x, x2)
        res[sw] = (x, info)
    (x0, i0), (x1, i1) = res['0'], res['1']
    assert 0 < i0.tcd_dict_total and (cm.nnz - i0.tcd_tail_entries) * 5 < cm.nnz, (i0.tcd_dict_total, i0.tcd_tail_entries, cm.nnz)   # the rule's case
    assert i1.tcd_dict_total == 0 and i1.tcd_xt_tails > i0.tcd_xt_tails and i1.tcd_xt_tails == i1.tcd_tail_entries
    assert i1.n_iter == i0.n_iter
    np.testing.assert_allclose(x1, x0, rtol=1e-11)
    cm.free()


def test_long_hub_rows_run_as_pieces(ctx, oracle_lib, monkeypatch):
    """Round 6: a hub row beyond HUB_SPLIT entries (16384; BASELINE's C5 has rows of 10^6) is cut into pieces, one workgroup each, whose
    sums kr_kb_hub_fin adds in piece order before the row's epilogue.  Forced here on a power-law map of 30,000 contigs (longest row
    ~6,000 entries) with pieces of 1024 and 256 entries, on the CSR row kernels and beside the dictionary / tail-pass forms: same
    iteration count, x to 1e-11 of the one-workgroup-per-row form and to 1e-9 of the oracle, bit-identical repeats."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(30000, 900000, 9, 'powerlaw')
    assert int(np.diff(sm.indptr).max()) > 2048
    nd = oracle_lib.site_norm(sm.indptr, sm.indices, sm.counts.astype(np.float64), sm.sites)
    rc, xo, st = oracle_lib.kr_balance(sm.indptr, sm.indices, nd)
    cm = ctx.upload(sm.indptr, sm.indices, sm.counts)
    fm = ctx.upload(sm.indptr, sm.indices, nd)
    for env in ({}, {'B3C_KR_NO_TCD': '1'}, {'B3C_KR_TCD_XT': '1'}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        _, x0, i0 = ctx.kr_balance_counts(cm, sm.sites)
        assert i0.n_iter == st.n_iter
        np.testing.assert_allclose(x0, xo, rtol=1e-9)
        for split in ('1024', '256'):
            monkeypatch.setenv('B3C_KR_HUB_SPLIT', split)
            _, x1, i1 = ctx.kr_balance_counts(cm, sm.sites)
            _, x2, _ = ctx.kr_balance_counts(cm, sm.sites)
            assert i1.n_iter == st.n_iter and np.array_equal(x1, x2)
            np.testing.assert_allclose(x1, x0, rtol=1e-11)
            monkeypatch.delenv('B3C_KR_HUB_SPLIT')
        for k in env:
            monkeypatch.delenv(k)
    # the fp64 matrix form (kr_kb_hub<double>)
    _, xf0, if0 = ctx.kr_balance(fm)
    monkeypatch.setenv('B3C_KR_HUB_SPLIT', '512')
    _, xf1, if1 = ctx.kr_balance(fm)
    assert if0.n_iter == if1.n_iter == st.n_iter
    np.testing.assert_allclose(xf1, xf0, rtol=1e-11)
    np.testing.assert_allclose(xf1, xo, rtol=1e-9)
    cm.free()
    fm.free()


def test_stream_policy_is_a_cache_hint_only(ctx, monkeypatch):
    """Round 6: a map whose SpMV streams more than the caches hold loads its tail words and hub rows non-temporally (KR_STREAM_NT_MIN).
    Forced on a 30,000-contig power-law map (B3C_KR_NT_MIN=0) beside the default policy, on the tail-pass form with hub rows whole and
    in pieces: the same iteration count and x BIT FOR BIT -- the bit changes where a line is kept, not what is read."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(30000, 900000, 9, 'powerlaw')
    cm = ctx.upload(sm.indptr, sm.indices, sm.counts)
    for env in ({'B3C_KR_TCD_XT': '1'}, {'B3C_KR_TCD_XT': '1', 'B3C_KR_HUB_SPLIT': '512'}, {'B3C_KR_NO_TCD': '1', 'B3C_KR_HUB_SPLIT': '512'}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        _, x0, i0 = ctx.kr_balance_counts(cm, sm.sites)
        monkeypatch.setenv('B3C_KR_NT_MIN', '0')
        _, x1, i1 = ctx.kr_balance_counts(cm, sm.sites)
        monkeypatch.delenv('B3C_KR_NT_MIN')
        assert i0.n_iter == i1.n_iter and np.array_equal(x0, x1)
        for k in env:
            monkeypatch.delenv(k)
    cm.free()


def test_all_tails_rule_decided_on_a_sample_of_the_tiles(ctx, oracle_lib, monkeypatch):
    """From 256 tiles on, the all-tails rule is first applied to every 16th tile and the full dictionary pass is skipped when the
    sample already says "no dictionary kernel" (round 5; the dictionary arrays of the other tiles are then never written).  A
    contract-law map of 60,000 contigs (312 tiles) with the tail pass forced: every entry a tail word, the transposed emission of a
    bitwise symmetric map, x equal to the mixed layout's to 1e-11 and to the oracle's to 1e-9, same iteration count."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(60000, 2500000, 7)
    nd = oracle_lib.site_norm(sm.indptr, sm.indices, sm.counts.astype(np.float64), sm.sites)
    rc, xo, st = oracle_lib.kr_balance(sm.indptr, sm.indices, nd)
    cm = ctx.upload(sm.indptr, sm.indices, sm.counts)
    monkeypatch.setenv('B3C_KR_TCD_XT', '1')
    monkeypatch.setenv('B3C_KR_TCD_MINUSE8', '96')           # 12 uses per column: only the densest tiles would keep a dictionary
    monkeypatch.setenv('B3C_KR_TCD_ALLTAILS', '0')
    _, x0, i0 = ctx.kr_balance_counts(cm, sm.sites)
    monkeypatch.setenv('B3C_KR_TCD_ALLTAILS', '1')
    _, x1, i1 = ctx.kr_balance_counts(cm, sm.sites)
    _, x2, _ = ctx.kr_balance_counts(cm, sm.sites)
    assert i1.tcd_tiles >= 256 and i1.tcd_dict_total == 0 and i1.tcd_xt_tails == i1.tcd_tail_entries
    assert 0 < i0.tcd_dict_total and (cm.nnz - i0.tcd_tail_entries) * 5 < cm.nnz, (i0.tcd_dict_total, i0.tcd_tail_entries, cm.nnz)
    assert i0.n_iter == i1.n_iter == st.n_iter and np.array_equal(x1, x2)
    np.testing.assert_allclose(x1, x0, rtol=1e-11)
    np.testing.assert_allclose(x1, xo, rtol=1e-9)
    cm.free()


def test_tail_pass_with_long_tails(ctx, monkeypatch):
    """Tail pass of the dictionary SpMV on a power-law map whose hub rows stay in the tiles (B3C_KR_NO_HUB): tails of
    thousands of entries narrow the 32-bit remainder sums of the fixed-point accumulation; x must agree with the CSR
    row kernels and repeat bit for bit."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(30000, 900000, 9, 'powerlaw')
    cm = ctx.upload(sm.indptr, sm.indices, sm.counts)
    assert int(np.diff(sm.indptr).max()) > 2048
    monkeypatch.setenv('B3C_KR_NO_TCD', '1')
    _, xv, iv = ctx.kr_balance_counts(cm, sm.sites)
    monkeypatch.delenv('B3C_KR_NO_TCD')
    monkeypatch.setenv('B3C_KR_TCD_XT', '1')
    for hub in (False, True):
        if not hub:
            monkeypatch.setenv('B3C_KR_NO_HUB', '1')
        else:
            monkeypatch.delenv('B3C_KR_NO_HUB')
        _, xt, it = ctx.kr_balance_counts(cm, sm.sites)
        assert it.tcd_xt_tails > 0 and it.n_iter == iv.n_iter
        np.testing.assert_allclose(xt, xv, rtol=1e-10)
        _, xt2, _ = ctx.kr_balance_counts(cm, sm.sites)
        assert np.array_equal(xt, xt2)


@pytest.mark.parametrize('n,e,env', [(6000, 150000, {}), (6000, 150000, {'B3C_KR_TCD_XT': '1'}), (6000, 150000, {'B3C_KR_NO_TCD': '1'}),
                                     (40000, 1300000, {}), (3000, 20000, {})])
def test_a_few_counts_beyond_16_bits_keep_the_16_bit_forms(ctx, oracle_lib, monkeypatch, n, e, env):
    """Hub-to-hub contacts of a power-law map exceed 65535; the map must still run on the 16-bit SpMV forms (dictionary tiles,
    tail pass, 16-bit CSR values) with the high halves of those few counts in a correction list, and agree with the oracle."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(n, e, 31)
    counts = sm.counts.copy()
    rows = np.repeat(np.arange(sm.n), np.diff(sm.indptr))
    rs = np.random.RandomState(3)
    up = np.flatnonzero(rows < sm.indices)
    pick = rs.choice(up, 40, replace=False)
    big = rs.choice([70000, 65536, 200000, 5000000, 4000000000], pick.shape[0]).astype(np.uint32)
    a = sp.csr_matrix((counts, sm.indices, sm.indptr), shape=(sm.n, sm.n)).tolil()
    for k, v in zip(pick.tolist(), big.tolist()):
        i, j = int(rows[k]), int(sm.indices[k])
        a[i, j] = v
        a[j, i] = v
    a = a.tocsr()
    a.sort_indices()
    ip, ix, cnt = a.indptr.astype(np.int64), a.indices.astype(np.int32), a.data.astype(np.uint32)
    nd = oracle_lib.site_norm(ip, ix, cnt.astype(np.float64), sm.sites)
    rc, xo, st = oracle_lib.kr_balance(ip, ix, nd)
    assert rc == oracle_lib.OK
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cm = ctx.upload(ip, ix, cnt)
    _, x, info = ctx.kr_balance_counts(cm, sm.sites)
    assert info.n_iter == st.n_iter
    assert info.value_bits == 16 and info.big_counts == 80, (info.value_bits, info.big_counts)
    np.testing.assert_allclose(x, xo, rtol=1e-9)
    _, x2, _ = ctx.kr_balance_counts(cm, sm.sites)
    assert np.array_equal(x, x2)
    cm.free()


@pytest.mark.parametrize('env', [{'B3C_KR_XT_S': '12288', 'B3C_KR_XT_CHUNKS': '16'}, {'B3C_KR_XT_S': '12288', 'B3C_KR_XT_CHUNKS': '8'},
                                 {'B3C_KR_XT_CHUNKS': '24'}, {'B3C_KR_XT_CHUNKS': '11'}])
def test_tail_pass_layouts_of_far_maps(ctx, oracle_lib, monkeypatch, env):
    """Maps whose operand vector is beyond the L2s (more than 1M contigs; C5) run the tail pass on 12288-row super-tiles with the
    column range cut into phases of eight slices (kr.hip, kr_kb_tails).  Forced here on a 40k-contig map: same iteration count
    as the oracle, x within 1e-9, and -- the row sums being exact fixed-point integers -- bit for bit the default layout's x."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(40000, 1300000, 37)
    nd = oracle_lib.site_norm(sm.indptr, sm.indices, sm.counts.astype(np.float64), sm.sites)
    rc, xo, st = oracle_lib.kr_balance(sm.indptr, sm.indices, nd)
    assert rc == oracle_lib.OK
    cm = ctx.upload(sm.indptr, sm.indices, sm.counts)
    monkeypatch.setenv('B3C_KR_TCD_XT', '1')
    _, x0, i0 = ctx.kr_balance_counts(cm, sm.sites)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _, x, info = ctx.kr_balance_counts(cm, sm.sites)
    assert info.tcd_xt_tails > 0 and info.tcd_xt_tails == i0.tcd_xt_tails
    assert info.n_iter == st.n_iter
    np.testing.assert_allclose(x, xo, rtol=1e-9)
    assert np.array_equal(x, x0)
    cm.free()


def test_symmetry_guard_fast_and_exact_paths(ctx):
    """Checksum path (bitwise symmetric), exact path (asymmetric beyond tol, and within tol but not bitwise)."""
    g = np.load(os.path.join(GOLDEN, 'kr_random1000.npz'))
    n = g['in_indptr'].shape[0] - 1
    _, _, info = ctx.kr_balance(ctx.upload(g['in_indptr'], g['in_indices'], g['in_data']))
    assert info.is_symmetric == 1 and info.bitwise_symmetric == 1
    d = g['in_data'].copy()
    rows = np.repeat(np.arange(n), np.diff(g['in_indptr']))
    k = np.flatnonzero(rows < g['in_indices'])[5]
    d[k] *= (1 + 1e-12)                                     # not bitwise symmetric, but |a_ij - a_ji| < tol
    _, _, info = ctx.kr_balance(ctx.upload(g['in_indptr'], g['in_indices'], d))
    assert info.is_symmetric == 1 and info.bitwise_symmetric == 0
    d[k] += 0.5                                             # beyond tol
    _, _, info = ctx.kr_balance(ctx.upload(g['in_indptr'], g['in_indices'], d))
    assert info.is_symmetric == 0 and info.bitwise_symmetric == 0


def test_graph_form_scaling_matches_reference_graph_weights(ctx):
    """b3c_scale_sym_graph: lower triangle == scipy's X A X bit for bit, upper == its mirror."""
    g = np.load(os.path.join(GOLDEN, 'kr_planted5000.npz'))
    n = g['in_indptr'].shape[0] - 1
    mat = ctx.upload(g['in_indptr'], g['in_indices'], g['in_data'])
    ctx.scale_sym_graph(mat, g['x'])
    _, _, data = mat.download(structure=False)
    rows = np.repeat(np.arange(n), np.diff(g['in_indptr']))
    low = rows >= g['in_indices']
    assert np.array_equal(data[low], g['out_data'][low])
    m = sp.csr_matrix((data, g['in_indices'], g['in_indptr']), shape=(n, n))
    assert (m != m.T).nnz == 0
    np.testing.assert_allclose(data, g["out_data"], rtol=5e-16)      # two roundings each side


def test_kr_on_permuted_layouts(ctx):
    """Canonical and unsorted (B3C_PERMUTE_UNSORTED) permutations: same x after mapping back, same warn counters."""
    g = np.load(os.path.join(GOLDEN, 'kr_zero_diag.npz'))
    n = g['in_indptr'].shape[0] - 1
    mat = ctx.upload(g['in_indptr'], g['in_indices'], g['in_data'])
    perm = np.random.RandomState(3).permutation(n).astype(np.int32)
    for unsorted in (False, True):
        pm = ctx.permute(mat, perm, unsorted=unsorted)
        ip, ix, _ = pm.download()
        sorted_rows = all(np.all(np.diff(ix[ip[r]:ip[r + 1]]) > 0) for r in range(n))
        assert sorted_rows == (not unsorted)
        _, xp, info = ctx.kr_balance(pm)
        assert info.n_zero_diag == 40 and info.n_iter == int(g['n_iter']) and info.bitwise_symmetric == 1
        x = np.empty(n)
        x[perm] = xp
        np.testing.assert_allclose(x, g['x'], rtol=1e-9)
        pm.free()
    # asymmetric beyond tol on an unsorted layout -> exact (scanning) guard path
    d = g['in_data'].copy()
    rows = np.repeat(np.arange(n), np.diff(g['in_indptr']))
    k = np.flatnonzero(rows < g['in_indices'])[7]
    d[k] += 0.5
    pm = ctx.permute(ctx.upload(g['in_indptr'], g['in_indices'], d), perm, unsorted=True)
    _, _, info = ctx.kr_balance(pm, check=False)
    assert info.is_symmetric == 0


@pytest.mark.parametrize('n,e,seed', [(1500, 40000, 23), (40000, 1300000, 21)])
def test_kr_on_raw_counts_equals_kr_on_normalised_matrix(ctx, oracle_lib, n, e, seed):
    """b3c_kr_balance_counts (8 B/entry, 1/sites folded in) against the oracle on the explicitly normalised matrix,
    and b3c_normalise_scale against site_norm + scale_sym bit for bit."""
    from bin3c_amd.synth import make_synth
    sm = make_synth(n, e, seed)
    sm.sites[::97] = 0.0                                               # zero sites count as one (contact_map.py:1107)
    nd = oracle_lib.site_norm(sm.indptr, sm.indices, sm.counts.astype(np.float64), sm.sites)
    rc, xo, st = oracle_lib.kr_balance(sm.indptr, sm.indices, nd)
    counts = ctx.upload(sm.indptr, sm.indices, sm.counts)
    _, x, info = ctx.kr_balance_counts(counts, sm.sites)
    assert info.n_iter == st.n_iter and info.n_zero_diag == st.n_zero_diag and info.bitwise_symmetric == 1
    np.testing.assert_allclose(x, xo, rtol=1e-9)
    a = ctx.site_norm(counts, sm.sites)
    ctx.scale_sym(a, x)
    _, _, want = a.download(structure=False)
    got = ctx.normalise_scale(counts, sm.sites, x, graph_form=False)
    assert np.array_equal(got.download(structure=False)[2], want)
    a2 = ctx.site_norm(counts, sm.sites)
    ctx.scale_sym_graph(a2, x)
    got = ctx.normalise_scale(counts, sm.sites, x, graph_form=True)
    assert np.array_equal(got.download(structure=False)[2], a2.download(structure=False)[2])


def test_kr_4d_wrapper(ctx):
    """kr_biostochastic_4d / dotdot (sparse_utils.py:480-508) on a synthetic N x N x 2 x 2 tip tensor."""
    from bin3c_amd.mzd import sparse_utils
    g = np.load(os.path.join(GOLDEN, 'kr_random1000.npz'))
    n = g['in_indptr'].shape[0] - 1
    rows = np.repeat(np.arange(n), np.diff(g['in_indptr']))
    rs = np.random.RandomState(4)
    # split every 2-D entry into its 2x2 block so that the block sums reproduce the fixture matrix
    parts = rs.dirichlet(np.ones(4), size=rows.shape[0])
    coords = np.vstack([np.repeat(rows, 4), np.repeat(g['in_indices'], 4), np.tile([0, 0, 1, 1], rows.shape[0]),
                        np.tile([0, 1, 0, 1], rows.shape[0])])
    data = (parts * g['in_data'][:, None]).ravel()
    t = sparse_utils.Coo4D(coords, data, (n, n, 2, 2))
    out, scl = sparse_utils.kr_biostochastic_4d(t, ctx=ctx)
    np.testing.assert_allclose(scl, g['x'], rtol=1e-9)
    np.testing.assert_allclose(out.data, data * scl[coords[0]] * scl[coords[1]], rtol=1e-15)
    # block sums of the scaled tensor are the balanced matrix
    back = sp.coo_matrix((out.data, (coords[0], coords[1])), shape=(n, n)).tocsr()
    np.testing.assert_allclose(np.asarray(back.sum(axis=1)).ravel(), 1.0, atol=5e-6)


def test_ill_conditioned_map_count_form_and_default_path(ctx):
    """tests/golden/illcond_kr_counts.npz (tools/make_golden3.py): 64 contigs, 126 entries, 262 iterations, 20 bound cuts.  The
    reference's two environments agree on x to 8e-9; the count form of the device balance was recorded 3.2e-6 away, the explicit
    fp64 form 6e-9.  (1) Both device forms reproduce the iteration count and balance the map; (2) the DEFAULT product path --
    run_hot_path with default arguments, and ContactMap.prepare_seq_map over it -- balances a map of this size in the
    reference's own formulation and returns x inside the 1e-6 contract; the count form stays
    available (strict_kr=False) and, on maps too large for the default rule, strict_kr=True repeats the balance in fp64."""
    import scipy.sparse as sp
    from bin3c_amd.pipeline import run_hot_path
    z = np.load(os.path.join(GOLDEN, 'illcond_kr_counts.npz'))
    ip, ix, cnt, sites = z['ip'], z['ix'], z['counts'], z['sites']
    n = ip.shape[0] - 1
    counts = ctx.upload(ip, ix, cnt)
    rel = lambda u, v: float(np.max(np.abs(u - v) / np.abs(v)))
    s = np.where(sites == 0, 1.0, sites)
    rows = np.repeat(np.arange(n), np.diff(ip))
    a = sp.csr_matrix((cnt.astype(np.float64) * (1.0 / (s[rows] * s[ix])), ix, ip), shape=(n, n))
    dz = (a.diagonal() == 0).astype(np.float64)
    resid = lambda x: float(np.abs(x * (a @ x) + dz * x * x - 1.0).max())
    rc, x_c, info_c = ctx.kr_balance_counts(counts, sites, max_iter=400, check=False)
    assert rc == 0 and info_c.n_iter == int(z['n_iter_ref']) and resid(x_c) < 1e-6
    assert rel(x_c, z['x_ref_a']) < 1e-4                       # the count form: valid, but beyond the contract on THIS map
    fm = ctx.upload(ip, ix, a.data)
    rc, x_f, info_f = ctx.kr_balance(fm, max_iter=400, check=False)
    fm.free()
    assert rc == 0 and info_f.n_iter == int(z['n_iter_ref'])
    assert rel(x_f, z['x_ref_a']) < 1e-6 and rel(x_f, z['x_ref_b']) < 1e-6
    lengths = np.full(n, 5000)
    loose = run_hot_path(ctx, counts, sites, lengths, 1000, 0, cluster=False, max_iter=400, strict_kr=False)
    default = run_hot_path(ctx, counts, sites, lengths, 1000, 0, cluster=False, max_iter=400)
    strict = run_hot_path(ctx, counts, sites, lengths, 1000, 0, cluster=False, max_iter=400, strict_kr=True)
    assert not loose.kr_strict and default.kr_strict and strict.kr_strict
    assert np.array_equal(loose.x, x_c) and np.array_equal(default.x, x_f) and np.array_equal(strict.x, x_f)
    assert default.kr_info.n_iter == int(z['n_iter_ref']) and default.kr_info.bitwise_symmetric == 1
    assert rel(default.x, z['x_ref_a']) < 1e-6 and rel(default.x, z['x_ref_b']) < 1e-6
    counts.free()
    # ... and through the reference's own call surface: ContactMap.prepare_seq_map(norm=True, bisto=True) (contact_map.py:911-945)
    from bin3c_amd.mzd.contact_map import ContactMap
    cm = ContactMap.from_arrays(sp.csr_matrix((cnt, ix, ip), shape=(n, n)), lengths, sites, min_len=1000, min_sig=1)
    cm._ctx = ctx
    cm.primary_acceptance_mask = np.ones(n, dtype=bool)
    cm.prepare_seq_map(norm=True, bisto=True)
    assert rel(cm.bisto_scale, z['x_ref_a']) < 1e-6 and rel(cm.bisto_scale, z['x_ref_b']) < 1e-6


def test_reference_spread_maps_device_balances_them(ctx):
    """tests/golden/spread_kr.npz: maps on which the reference disagrees with itself beyond 1e-6 (tests/test_kr_spread.py).  The
    device must return OK, balance each map to the tolerance, and take a number of iterations inside the reference's own range
    (+- 10 %) -- x itself is not determined there."""
    import scipy.sparse as sp
    z = np.load(os.path.join(GOLDEN, 'spread_kr.npz'))
    k = 0
    while 'ip_%d' % k in z:
        ip, ix, cnt, sites = z['ip_%d' % k], z['ix_%d' % k], z['counts_%d' % k], z['sites_%d' % k]
        n = ip.shape[0] - 1
        s = np.where(sites == 0, 1.0, sites)
        rows = np.repeat(np.arange(n), np.diff(ip))
        a = sp.csr_matrix((cnt.astype(np.float64) * (1.0 / (s[rows] * s[ix])), ix, ip), shape=(n, n))
        dz = (a.diagonal() == 0).astype(np.float64)
        its = [int(z['n_iter_ref_a_%d' % k]), int(z['n_iter_ref_b_%d' % k]), int(z['n_iter_ref_perm_%d' % k])]
        counts = ctx.upload(ip, ix, cnt)
        fm = ctx.upload(ip, ix, a.data)
        for rc, x, info in (ctx.kr_balance_counts(counts, sites, max_iter=400, check=False), ctx.kr_balance(fm, max_iter=400, check=False)):
            assert rc == 0
            assert float(np.abs(x * (a @ x) + dz * x * x - 1.0).max()) < 1e-6
            assert 0.9 * min(its) <= info.n_iter <= 1.1 * max(its), (k, info.n_iter, its)
        counts.free()
        fm.free()
        k += 1
    assert k == 8
