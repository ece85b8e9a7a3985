"""
Worker for the world_size-2 tests (launched with torch.distributed.run, backend gloo).

  cpu : host logic of the sharded path on CPU tensors -- row ranges, in-place all-gather of every buffer type,
        rank-ordered reduction of the packed scalars (no GPU, no libb3c compute)
  gpu : the real sharded hot path, two ranks sharing cuda:0 (gloo moves the CUDA buffers), against the
        single-rank run of the same map
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def cpu_mode(rank, world):
    from bin3c_amd.dist import allgather_inplace, shard_range
    n = 1000
    r0, r1, chunk = shard_range(n, rank, world)
    assert chunk % 64 == 0 and chunk * world >= n
    ranges = [shard_range(n, r, world)[:2] for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    for dtype in (torch.float64, torch.int32, torch.int64):
        full = torch.full((chunk * world,), -1, dtype=dtype)
        full[rank * chunk:(rank + 1) * chunk] = torch.arange(rank * chunk, (rank + 1) * chunk).to(dtype)
        allgather_inplace(full, rank, world)
        assert torch.equal(full, torch.arange(chunk * world).to(dtype)), dtype
    # packed scalars: every rank reduces the gathered slots in rank order -> bit-identical on all ranks
    scal = torch.zeros(4 * world, dtype=torch.float64)
    scal[4 * rank:4 * rank + 4] = torch.tensor([0.1 * (rank + 1), 1.0 - rank, 3.0 + rank, 7.0 - rank], dtype=torch.float64)
    allgather_inplace(scal, rank, world)
    v = scal.view(world, 4)
    s = 0.0
    for r in range(world):
        s += float(v[r, 0])
    out = torch.tensor([s, float(v[:, 1].min()), float(v[:, 2].max())], dtype=torch.float64)
    ref = out.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(out, ref)
    # ragged row ranges balanced by stored entries: every rank computes the same cuts; the ragged in-place gather
    # moves each rank's slice to its place
    from bin3c_amd.dist import allgather_ragged, balanced_starts
    rs = np.random.RandomState(4)
    lens = rs.randint(1, 50, 5000).astype(np.int64)
    lens[:40] = 20000                                  # a few hub rows at the front: equal row counts would be 3x out of balance
    st = balanced_starts(lens, world)
    assert st[0] == 0 and st[-1] == lens.shape[0] and np.all(np.diff(st) >= 0) and np.all(st[1:-1] % 64 == 0)
    per = np.add.reduceat(lens, st[:-1])
    assert per.max() <= 1.10 * per.mean() + 64 * 20000 / world, (per, st)
    ref = torch.as_tensor(st.copy())
    dist.broadcast(ref, 0)
    assert torch.equal(torch.as_tensor(st), ref)
    for dtype in (torch.float64, torch.int32):
        full = torch.full((int(st[-1]),), -7, dtype=dtype)
        full[st[rank]:st[rank + 1]] = torch.arange(int(st[rank]), int(st[rank + 1])).to(dtype)
        allgather_ragged(full, st, rank)
        assert torch.equal(full, torch.arange(int(st[-1])).to(dtype)), dtype
    assert np.array_equal(balanced_starts(np.zeros(300, dtype=np.int64), world)[[0, -1]], [0, 300])
    print('rank %d cpu ok' % rank)


def gpu_mode(rank, world):
    import graphs
    from bin3c_amd.dist import DistSession
    from bin3c_amd.pipeline import make_context, run_hot_path
    from bin3c_amd.synth import make_synth
    ctx = make_context(0)
    total = 0
    n_block = 0
    # a flat map, and a power-law one whose hub rows go through kr_kb_hub and the Louvain hub lists on both ranks
    # (the third is too small for the locality ordering: it runs the sharded kernels on a map every rank holds whole)
    # ... and the workload as SURVEY 8(d) writes it (one genome holds 38 % of the contigs)
    maps = ((40000, 1300000, 21), (70000, 4000000, 9, 'powerlaw'), (30000, 900000, 9, 'powerlaw'),
            (60000, 3000000, 4, 'flat', 0.1, 5, None, 'survey'))
    maps = maps[:int(os.environ.get('B3C_DIST_WORKER_MAPS', len(maps)))]
    for k, args in enumerate(maps):
        # the second map also takes the tail pass of the KR SpMV (automatic only beyond 2^19 contigs) on both row blocks
        if k == 1:
            os.environ['B3C_KR_TCD_XT'] = '1'
        else:
            os.environ.pop('B3C_KR_TCD_XT', None)
        ctx.reload_env()
        sm = make_synth(*args)
        counts = ctx.upload(sm.indptr, sm.indices, sm.counts)
        single = run_hot_path(ctx, counts, sm.sites, sm.lengths, 1000, 5)            # before sharding is installed
        sess = DistSession(ctx, sm.n)
        if k == 1:
            # ... in the layout of maps beyond 1M contigs on the row blocks (12288-row super-tiles, column slices in phases of eight):
            # the row sums are exact integers, so the layout cannot show in x
            os.environ['B3C_KR_XT_S'] = '12288'
            os.environ['B3C_KR_XT_CHUNKS'] = '16'
            ctx.reload_env()
        try:
            res = run_hot_path(ctx, counts, sm.sites, sm.lengths, 1000, 5)
        finally:
            os.environ.pop('B3C_KR_XT_S', None)
            os.environ.pop('B3C_KR_XT_CHUNKS', None)
            ctx.reload_env()
        sess.check()
        # the map was SHARDED, not replicated: this rank formed and touched only its row block of the ordered map
        if sm.n >= 32768 and counts.nnz >= (1 << 21):
            assert res.row_block == (sess.r0, sess.r1) and res.work_nnz < 0.7 * counts.nnz, (res.row_block, res.work_nnz, counts.nnz)
            tot_nnz = torch.tensor([res.work_nnz], dtype=torch.int64)
            dist.all_reduce(tot_nnz)
            assert int(tot_nnz.item()) == counts.nnz
            # the row blocks are balanced by STORED ENTRIES (round 4), whatever the row lengths: hub rows of the
            # power-law maps, the giant genome of the surveyed law
            most = torch.tensor([res.work_nnz], dtype=torch.int64)
            dist.all_reduce(most, op=dist.ReduceOp.MAX)
            assert int(most.item()) <= 1.10 * counts.nnz / world, (int(most.item()), counts.nnz, world, sess.starts)
            assert sess.starts is not None and int(sess.starts[-1]) == sm.n
            n_block += 1
        # ONE exchange per KR tick (round 3: the rank partial of p.w | rk.rk travels with the slice of w | v, and the vector step
        # after it runs over all rows on every rank); ticks are enqueued in batches of 4 and the state is read one batch
        # behind, a bound cut costs a tick without SpMV; per Louvain sweep of level 0 one label exchange per sub-round and
        # one integer sum per modularity evaluation; a dozen one-off exchanges (ordering 3, checksums, zero diagonals,
        # acceptance signal, entry count, maximum, block bounds, degrees, aggregation)
        n_spmv = res.kr_info.n_iter + 1
        cuts = res.kr_info.hit_delta + res.kr_info.hit_Delta
        assert n_spmv <= sess.n_exchanges <= (n_spmv + cuts + 12) + 5 * (res.lv_info.n_sweeps + 1) + 14, (sess.n_exchanges, n_spmv, cuts, res.lv_info.n_sweeps)
        # the ordering computed from per-rank row ranges is the single-GPU ordering, bit for bit
        if res._n2o is not None:
            assert single._n2o is not None and torch.equal(res._n2o, single._n2o)
        assert res.kr_info.n_iter == single.kr_info.n_iter
        if k == 1:
            assert res.kr_info.tcd_xt_tails > 0 and single.kr_info.tcd_xt_tails > res.kr_info.tcd_xt_tails, (res.kr_info.tcd_xt_tails, single.kr_info.tcd_xt_tails)
        assert np.array_equal(res.mask, single.mask) and np.array_equal(res.signal, single.signal)
        np.testing.assert_allclose(res.x, single.x, rtol=1e-10)
        assert graphs.same_partition(res.labels, single.labels)
        assert res.lv_info.n_sweeps == single.lv_info.n_sweeps and res.lv_info.n_levels == single.lv_info.n_levels
        # all ranks hold identical results
        lab = torch.as_tensor(res.labels.astype(np.int64))
        ref = lab.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(lab, ref)
        xs = torch.as_tensor(res.x)
        xr = xs.clone()
        dist.broadcast(xr, 0)
        assert torch.equal(xs, xr)
        sess.close()
        again = run_hot_path(ctx, counts, sm.sites, sm.lengths, 1000, 5)
        assert np.array_equal(again.x, single.x) and np.array_equal(again.labels, single.labels)
        total += sess.n_exchanges
        counts.free()
    assert n_block >= min(2, len(maps))
    print('rank %d gpu ok (%d exchanges, %d maps sharded by row blocks)' % (rank, total, n_block))
    ctx.close()


def _agree(res, single, rank):
    import graphs
    assert res.kr_info.n_iter == single.kr_info.n_iter
    assert np.array_equal(res.mask, single.mask)
    np.testing.assert_allclose(res.x, single.x, rtol=1e-10)
    assert graphs.same_partition(res.labels, single.labels)
    assert res.lv_info.n_sweeps == single.lv_info.n_sweeps and res.lv_info.n_levels == single.lv_info.n_levels
    # (x of the sharded run differs from the single-GPU one in the last bits -- other reduction trees -- and so do a few
    # quantised weights: the integer modularity agrees to that, the partition and the schedule exactly)
    assert abs(res.lv_info.modularity - single.lv_info.modularity) < 1e-9
    lab = torch.as_tensor(res.labels.astype(np.int64))
    ref = lab.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(lab, ref)


def gpu_many_mode(rank, world):
    """Row blocks whose level 0 leaves MORE communities than the kc x kc bin table holds (4096): the aggregation goes through
    per-rank pair runs, the ragged gather and the second reduction (b3c_dist_gather_pairs).  A power-law map of thousands of
    small genomes (C5's shape: BASELINE.json configs[4] ends level 0 with 5,590+ communities), and the pruned neighbour scan of
    the later sweeps on row blocks (validity stamps)."""
    from bin3c_amd.dist import DistSession
    from bin3c_amd.pipeline import make_context, run_hot_path
    from bin3c_amd.synth import make_synth
    ctx = make_context(0)
    sm = make_synth(90000, 2400000, 31, 'powerlaw', genomes=6000)
    counts = ctx.upload(sm.indptr, sm.indices, sm.counts)
    single = run_hot_path(ctx, counts, sm.sites, sm.lengths, 1000, 5, max_iter=5000)
    assert single.lv_info.n_levels >= 1 and single.lv_info.agg_gathered_runs == 0
    assert single.lv_info.n_comm_level0 > 4096, single.lv_info.n_comm_level0          # beyond the kc x kc bin table
    sess = DistSession(ctx, sm.n)
    res = run_hot_path(ctx, counts, sm.sites, sm.lengths, 1000, 5, max_iter=5000)
    sess.check()
    assert res.row_block == (sess.r0, sess.r1) and res.work_nnz < 0.8 * counts.nnz
    _agree(res, single, rank)
    # the pair path ran on the row blocks: this rank received the runs of the communities it owns, level 0 left more communities than
    # the bins hold, and level 1 is a row block again (built by owner: round 6)
    assert res.lv_info.agg_gathered_runs > 2048, res.lv_info.agg_gathered_runs
    assert res.lv_info.levels_sharded >= 1
    sess.close()
    # the same path forced on a map with few communities (every level through sort + gather), pruning on and off
    sm2 = make_synth(40000, 1300000, 21, noise=0.4, intra_boost=1)
    c2 = ctx.upload(sm2.indptr, sm2.indices, sm2.counts)
    single2 = run_hot_path(ctx, c2, sm2.sites, sm2.lengths, 1000, 5)
    # ... and with the sweeps of EVERY level sharded over the ranks (round 6: automatic for induced graphs of 2^24 entries and more --
    # the contract workload's level 1 at C3 -- forced here on the small levels of a hard map)
    for env in ({'B3C_LV_AGG_SORT': '1'}, {'B3C_LV_NO_PRUNE': '1'}, {'B3C_LV_DIST_MIN_NNZ': '1'}, {'B3C_LV_DIST_MIN_NNZ': '1', 'B3C_LV_NO_PRUNE': '1'}, {}):
        os.environ.update(env)
        ctx.reload_env()
        sess = DistSession(ctx, sm2.n)
        res2 = run_hot_path(ctx, c2, sm2.sites, sm2.lengths, 1000, 5)
        sess.check()
        _agree(res2, single2, rank)
        if 'B3C_LV_AGG_SORT' in env:
            assert res2.lv_info.agg_gathered_runs > 0
        # (the pair path builds the next level BY OWNER: a row block again, sharded whatever its size; the bins leave it whole on
        # every rank, and a whole level is sharded from 2^24 entries on)
        sharded_all = 'B3C_LV_DIST_MIN_NNZ' in env or 'B3C_LV_AGG_SORT' in env
        want = res2.lv_info.n_levels if sharded_all else (1 if single2.lv_info.n_comm_level0 > 4096 else 0)
        assert res2.lv_info.levels_sharded == want, (env, res2.lv_info.levels_sharded, res2.lv_info.n_levels, single2.lv_info.n_comm_level0)
        assert single2.lv_info.n_levels >= 2
        sess.close()
        for k in env:
            os.environ.pop(k)
        ctx.reload_env()
    c2.free()
    counts.free()
    assert res.lv_info.n_comm_level0 == single.lv_info.n_comm_level0
    print('rank %d gpu-many ok (%d communities after level 0, runs gathered: %d)' % (rank, single.lv_info.n_comm_level0, res.lv_info.agg_gathered_runs))
    ctx.close()


def gpu_large_mode(rank, world):
    """the sharded path beyond a million contigs.  A C5-shaped map (power law, 6000 genomes) of 1.2M contigs
    / 4.8e7 edges, drawn on the device by every rank (same seed, same stack: the same map): row blocks balanced by stored entries,
    level 0 beyond the bin table on row blocks (pair runs gathered over the ranks), same iteration count, x, mask and partition as
    one rank, bitwise equal results on all ranks."""
    from bin3c_amd.dist import DistSession
    from bin3c_amd.pipeline import make_context, run_hot_path
    from bin3c_amd.synth_device import make_counts_device
    ctx = make_context(0)
    n, e = 1200000, 48000000
    counts, n_edges, lengths, sites, genome, info = make_counts_device(ctx, n, e, 11, 'powerlaw', genomes=6000)
    chk = torch.tensor([counts.nnz, n_edges], dtype=torch.int64)
    ref = chk.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(chk, ref), 'the ranks drew different maps'
    single = run_hot_path(ctx, counts, sites, lengths, 1000, 5, max_iter=20000)
    assert single.lv_info.n_comm_level0 > 4096
    sess = DistSession(ctx, n)
    res = run_hot_path(ctx, counts, sites, lengths, 1000, 5, max_iter=20000)
    sess.check()
    assert res.row_block == (sess.r0, sess.r1) and sess.starts is not None
    most = torch.tensor([res.work_nnz], dtype=torch.int64)
    dist.all_reduce(most, op=dist.ReduceOp.MAX)
    assert int(most.item()) <= 1.10 * counts.nnz / world, (int(most.item()), counts.nnz)
    _agree(res, single, rank)
    assert res.lv_info.agg_gathered_runs > 4096 and res.lv_info.n_comm_level0 == single.lv_info.n_comm_level0
    sess.close()
    nnz_all = counts.nnz
    counts.free()
    # What a rank HOLDS: a caller that hands the raw map over (adopt_counts=True, what `bin3C.py cluster`
    # does) keeps 1/world of the map per GPU once the row block exists.  Device bytes of the context with the result still alive
    # (keep=True: working map, balanced sub-map) and the pool trimmed -- one rank against two, on the same map drawn again.
    def held(sharded):
        c2, _e, l2, s2, _g, _i = make_counts_device(ctx, n, e, 11, 'powerlaw', genomes=6000)
        sess2 = DistSession(ctx, n) if sharded else None
        r2 = run_hot_path(ctx, c2, s2, l2, 1000, 5, max_iter=20000, adopt_counts=True, keep=True)
        if sess2 is not None:
            sess2.check()
        ctx.sync()
        ctx.trim()
        b = ctx.bytes_in_use()
        x = r2.x.copy()
        lab = r2.labels.copy()
        r2.free()
        if sess2 is not None:
            sess2.close()
        ctx.trim()
        return b, x, lab
    b1, x1, lab1 = held(False)
    b2, x2, lab2 = held(True)
    assert np.array_equal(x1, single.x) and np.array_equal(lab1, single.labels)
    np.testing.assert_allclose(x2, x1, rtol=1e-10)
    import graphs
    assert graphs.same_partition(lab2, lab1)
    assert b2 <= 0.6 * b1, (b2, b1)
    print('rank %d gpu-large ok (%d contigs, %d stored entries, %d communities after level 0; device bytes held with the result alive: %d MB on one rank, %d MB per rank on two)' % (
        rank, n, nnz_all, single.lv_info.n_comm_level0, b1 >> 20, b2 >> 20))
    ctx.close()


def gpu_c4_mode(rank, world):
    """BASELINE.json configs[3] -- C4: the 1M-contig / 100M-edge map of bench.py's headline (the contract generator, G = n/200 genomes
    ~ Zipf(1.5)) row-sharded -- through the sharded code with the ranks sharing one GPU: row blocks balanced by stored entries (one
    genome holds 38 % of the contigs), 162 SpMVs with one exchange each, three Louvain levels of which the first ends beyond the bin
    table (185k communities: pair runs gathered over the ranks), identical results to one rank and on all ranks.  The 8-GPU form
    (xGMI) is the driver's; what can differ there is the transport, not the arithmetic."""
    from bin3c_amd.dist import DistSession
    from bin3c_amd.pipeline import make_context, run_hot_path
    from bin3c_amd.synth import CONFIGS, make_edges
    ctx = make_context(0)
    n, e, seed, law = CONFIGS['C3']
    lo, hi, c, diag, lengths, sites, genome = make_edges(n, e, seed, law, genome_law='survey')
    counts, n_edges = ctx.coo_symmetrise(n, lo, hi, c, diag)
    del lo, hi, c
    chk = torch.tensor([counts.nnz, n_edges], dtype=torch.int64)
    ref = chk.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(chk, ref), 'the ranks built different maps'
    single = run_hot_path(ctx, counts, sites, lengths, 1000, 5)
    assert single.kr_info.n_iter > 100 and single.lv_info.n_levels >= 3 and single.lv_info.n_comm_level0 > 4096
    bytes_single = ctx.bytes_in_use()
    sess = DistSession(ctx, n)
    res = run_hot_path(ctx, counts, sites, lengths, 1000, 5)
    sess.check()
    assert res.row_block == (sess.r0, sess.r1) and sess.starts is not None
    most = torch.tensor([res.work_nnz], dtype=torch.int64)
    dist.all_reduce(most, op=dist.ReduceOp.MAX)
    assert int(most.item()) <= 1.10 * counts.nnz / world, (int(most.item()), counts.nnz)
    _agree(res, single, rank)
    assert res.lv_info.n_comm_level0 == single.lv_info.n_comm_level0 and res.lv_info.n_comm == single.lv_info.n_comm
    # level 1 of this map (185,559 nodes, 1.4e8 entries: as large as level 0) is sharded too; the small levels after it are not
    assert res.lv_info.levels_sharded == 1 and single.lv_info.levels_sharded == 0, res.lv_info.levels_sharded
    n_spmv = res.kr_info.n_iter + 1
    cuts = res.kr_info.hit_delta + res.kr_info.hit_Delta
    assert n_spmv <= sess.n_exchanges <= (n_spmv + cuts + 12) + 5 * (res.lv_info.n_sweeps + 1) + 14 + 8 * res.lv_info.n_levels, (sess.n_exchanges, n_spmv)
    sess.close()
    counts.free()
    print('rank %d gpu-c4 ok (%d contigs, %d stored entries, %d SpMVs, %d communities after level 0 -> %d, %d exchanges; pool %d MB single)' % (
        rank, n, counts.nnz, n_spmv, single.lv_info.n_comm_level0, single.lv_info.n_comm, sess._final, bytes_single >> 20))
    ctx.close()


def gpu_direct_mode(rank, world):
    """The direct transport (round 6): the all-gathers of a sharded run as peer writes into IPC-mapped windows (here: between processes
    sharing ONE GPU) instead of collectives.  The same bytes arrive in the same places, so every result must be BITWISE that of the
    callback transport: x, mask, signal, labels, iteration and sweep counts -- on a flat map, on the contract law (ragged row blocks,
    a level built by owner) and on a power-law map of thousands of genomes; and the exchanges are counted by kind."""
    from bin3c_amd.dist import DistSession
    from bin3c_amd.pipeline import make_context, run_hot_path
    from bin3c_amd.synth import make_synth
    ctx = make_context(0)
    maps = [make_synth(40000, 1300000, 21, genome_law='sqrt'), make_synth(60000, 3000000, 4),
            make_synth(90000, 2400000, 31, 'powerlaw', genomes=6000)]
    if os.environ.get('B3C_DIRECT_C4'):
        maps = []
    total_d = total_c = 0
    for sm in maps:
        counts = ctx.upload(sm.indptr, sm.indices, sm.counts)
        out = {}
        for tr in ('callback', 'direct'):
            os.environ['B3C_DIST_TRANSPORT'] = tr
            sess = DistSession(ctx, sm.n)
            assert sess.direct == (tr == 'direct'), sess.transport
            res = run_hot_path(ctx, counts, sm.sites, sm.lengths, 1000, 5, max_iter=5000)
            sess.check()
            nd, nall = sess.n_exchanges_direct, sess.n_exchanges
            sess.close()
            out[tr] = (res, nd, nall)
        os.environ.pop('B3C_DIST_TRANSPORT')
        (rc, nd_c, na_c), (rd, nd_d, na_d) = out['callback'], out['direct']
        assert nd_c == 0 and nd_d > 0.8 * na_d and na_d == na_c, (nd_c, nd_d, na_d, na_c)     # every all-gather went through the window
        assert rd.kr_info.n_iter == rc.kr_info.n_iter and np.array_equal(rd.x, rc.x)
        assert np.array_equal(rd.mask, rc.mask) and np.array_equal(rd.signal, rc.signal) and np.array_equal(rd.labels, rc.labels)
        assert (rd.lv_info.n_sweeps, rd.lv_info.n_levels, rd.lv_info.modularity) == (rc.lv_info.n_sweeps, rc.lv_info.n_levels, rc.lv_info.modularity)
        lab = torch.as_tensor(rd.labels.astype(np.int64))
        ref = lab.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(lab, ref)
        total_d += nd_d
        total_c += na_d - nd_d
        counts.free()
    if os.environ.get('B3C_DIRECT_C4'):
        # C4's workload (the C3 contract map) through the direct transport, against the single-rank run
        from bin3c_amd.synth import CONFIGS, make_edges
        n, e, seed, law = CONFIGS['C3']
        lo, hi, c, diag, lengths, sites, genome = make_edges(n, e, seed, law, genome_law='survey')
        counts, n_edges = ctx.coo_symmetrise(n, lo, hi, c, diag)
        del lo, hi, c
        single = run_hot_path(ctx, counts, sites, lengths, 1000, 5)
        os.environ['B3C_DIST_TRANSPORT'] = 'direct'
        sess = DistSession(ctx, n)
        res = run_hot_path(ctx, counts, sites, lengths, 1000, 5)
        sess.check()
        total_d, total_c = sess.n_exchanges_direct, sess.n_exchanges - sess.n_exchanges_direct
        sess.close()
        os.environ.pop('B3C_DIST_TRANSPORT')
        _agree(res, single, rank)
        assert res.lv_info.levels_sharded == 1 and total_d > res.kr_info.n_iter
        counts.free()
    print('rank %d gpu-direct ok (%d exchanges by peer writes, %d by collectives)' % (rank, total_d, total_c))
    ctx.close()


def gpu_empty_mode(rank, world):
    """One rank's row block is masked out completely (all its contigs shorter than min_len): that rank still has to take part
    in every collective of the cluster stage (maximum, degrees, label gathers, bins) with an empty sub-map."""
    from bin3c_amd.dist import DistSession, shard_range
    from bin3c_amd.pipeline import make_context, run_hot_path
    from bin3c_amd.synth import make_synth
    ctx = make_context(0)
    sm = make_synth(40000, 1300000, 23)
    counts = ctx.upload(sm.indptr, sm.indices, sm.counts)
    probe = run_hot_path(ctx, counts, sm.sites, sm.lengths, 1000, 5, cluster=False)
    n2o = probe._n2o.cpu().numpy()
    lengths = np.maximum(sm.lengths, 1000)
    r0, r1, _ = shard_range(sm.n, world - 1, world)
    lengths[n2o[r0:r1]] = 500                       # the whole block of the last rank falls under min_len
    single = run_hot_path(ctx, counts, sm.sites, lengths, 1000, 5)
    assert not single.mask[n2o[r0:r1]].any() and single.mask.sum() > 0
    sess = DistSession(ctx, sm.n)
    res = run_hot_path(ctx, counts, sm.sites, lengths, 1000, 5, keep=True, balance_rows=False)      # (equal row ranges: the block masked above)
    sess.check()
    if rank == world - 1:
        assert res.sub is not None and res.sub.nnz == 0, res.sub.nnz       # this rank's block of the sub-map is empty
    _agree(res, single, rank)
    res.free()
    # ... and nothing accepted anywhere: every rank takes the early return together
    none = run_hot_path(ctx, counts, sm.sites, np.full(sm.n, 500), 1000, 5)
    assert none.mask.sum() == 0 and none.labels.shape[0] == 0
    sess.close()
    counts.free()
    print('rank %d gpu-empty ok' % rank)
    ctx.close()


if __name__ == '__main__':
    dist.init_process_group(backend='gloo')
    print('rank %d rendezvous done' % dist.get_rank(), flush=True)
    r, w = dist.get_rank(), dist.get_world_size()
    {'cpu': cpu_mode, 'gpu': gpu_mode, 'gpu-many': gpu_many_mode, 'gpu-empty': gpu_empty_mode, 'gpu-large': gpu_large_mode,
     'gpu-c4': gpu_c4_mode, 'gpu-direct': gpu_direct_mode}[sys.argv[1]](r, w)
    dist.barrier()
    dist.destroy_process_group()
