#!/usr/bin/env python3
"""
bench.py -- contact-graph edges/s through KR-normalise + Louvain (BASELINE.json metric).

One "step" = one pass of the cluster hot path over one synthetic contact map that is already
resident in HBM: acceptance mask -> site normalisation -> Knight-Ruiz balancing -> X A X ->
mask compaction -> weight scaling -> Louvain labels on the host (bin3c_amd/pipeline.py).

The headline leg (`value`, `roofline`, `cpu_baseline`) runs the workload as SURVEY.md 8(d) defines it (hic-synth-v1,
G = n/200 genomes ~ Zipf(1.5)); the rank^-0.5 genome-size variant rounds 1-4 were tuned on follows as a labelled extra leg
("variant_workload").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C1|C2|C3|C5]

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU, RCCL).  `--mode`:
  sharded   contig-row blocks of ONE map over all GPUs (BASELINE.json configs[3]; strong scaling; bin3c_amd/dist.py:
            every rank forms and works on its row block of the ordered map; RCCL collectives issued inside libb3c, one per
            SpMV and one per Louvain sub-round)
  replicas  one map per GPU (weak scaling, no data-path collective)
  auto      (default) both, replicas first: `value` is the SHARDED figure ("scaling": "strong") and the replica
            throughput is reported beside it under "replicas".  The sharded leg runs under a watchdog: should the
            RCCL path fail or never return, the run FAILS (non-zero exit status on every rank); the line is still
            printed for the record with "value": null, the replica figure under "replicas" and the failure under
            "sharded".
Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import os
import sys
import time

os.environ.setdefault('B3C_PROFILE', '1')      # HIP events around the SpMV / delta-Q launches
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
L1_FILL_GATHERS_PER_S = 2.6e11   # random 4-byte gathers per second from an L2-resident array, chip-wide (profiles/r06_l1fill_probe.txt)


def measured_traffic(key, kernel, algorithmic=None, field='hbm_bytes_per_launch'):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes (profiles/traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate passes on this workload, corrected as MI355X_MICROARCH.md prescribes); None when no
    measurement is on file -- or when the record on file is for another map of the same name (its algorithmic byte count
    differs from this run's)."""
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        with open(path) as fh:
            rec = json.load(fh).get(key, {})
        base = lambda name: name.replace('<', ' ').split()[0] if name else ''
        if rec and base(rec.get('kernel', '')) == base(kernel):
            ref = rec.get('algorithmic_bytes_per_launch')
            if ref and algorithmic and abs(ref - algorithmic) > 0.02 * ref:
                return None
            return rec.get(field)
    except (IOError, ValueError):
        pass
    return None


def measured_copy_gbs():
    """Achievable HBM rate on this box: device-to-device copy of 1 GiB (read + write), best of 5, torch stream."""
    import torch
    a = torch.empty(1 << 28, dtype=torch.float32, device='cuda')
    b = torch.empty_like(a)
    a.fill_(1.0)
    best = None
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    del a, b
    return 2.0 * (1 << 30) / (best * 1e-3) / 1e9


def cpu_baseline(ip, ix, cnt, sites, lengths, e_true, what, threads=1):
    """Oracle (scalar C port of the reference path, kind "port") on the host cores, timed over the whole cluster step on
    the given map: mask, site normalisation, KR, X A X, compress, graph weights, sequential Louvain.  threads > 1: the
    row-parallel loops (SpMV, symmetry guard, site normalisation, scaling, max_offdiag, compress) run under OpenMP -- every
    row is still processed by one thread, so the results are bit for bit those of one thread; the python-louvain-style
    Louvain stays sequential, as its semantics are."""
    import oracle
    oracle.build()
    oracle.set_threads(threads)
    try:
        t0 = time.perf_counter()
        sig = oracle.max_offdiag_u32(ip, ix, cnt)
        mask = (lengths >= 1000) & (sig >= 5)
        nd = oracle.site_norm(ip, ix, cnt.astype(np.float64), sites)
        rc, x, st = oracle.kr_balance(ip, ix, nd)
        t_kr = time.perf_counter() - t0
        proc = oracle.scale_sym(ip, ix, nd, x)
        del nd
        m, sip, six, a = oracle.compress(ip, ix, proc, mask)
        del proc
        scl = 1.0 / oracle.max_value(sip, a)
        w = oracle.graph_weights(sip, six, a, scl)
        del a
        r = oracle.louvain_seq(sip, six, w)
        dt = time.perf_counter() - t0
    finally:
        oracle.set_threads(1)
    return {'value': e_true / dt, 'unit': 'edges/s', 'cores': threads, 'kind': 'port', 'seconds': dt,
            'kr_spmv': int(st.n_iter + 1), 'louvain_sweeps': int(r['n_sweeps']), 'communities': int(r['n_comm']),
            'sample': '{}: full mask+norm+KR({} SpMV)+compress+sequential Louvain ({} sweeps) in {:.1f}s (mask+norm+KR '
                      '{:.1f}s) on {} host thread{}'.format(what, st.n_iter + 1, r['n_sweeps'], dt, t_kr, threads,
                                                            's' if threads > 1 else '')}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='C3', choices=['C1', 'C2', 'C3', 'C5'],
                    help="C5: 5M contigs / 1e9 edge draws, power law (BASELINE.json's largest map; fits ONE GPU; no CPU baseline)")
    ap.add_argument('--max-iter', type=int, default=None,
                    help='KR iteration limit (the reference default 1000; C5 defaults to 20000 | 5000 by --c5-counts: its heavy-tailed degrees need 10,903 | 2,967 SpMVs)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline', default='same', choices=['same', 'sample'],
                    help="'same': the oracle on the very map the GPU ran (about two minutes at C3); 'sample': on a bounded "
                         "sample of the same generator (n = 200k, E = 10M)")
    ap.add_argument('--mode', default='auto', choices=['auto', 'sharded', 'replicas'], help='N > 1 only')
    ap.add_argument('--synth', default=None, choices=['survey', 'sqrt'],
                    help="genome layout of the synthetic map: 'survey' (default, the contract workload) = SURVEY.md 8(d) to the letter, "
                         "G = n/200 genomes ~ Zipf(1.5); 'sqrt' = the variant rounds 1-4 were tuned on, G = n/500 genomes ~ rank^-0.5 "
                         "(runs as a labelled extra leg, \"variant_workload\", after the headline).  '--config C5 --c5-counts unique' "
                         "defaults to 'sqrt' (the map of rounds 3-4): on the survey's law the device draw stream ran out of GPU memory before it "
                         "reached 1e9 UNIQUE edges (round 5; most genomes there hold under 10 contigs and saturate)")
    ap.add_argument('--c5-counts', default='summed-draws', choices=['unique', 'summed-draws'],
                    help="power-law maps: 'summed-draws' (default; SURVEY 8(d): duplicates summed) = the stated number of DRAWS with "
                         "duplicates summed, the host generator's rule (C5: 4.6e8 edges whose hub pairs carry counts of 10^5..10^6 -- about "
                         "2,450 KR iterations); 'unique' = the stated number of UNIQUE edges, one count per pair (an easier map for KR: "
                         "report it as an additional, labelled line only; on the VARIANT genome-size law unless --synth says otherwise)")
    ap.add_argument('--no-variant-leg', action='store_true',
                    help='skip the second leg on the rank^-0.5 genome-size variant (printed under "variant_workload")')
    ap.add_argument('--n-sub', type=int, default=0, help='sub-rounds per Louvain sweep (0: the library default, 4)')
    ap.add_argument('--hard', action='store_true',
                    help='a HARD clustering problem on the same contigs: 45 %% inter-genome noise, intra-genome counts not '
                         'boosted (several Louvain levels, a dozen sweeps) instead of the easy planted structure')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: torch.cuda.is_available() is False and libb3c has no CPU path')
    if os.environ.get('B3C_BENCH_ONE_GPU'):      # test hook: all ranks share cuda:0 (tests/test_dist.py, gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # test hook: a one-rank process group exercises the RCCL code path of the sharded leg on a single GPU
    one_rank_group = world == 1 and bool(os.environ.get('B3C_BENCH_FORCE_SHARDED')) and 'RANK' in os.environ
    if world > 1 or one_rank_group:
        backend = os.environ.get('B3C_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend=backend)

    from bin3c_amd.pipeline import make_context, run_hot_path
    from bin3c_amd.synth import CONFIGS, make_edges

    # C5: the reference's default of 1000 would raise on either map; the device-drawn one with duplicate draws summed (hub pairs
    # of 10^5..10^6 contacts beside tail pairs of 5) takes 10,903 SpMVs, the one with one count per unique pair 2,967
    max_iter = args.max_iter if args.max_iter else ((20000 if args.c5_counts == 'summed-draws' else 5000) if args.config == 'C5' else 1000)
    if args.config == 'C5':
        args.no_cpu_baseline = True          # the oracle would need hours on this map
    ctx = make_context(local_rank, pin=True)       # one process per GPU: stay on the CPUs of its NUMA node
    min_len, min_sig = 1000, 5

    class Workload(object):
        """one synthetic map resident in HBM: count matrix + the two per-contig tables, and what to call it"""

    def build_workload(config, synth, hard):
        # pair stream drawn on the host (numpy RandomState, deterministic), accumulated into the symmetric count matrix on
        # the device (b3c_coo_symmetrise: the ingest tail, bit-identical to synth.symmetrise_to_csr).  Every rank builds
        # the same map: the replica leg processes one copy per GPU, the sharded leg one copy over all GPUs.
        w = Workload()
        w.config = config
        w.n, n_edges, w.seed, w.law = CONFIGS[config]
        genome_law = 'survey' if synth == 'survey' else 'sqrt'
        noise, boost = (0.45, 1) if hard else (0.1, 5)
        synth_note = ''
        if w.law == 'powerlaw':
            # the host generator sizes its draws for uniform endpoints; under the power law they collapse onto hub pairs (1e9
            # draws -> 4.6e8 unique edges at C5).  'summed-draws' (default: SURVEY 8(d)'s "duplicates summed") keeps that map;
            # 'unique' draws on the device until the number of UNIQUE edges is the stated one, one count per pair.
            from bin3c_amd.synth_device import make_counts_device
            summed = args.c5_counts == 'summed-draws'
            w.counts, w.e_true, w.lengths_h, w.sites_h, _genome, sinfo = make_counts_device(
                ctx, w.n, n_edges, w.seed, w.law, noise, boost, genome_law=genome_law, dup='sum' if summed else 'first', draws_only=summed)
            synth_note = ', drawn on the device ({:.3g} intra-genome draws for {:.3g} unique intra-genome edges; {})'.format(
                sinfo['intra_draws'], sinfo['intra_unique'], 'duplicate draws summed' if summed else 'one count per unique pair')
        else:
            lo, hi, cnt, diag, w.lengths_h, w.sites_h, _genome = make_edges(w.n, n_edges, w.seed, w.law, noise, boost, genome_law=genome_law)
            w.counts, w.e_true = ctx.coo_symmetrise(w.n, lo, hi, cnt, diag)
            del lo, hi, cnt
        # which genome-size law: the headline is the workload as SURVEY 8(d) defines it; the law rounds 1-4 were tuned on
        # DEVIATES from it and runs beside it as a labelled extra leg, "variant_workload" (alone with --synth sqrt)
        variant = (' [genome sizes: G = n/500 ~ rank^-0.5 -- a VARIANT, not SURVEY 8(d)\'s law]'
                   if synth != 'survey' else ' [genome sizes as SURVEY 8(d): G = n/200 ~ Zipf(1.5)]') + (
                   ' [hard: 45 % noise, no intra boost]' if hard else '')
        w.synth, w.hard = synth, hard
        w.label = '{}: hic-synth-v1{} n={} contigs, E={} undirected edges (nnz={}), seed={}{}{}'.format(
            config, ' power law' if w.law == 'powerlaw' else '', w.n, w.e_true, w.counts.nnz, w.seed, variant, synth_note)
        # inputs resident in HBM before the timed region: the count matrix and the two per-contig tables
        w.sites = torch.as_tensor(w.sites_h, device='cuda')
        w.lengths = torch.as_tensor(w.lengths_h.astype(np.int64), device='cuda')
        w.nnz = w.counts.nnz
        return w

    if args.synth is None:
        args.synth = 'sqrt' if (args.config == 'C5' and args.c5_counts == 'unique') else 'survey'
    W = build_workload(args.config, args.synth, args.hard)

    def barrier():
        if world > 1 or one_rank_group:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    def timed_leg(w, steps, warmup, hand_over=False):
        """W warm-up steps, then exactly K timed steps between barriers; time = max over ranks.
        hand_over (the sharded leg): every step gets its OWN resident copy of the raw map and hands it over
        (run_hot_path(adopt_counts=True), what `bin3C.py cluster` does): the copy is freed as soon as the rank's row block of
        the ordered map exists, so from there on a rank holds 1/world of the map, as a production run does.  The copies are
        made before the timed region (inputs resident in HBM); when they would not fit beside the map, the leg keeps one."""
        acc = {'stage': {}, 'spmv_ms': 0.0, 'spmv_n': 0, 'move_ms': 0.0, 'move_n': 0, 'move_nnz': 0, 'move_rows': 0,
               'first_ms': 0.0, 'first_n': 0}
        last = None
        copies = []
        if hand_over:
            free_b, _tot = torch.cuda.mem_get_info()
            per = w.nnz * 8 + (w.n + 1) * 8
            if (steps + warmup) * per < 0.5 * free_b:
                copies = [ctx.clone(w.counts) for _ in range(steps + warmup)]
        acc['raw_map_handed_over'] = bool(copies)

        def step():
            if copies:
                return run_hot_path(ctx, copies.pop(), w.sites, w.lengths, min_len, min_sig, max_iter=max_iter, n_sub=args.n_sub, adopt_counts=True)
            return run_hot_path(ctx, w.counts, w.sites, w.lengths, min_len, min_sig, max_iter=max_iter, n_sub=args.n_sub)
        for _ in range(warmup):
            last = step()
        gc.collect()
        gc.disable()            # a generation-2 collection in the middle of a step shows up as a 20 ms hole in the GPU timeline
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
            for k, v in last.timings.items():
                acc['stage'][k] = acc['stage'].get(k, 0.0) + v
            acc['spmv_ms'] += last.kr_info.spmv_ms
            acc['spmv_n'] += last.kr_info.n_spmv
            acc['move_ms'] += last.lv_info.move_ms
            acc['move_n'] += last.lv_info.n_move_launch
            acc['move_nnz'] += last.lv_info.move_nnz
            acc['move_rows'] += last.lv_info.move_rows
            acc['first_ms'] += last.lv_info.first_sweep_ms
            acc['first_n'] += last.lv_info.first_sweep_launches
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            red_dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
            tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, last, acc

    def traffic_key(w):
        """name of this workload's records in profiles/traffic.json: '<config>' for the contract workload, '<config>_variant' for
        the rank^-0.5 genome-size variant; none for the hard map and the power-law draws"""
        if w.hard:
            return '-'
        if w.law == 'powerlaw':
            # BASELINE's largest map to the letter of the contract generator (survey genome sizes, duplicate draws summed): 'C5'
            return w.config if (w.synth == 'survey' and args.c5_counts == 'summed-draws') else '-'
        return w.config if w.synth == 'survey' else w.config + '_variant'

    def spmv_roofline(w, last, acc, share, sharded):
        """The KR SpMV (all launches of one product).  `achieved` / `frac` are PHYSICAL: the HBM bytes one SpMV moves -- by the
        counters when a pass over this kernel on this workload is on file (profiles/traffic.json), else the bytes the kernel's
        encoding streams by construction (b3c_kr_info.spmv_stream_bytes: the library prices its own layout -- 3 or 4 bytes per
        dictionary entry, 4 per packed tail word, 8 per residual word, the dictionaries, row tables and epilogue vectors) --
        over its mean duration, against the 8 TB/s peak.  SURVEY 8(d)'s algorithmic figure (12 B per stored entry + 20 B per
        row: what a CSR fp64 SpMV would have to move) rides beside it as `achieved_algorithmic_equiv` /
        `frac_algorithmic_equiv`: an equivalent rate, not a fraction of anything -- the encodings stream fewer bytes than that."""
        nnz, n = w.nnz, w.n
        algorithmic = (12.0 * nnz + 20.0 * n) * share     # one KR SpMV
        ki = last.kr_info
        streamed = float(ki.spmv_stream_bytes)             # this rank's rows, as the library laid them out
        tcd = getattr(ki, 'tcd_tiles', 0) > 0              # tile-dictionary layout (automatic from 4096 contigs on)
        xt = getattr(ki, 'tcd_xt_tails', 0) > 0            # ... with the tails summed by their own pass (beyond 2^19 contigs)
        nodict = xt and getattr(ki, 'tcd_dict_total', 0) == 0     # every entry a tail word (the library's rule for thin dictionary shares)
        kname = ('kr_kb_tails + kr_kb_epi (ONE SpMV: every stored entry a tail word, summed in column order in fixed point, then the fused '
                 'KR epilogue per row; times and bytes are those of all its launches)' if nodict
                 else 'kr_kb_tails + kr_kb_tcd2 + kr_kb_epi (ONE SpMV: tails summed in column order in fixed point, then dictionary '
                 'rows + fused KR epilogue; times and bytes are those of all its launches)' if xt
                 else 'kr_kb_tcd2 (tile-dictionary SpMV on raw counts + fused KR epilogue)' if tcd
                 else 'kr_kb_vec (CSR SpMV on raw counts + fused KR epilogue)')
        avg_ms = acc['spmv_ms'] / max(acc['spmv_n'], 1)
        traffic = measured_traffic(traffic_key(w), kname, algorithmic) if not sharded else None
        physical = traffic if traffic else streamed
        rate = lambda b: b / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        return {'bound': 'hbm', 'kernel': kname, 'achieved': rate(physical), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': rate(physical) / HBM_PEAK_GBS,
                'frac_basis': 'HBM bytes per launch by the rocprofv3 counters (profiles/traffic.json)' if traffic
                              else 'bytes the kernel streams by construction (no counter pass on file for this kernel and workload)',
                'traffic': traffic, 'bytes_per_launch': physical, 'bytes_streamed_per_launch': streamed, 'avg_launch_ms': avg_ms,
                'achieved_streamed': rate(streamed), 'frac_streamed': rate(streamed) / HBM_PEAK_GBS,
                'bytes_algorithmic_per_launch': algorithmic, 'achieved_algorithmic_equiv': rate(algorithmic),
                'frac_algorithmic_equiv': rate(algorithmic) / HBM_PEAK_GBS,
                'tail_share': float(ki.tcd_tail_entries) / max(nnz * share, 1.0) if tcd else None,
                'word_bytes': {'dictionary': int(ki.tcd_word_bytes), 'tail': int(ki.xt_word_bytes),
                               'residual_words': int(ki.xt_residual_words)} if tcd else None}

    def louvain_roofline(w, last, acc, share, n_acc, sharded):
        """The delta-Q neighbour scan; one launch = one sub-round = 1/n_sub of a sweep.  SURVEY 8(d): 16 B per stored entry + 28 B
        per node and sweep.  `achieved` / `frac` are those of the FIRST sweep of level 0, whose launches read every row (no
        node can be settled without its row yet); the mean over all launches of a step is flattered by the pruned sub-rounds,
        which skip most rows but are charged the full bytes, and rides beside it as `frac_mean_incl_pruned`.  `traffic`: the
        counters' HBM bytes of a first-sweep launch when a pass on this workload is on file."""
        move_n = max(acc['move_n'], 1)
        mean_bytes = (16.0 * acc['move_nnz'] + 28.0 * acc['move_rows']) / move_n
        mean_ms = acc['move_ms'] / move_n
        rate = lambda b, ms: b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        nsub = args.n_sub if args.n_sub > 0 else 4
        sub = last.sub_nnz if getattr(last, 'sub_nnz', None) else None
        first_bytes = (16.0 * (sub if sub else w.nnz * share) + 28.0 * n_acc * share) / nsub
        first_ms = acc['first_ms'] / max(acc['first_n'], 1)
        use_first = acc['first_n'] > 0 and first_ms > 0
        b, ms = (first_bytes, first_ms) if use_first else (mean_bytes, mean_ms)
        kname = 'k_lv_decide (Louvain delta-Q neighbour scan; one launch = one sub-round: 1/n_sub of a sweep)'
        traffic = measured_traffic(traffic_key(w) + '_k_lv_scan', 'k_lv_decide', None, 'first_sweep_hbm_bytes_per_launch') if (use_first and not sharded) else None
        return {'bound': 'hbm', 'kernel': kname,
                'achieved': rate(b, ms), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': rate(b, ms) / HBM_PEAK_GBS,
                'frac_basis': 'SURVEY 8(d) bytes of the unpruned first sweep of level 0 over its mean launch' if use_first
                              else 'SURVEY 8(d) bytes over the mean launch of the step',
                'traffic': traffic, 'frac_traffic': rate(traffic, ms) / HBM_PEAK_GBS if traffic else None,
                'bytes_per_launch': b, 'avg_launch_ms': ms,
                'frac_mean_incl_pruned': rate(mean_bytes, mean_ms) / HBM_PEAK_GBS, 'avg_launch_ms_all': mean_ms,
                'launches_per_step': acc['move_n'] / max(1, acc.get('steps', 1)),
                # what the scan is bound by: one 4-byte label gather per stored entry, nearly every one an L1 miss
                # that fills a 128-byte line.  Ceiling: tools/probes/l1fill_probe.hip on this chip, random 4-byte gathers from an
                # L2-resident array, the same under every cache policy (profiles/r06_l1fill_probe.txt)
                'label_gathers': {'per_s': ((sub if sub else w.nnz * share) / nsub) / (first_ms * 1e-3) if use_first else None,
                                  'ceiling_per_s': L1_FILL_GATHERS_PER_S,
                                  'frac': ((sub if sub else w.nnz * share) / nsub) / (first_ms * 1e-3) / L1_FILL_GATHERS_PER_S if use_first else None}}

    def step_algorithmic(w, last, n_acc):
        """SURVEY 8(d)'s algorithmic bytes of the WHOLE step (what a CSR fp64 implementation of every stage would move), so that the
        stage-level fraction is on record every round: max_offdiag 8 nnz + 8 n; site norm 20 nnz + 8 n; KR (n_iter + 1) SpMVs of
        12 nnz + 20 n, n_iter vector steps of 96 n, the final X A X 20 nnz + 8 n (the step forms only its masked part: see
        "processed_map"); compress 12 nnz + 12 nnz' + n; Louvain 16 nnz_l + 28 n_l per sweep, 2 * 16 nnz_l per aggregation."""
        nnz, n = float(w.nnz), float(w.n)
        it = float(last.kr_info.n_iter)
        sub = float(last.sub_nnz or 0)
        lv = last.lv_info
        parts = {'max_offdiag': 8 * nnz + 8 * n, 'site_norm': 20 * nnz + 8 * n,
                 'kr': (it + 1) * (12 * nnz + 20 * n) + it * 96 * n + 20 * nnz + 8 * n,
                 'compress': 12 * nnz + 12 * sub + n,
                 'louvain': 16.0 * lv.move_nnz + 28.0 * lv.move_rows + 32.0 * lv.agg_nnz}
        return parts

    def line(w, dt, last, acc, steps, warmup, sharded, extra):
        ms_per_step = dt / steps * 1e3
        e_total = float(w.e_true) if (sharded or world == 1) else float(w.e_true) * world
        value = e_total * steps / dt
        acc['steps'] = steps
        spmv_ms, spmv_n, move_ms = acc['spmv_ms'], acc['spmv_n'], acc['move_ms']
        n_acc = int(last.mask.sum())
        # (this rank's launches: a row range of the map when sharded)
        share = 1.0 / world if sharded else 1.0
        r_spmv = spmv_roofline(w, last, acc, share, sharded)
        r_lv = louvain_roofline(w, last, acc, share, n_acc, sharded)
        alg = step_algorithmic(w, last, n_acc)
        alg_total = sum(alg.values())
        out = {
            'metric': 'contact-graph edges/s through KR-normalise + Louvain',
            'value': value, 'unit': 'edges/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong' if sharded else 'weak',
            'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': w.label,
                       'kr_max_iter': max_iter, 'kr_max_iter_reference_default': 1000,
                       'parallelism': 'single GPU' if world == 1 else (
                           'one map over {} GPUs by contig-row blocks (each rank forms and works on its rows of the ordered map); RCCL: 1 exchange per SpMV, 1 per Louvain sub-round'.format(world)
                           if sharded else 'replicas x{} (one map per GPU)'.format(world)),
                       # what the step does NOT write: the full balanced map the reference caches (contact_map.py:945) is formed on
                       # first access; the step computes its masked part straight from the counts (bit for bit the same values)
                       'processed_map': 'lazy',
                       'accepted_contigs': n_acc, 'kr_spmv_per_step': int(spmv_n // steps),
                       'louvain_sweeps_per_step': int(last.lv_info.n_sweeps), 'louvain_levels': int(last.lv_info.n_levels),
                       'communities_level0': int(last.lv_info.n_comm_level0),
                       'communities': int(last.lv_info.n_comm), 'modularity': last.lv_info.modularity,
                       'stage_ms': {k: v / steps * 1e3 for k, v in acc['stage'].items()},
                       'kernel_ms': {'kr_spmv_total': spmv_ms / steps, 'louvain_decide_total': move_ms / steps}},
            # fixed order: `roofline` is ALWAYS the KR SpMV, `roofline_other` ALWAYS the Louvain delta-Q scan; which of the two
            # families holds more device time in this run is said by name, so a millisecond either way cannot swap the headline
            'roofline': r_spmv,
            'roofline_other': r_lv,
            'dominant': 'kr_spmv' if spmv_ms >= move_ms else 'louvain_scan',
            # SURVEY 8(d)'s algorithmic bytes of the whole step over its wall time: an EQUIVALENT rate (the encodings stream fewer
            # bytes than a CSR fp64 implementation would), printed so that the stage-level figure is on record every round
            'step_algorithmic': {'bytes': alg_total * share, 'GB_per_s_equiv': alg_total * share / (ms_per_step * 1e-3) / 1e9,
                                 'frac_of_peak_equiv': alg_total * share / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 'by_stage_GB': {k: v * share / 1e9 for k, v in alg.items()}},
        }
        out.update(extra)
        return out

    def variant_leg():
        """The genome-size law rounds 1-4 were tuned on (G = n/500 genomes ~ rank^-0.5: an easy planted structure, one Louvain
        level) on the same GPU right after the headline leg, 3 steps, its own stage times and rooflines: a labelled extra, NOT the
        contract workload."""
        ws = build_workload(args.config, 'sqrt', False)
        try:
            dt, last, acc = timed_leg(ws, 3, 2)
            o = line(ws, dt, last, acc, 3, 2, False, {})
        finally:
            ws.counts.free()
        keep = ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'roofline', 'roofline_other', 'step_algorithmic')
        res = {'workload': o['config']['workload']}
        res.update({k: o[k] for k in keep})
        res.update({k: o['config'][k] for k in ('accepted_contigs', 'kr_spmv_per_step', 'louvain_sweeps_per_step', 'louvain_levels',
                                                'communities_level0', 'communities', 'modularity', 'stage_ms', 'kernel_ms')})
        return res

    def emit(out):
        if rank == 0:
            counts, n, nnz, e_true = W.counts, W.n, W.nnz, W.e_true
            sites_h, lengths_h, seed = W.sites_h, W.lengths_h, W.seed
            if world == 1:
                copy = measured_copy_gbs()                            # what a plain copy reaches on this box, beside `peak`
                for key in ('roofline', 'roofline_other'):
                    if key in out:
                        out[key]['copy_gbs'] = copy
            if world == 1 and not args.no_cpu_baseline:
                nthr = min(64, len(os.sched_getaffinity(0))) if hasattr(os, 'sched_getaffinity') else 1
                if args.cpu_baseline == 'same' or args.config != 'C3':
                    # like for like: the same map, downloaded from HBM (SURVEY 8(d) ii and iii)
                    hip_, hix_, hcnt_ = counts.download()
                    what = '{} (the map the GPU ran: n={} E={} nnz={}, {})'.format(
                        args.config, n, e_true, nnz, 'genome sizes as SURVEY 8(d)' if W.synth == 'survey' else 'rank^-0.5 genome-size variant')
                    base_args = (hip_, hix_, hcnt_, sites_h, lengths_h, e_true, what)
                else:
                    from bin3c_amd.synth import make_synth
                    sm = make_synth(200000, 10000000, seed, genome_law='survey' if W.synth == 'survey' else 'sqrt')
                    base_args = (sm.indptr, sm.indices, sm.counts, sm.sites, sm.lengths, sm.n_edges,
                                 'hic-synth-v1 sample n=200000 E={} seed={} ({})'.format(
                                     sm.n_edges, seed, 'genome sizes as SURVEY 8(d)' if W.synth == 'survey' else 'rank^-0.5 variant'))
                if nthr > 1:
                    # the multi-thread figure first: it also tells how long the one-thread run will take
                    out['cpu_baseline_mt'] = cpu_baseline(*base_args, threads=nthr)
                out['cpu_baseline'] = cpu_baseline(*base_args, threads=1)
                out['cpu_baseline']['gpu_speedup'] = out['value'] / out['cpu_baseline']['value']
                if nthr > 1:
                    out['cpu_baseline_mt']['gpu_speedup'] = out['value'] / out['cpu_baseline_mt']['value']
            print(json.dumps(out), flush=True)

    want_replicas = world == 1 or args.mode in ('auto', 'replicas')
    want_sharded = (world > 1 or one_rank_group) and args.mode in ('auto', 'sharded')
    rep = None
    if want_replicas:
        dt, last, acc = timed_leg(W, args.steps, args.warmup)
        rep = line(W, dt, last, acc, args.steps, args.warmup, False, {})
        rep_ref = last
        # the variant law beside the headline (single GPU, the default C3 / C2 legs only)
        if world == 1 and args.synth == 'survey' and not args.hard and args.config in ('C2', 'C3') and not args.no_variant_leg:
            rep['variant_workload'] = variant_leg()
    if not want_sharded:
        emit(rep)
    else:
        import threading
        from bin3c_amd.dist import DistSession

        def give_up(why):
            # a failed or hung sharded leg is a FAILED run: the replica figure (if any) is printed for the record with
            # "value": null and the failure under "sharded", and every rank leaves with a non-zero status
            if rank == 0:
                if rep is not None:
                    rep['replicas'] = {'value': rep['value'], 'unit': 'edges/s', 'ms_per_step': rep['ms_per_step'],
                                       'scaling': 'weak', 'parallelism': rep['config']['parallelism']}
                    rep['value'] = None
                    rep['scaling'] = 'strong'
                    rep['sharded'] = {'error': why}
                    print(json.dumps(rep), flush=True)
                sys.stderr.write('[bench] sharded leg failed: {}\n'.format(why))
                sys.stderr.flush()
            os._exit(1)

        watchdog = threading.Timer(float(os.environ.get('B3C_BENCH_WATCHDOG', '240')),
                                   lambda: give_up('timed out (a collective never returned)'))
        watchdog.daemon = True
        watchdog.start()
        try:
            sess = DistSession(ctx, W.n)
            if os.environ.get('B3C_BENCH_FAIL_SHARDED'):          # test hook (tests/test_dist.py)
                raise RuntimeError('sharded leg disabled by B3C_BENCH_FAIL_SHARDED')
            dt, last, acc = timed_leg(W, args.steps, args.warmup, hand_over=True)
            sess.check()
            n_exchanges = sess.n_exchanges
            n_direct = sess.n_exchanges_direct
            transport = sess.transport
            sess.close()
        except Exception as ex:
            watchdog.cancel()
            give_up(repr(ex))
        watchdog.cancel()
        per = args.steps + args.warmup
        extra = {'exchanges_per_step': n_exchanges // per,
                 # by kind: collectives (RCCL in libb3c, or the callback transport) / peer writes into IPC windows (B3C_DIST_TRANSPORT=direct)
                 'exchanges_per_step_by_kind': {'collective': (n_exchanges - n_direct) // per, 'direct': n_direct // per},
                 'transport': transport,
                 'raw_map_handed_over': acc.get('raw_map_handed_over', False),
                 'louvain_levels_sharded': int(getattr(last.lv_info, 'levels_sharded', 0))}
        if rep is not None:
            extra['replicas'] = {'value': rep['value'], 'unit': 'edges/s', 'ms_per_step': rep['ms_per_step'],
                                 'scaling': 'weak', 'parallelism': rep['config']['parallelism']}
            if rank == 0:
                extra['agrees_with_single_gpu'] = bool(
                    last.kr_info.n_iter == rep_ref.kr_info.n_iter and np.allclose(last.x, rep_ref.x, rtol=1e-9, atol=0)
                    and last.lv_info.n_comm == rep_ref.lv_info.n_comm and np.array_equal(last.mask, rep_ref.mask))
        emit(line(W, dt, last, acc, args.steps, args.warmup, True, extra))
    if world > 1 or one_rank_group:
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()
