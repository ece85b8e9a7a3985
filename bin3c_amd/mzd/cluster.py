"""
Host-side mirror of the reference's ``mzd/cluster.py`` for the cluster hot path.

    cluster_map        mzd/cluster.py:44-226   solver dispatch + integer standardisation of the result
    to_graph           mzd/cluster.py:278-325  contact map -> weighted graph (here: CSR triple or networkx)
    add_cluster_names  mzd/cluster.py:23-41
    write_mcl          mzd/cluster.py:521-548
    readers/writers of the solver wire format: _read_mcl :57, _read_table :76, _read_tree :111, _write_edges :139

``method='louvain'`` runs mask -> normalise -> KR -> compress -> Louvain on the GPU through
bin3c_amd.pipeline (the reference's own louvain branch is latently broken downstream of
louvain.cluster, cluster.py:208-209; the reader contract ``{cluster_id: ndarray of gapless seq ids}`` is
what is honoured here).  External solvers still plug in through the file formats below.
"""
import logging
import os
import subprocess

import numpy as np
import scipy.sparse as sp

from .contact_map import SeqOrder
from .exceptions import NoneAcceptedException, NoRemainingClustersException
from .utils import app_path

logger = logging.getLogger(__name__)


def add_cluster_names(clustering, prefix='CL'):
    """Name clusters CL1.. with enough zero padding for the largest id (cluster.py:23-41); in place."""
    try:
        num_width = max(1, int(np.ceil(np.log10(max(clustering) + 1))))
    except OverflowError:
        num_width = 1
    for cl_id in clustering:
        clustering[cl_id]['name'] = '{0}{1:0{2}d}'.format(prefix, cl_id + 1, num_width)


# ---- solver wire formats --------------------------------------------------------------------------

def _records(pathname, skip_comments=False):
    """The stripped lines of a solver output file up to the first empty one (where the reference's readers stop)."""
    with open(pathname, 'r') as handle:
        for raw in handle:
            text = raw.strip()
            if not text:
                return
            if skip_comments and text[0] == '#':
                continue
            yield text


def _grouped(keys, members):
    """[(key, int64 array of its members in file order)] in order of the keys' first appearance -- the iteration order of the
    dict the reference fills line by line.  One stable argsort instead of a Python append per line."""
    members = np.asarray(members, dtype=np.int64)
    if members.shape[0] == 0:
        return []
    uniq, first, inverse = np.unique(np.asarray(keys), return_index=True, return_inverse=True)
    by_key = np.argsort(inverse, kind='stable')
    ends = np.cumsum(np.bincount(inverse, minlength=uniq.shape[0]))
    chunks = np.split(members[by_key], ends[:-1])
    return [(uniq[k], chunks[k]) for k in np.argsort(first, kind='stable')]


def read_mcl(pathname):
    """MCL output: one cluster per line, space separated gapless ids, returned ascending; cluster id = line number
    (cluster.py:57-74)."""
    return {cl_id: np.sort(np.array(text.split(), dtype=np.int64)) for cl_id, text in enumerate(_records(pathname))}


def read_table(pathname, seq_col=0, cl_col=1):
    """``seq cl`` pairs, or with seq_col=None one cluster id per line for sequence 0, 1, 2 ... (cluster.py:76-109)."""
    assert seq_col != cl_col, 'sequence and cluster columns must be different'
    lines = list(_records(pathname))
    if seq_col is None:
        clusters = [int(text) for text in lines]
        sequences = np.arange(len(lines), dtype=np.int64)
    else:
        tokens = [text.split() for text in lines]
        for text, tok in zip(lines, tokens):
            if len(tok) != 2:
                logger.warning('invalid line encountered when reading cluster table: {}'.format(text))
        clusters = [int(tok[cl_col]) for tok in tokens]
        sequences = [int(tok[seq_col]) for tok in tokens]
    return {int(cl_id): ids for cl_id, ids in _grouped(clusters, sequences)}


def read_tree(pathname):
    """Infomap ``.tree``: a leaf line is ``path flow "name" id``; its cluster is the path minus the leaf position, and clusters
    are renumbered by descending size, equal sizes in order of first appearance (cluster.py:111-137)."""
    paths, leaves = [], []
    for text in _records(pathname, skip_comments=True):
        fields = text.split()
        paths.append(fields[0].rpartition(':')[0])
        leaves.append(int(fields[-1]))
    groups = _grouped(paths, leaves)
    ranked = sorted(range(len(groups)), key=lambda k: -groups[k][1].shape[0])      # stable: ties keep first-appearance order
    return {rank: groups[k][1] for rank, k in enumerate(ranked)}


def write_edges(graph, parent_dir, base_name, sep=' '):
    """``u v weight`` per undirected edge, the file every external solver reads (cluster.py:139-151).
    ``graph`` is the (indptr, indices, weights) triple of :func:`to_graph`.  The file is byte for byte what
    ``nx.write_edgelist(g, f, data=['weight'], delimiter=sep)`` writes for the graph cluster.py:316-321 builds from the
    same map (entries added in row-major order): networkx walks the nodes in insertion order -- the order of first
    appearance in the (row, col) stream -- and prints every edge once, from the endpoint inserted first, neighbours in
    the order they were attached (ascending ids here); self-loops included; weights as ``str(float)``
    (tests/golden/edges_nx.npz holds networkx's own bytes)."""
    edge_file = os.path.join(parent_dir, '{}.edges'.format(base_name))
    ip, ix, w = graph
    n = ip.shape[0] - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ip))
    cols = np.asarray(ix, dtype=np.int64)
    # node insertion order = first appearance in r0, c0, r1, c1, ...
    stream = np.empty(2 * rows.shape[0], dtype=np.int64)
    stream[0::2] = rows
    stream[1::2] = cols
    nodes, first = np.unique(stream, return_index=True)
    rank = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
    rank[nodes[np.argsort(first, kind='stable')]] = np.arange(nodes.shape[0])
    keep = rows <= cols                                    # every undirected edge once (the map is symmetric)
    u, v, ww = rows[keep], cols[keep], np.asarray(w)[keep]
    swap = rank[v] < rank[u]
    a = np.where(swap, v, u)
    b = np.where(swap, u, v)
    order = np.lexsort((b, rank[a]))
    with open(edge_file, 'w') as out:
        for x, y, z in zip(a[order].tolist(), b[order].tolist(), ww[order].tolist()):
            out.write('{}{}{}{}{}\n'.format(x, sep, y, sep, repr(z)))
    return edge_file


# ---- graph ------------------------------------------------------------------------------------------

def to_graph(contact_map, norm=True, bisto=False, scale=False, extern_ids=False, min_len=None, min_sig=None,
             as_networkx=False):
    """
    cluster.py:278-325: accept/refresh the mask, prepare the map if needed, take the accepted subspace and
    scale the weights by 1/max.  Returns the graph as a symmetric CSR triple (indptr, indices, weights)
    whose both directions hold the value nx.Graph would hold (the later, lower-triangle entry wins,
    cluster.py:319-321) -- or a networkx.Graph with as_networkx=True for code that wants the reference's type.
    """
    if min_len or min_sig:                      # explicit criteria replace a cached mask (cluster.py:301-304)
        contact_map.set_primary_acceptance_mask(min_len, min_sig, update=True)
    else:
        contact_map.set_primary_acceptance_mask()
    if contact_map.processed_map is None:
        contact_map.prepare_seq_map(norm=norm, bisto=bisto)
    c = sp.csr_matrix(contact_map.get_subspace(marginalise=True, flatten=False))
    logger.info('Graph will have {} nodes'.format(contact_map.order.count_accepted()))
    c.sort_indices()
    scl = 1.0 / c.max() if scale else 1
    # lower triangle wins: mirror it over the upper triangle
    low = sp.tril(c, k=0, format='csr')
    sym = low + sp.tril(c, k=-1, format='csr').T
    sym = sym.tocsr()
    sym.sort_indices()
    w = sym.data * scl
    if not as_networkx:
        return sym.indptr.astype(np.int64), sym.indices.astype(np.int32), w
    import networkx as nx
    g = nx.Graph(name='contact_graph')
    _nn = (lambda x: contact_map.seq_info[x].name) if extern_ids else (lambda x: x)
    coo = c.tocoo()
    for u, v, ww in zip(coo.row.tolist(), coo.col.tolist(), coo.data.tolist()):
        g.add_edge(_nn(u), _nn(v), weight=ww * scl)
    return g


# ---- clustering -------------------------------------------------------------------------------------

def _louvain_device(contact_map, min_len, min_sig):
    """mask -> normalise -> KR -> compress -> Louvain without leaving HBM; -> {cluster: gapless ids}."""
    from ..pipeline import run_hot_path, cluster_prepared
    if not min_len and not min_sig:
        mask = contact_map.set_primary_acceptance_mask()
    else:
        mask = contact_map.set_primary_acceptance_mask(min_len, min_sig, update=True)
    contact_map.order.set_mask_only(mask)
    if contact_map.order.count_accepted() < 1:          # prepare_seq_map, contact_map.py:926-927
        raise NoneAcceptedException()
    ctx = contact_map._context()
    hot = contact_map._hot
    if hot is not None and hot.has_processed():
        res = cluster_prepared(ctx, hot, mask)
    elif contact_map.processed_map is not None:
        # a map that was prepared in an earlier session: upload the cached balanced matrix
        c = contact_map.processed_map.tocsr()
        c.sort_indices()
        from ..pipeline import HotPathResult
        res = HotPathResult()
        res.processed = ctx.upload(c.indptr, c.indices, c.data.astype(np.float64))
        contact_map._hot = res
        res = cluster_prepared(ctx, res, mask)
    else:
        c = contact_map._counts_csr()
        counts = ctx.upload(c.indptr, c.indices, c.data)
        try:
            res = run_hot_path(ctx, counts, contact_map._get_sites(), contact_map.order.lengths(),
                               contact_map.min_len, contact_map.min_sig, mask=mask, keep=True)
        finally:
            counts.free()
        contact_map._report_kr(res.kr_info)
        contact_map.bisto_scale = res.x
        contact_map._processed_host = None
        contact_map._hot = res
    logger.info('Graph will have {} nodes'.format(contact_map.order.count_accepted()))
    labels = res.labels
    info = res.lv_info
    logger.info('Louvain: {} communities, {} levels, {} sweeps, modularity {:.6f}'.format(
        info.n_comm, info.n_levels, info.n_sweeps, info.modularity))
    order = np.argsort(labels, kind='stable')
    srt = labels[order]
    first = np.flatnonzero(np.concatenate(([True], srt[1:] != srt[:-1])))
    bounds = np.concatenate((first, [len(srt)]))
    return {int(srt[bounds[i]]): order[bounds[i]:bounds[i + 1]] for i in range(len(first)) if srt[bounds[i]] >= 0}


def cluster_map(contact_map, seed, method='infomap', min_len=None, min_sig=None, work_dir='.'):
    """
    Cluster a contact map into genome bins (cluster.py:44-226).

    :param contact_map: ContactMap
    :param seed: random seed handed to external solvers (the GPU Louvain is deterministic)
    :param method: the reference's signature default 'infomap' is kept (cluster.py:44); 'louvain' runs the hot path on
                   the GPU (what ``bin3C.py cluster`` passes here); 'infomap', 'mcl', 'slm', 'simap' need their
                   executable under external/
    :return: {rank: {'seq_ids': ndarray, 'extent': int, 'name': str}} ordered by descending extent
    """
    assert os.path.exists(work_dir), 'supplied output path [{}] does not exist'.format(work_dir)
    base_name = 'cm_graph'
    method = method.lower()
    logger.info('Clustering contact graph using method: {}'.format(method))

    if method == 'louvain':
        cl_to_ids = _louvain_device(contact_map, min_len, min_sig)
    elif method in ('infomap', 'mcl', 'slm', 'simap'):
        g = to_graph(contact_map, min_len=min_len, min_sig=min_sig, norm=True, bisto=True, scale=True)
        cl_to_ids = _run_external(method, g, seed, work_dir, base_name)
    else:
        raise RuntimeError('unimplemented method: {}'.format(method))

    logger.info('Clustering using {} resulted in {} clusters'.format(method, len(cl_to_ids)))
    return standardise(contact_map, cl_to_ids)


def standardise(contact_map, cl_to_ids):
    """cluster.py:205-226: ascending member ids, gapless -> surrogate ids, extent, descending-extent order, names."""
    lengths = contact_map.order.lengths()
    clustering = {}
    for cl_id, _seqs in cl_to_ids.items():
        _ord = SeqOrder.asindex(np.sort(_seqs))
        _seqs = contact_map.order.remap_gapless(_ord)['index']
        clustering[cl_id] = {'seq_ids': _seqs, 'extent': lengths[_seqs].sum()}
    sorted_keys = sorted(clustering, key=lambda k: clustering[k]['extent'], reverse=True)
    clustering = {n: clustering[k] for n, k in enumerate(sorted_keys)}
    add_cluster_names(clustering)
    return clustering


def _run_external(method, g, seed, work_dir, base_name):
    """The reference's subprocess branches (cluster.py:163-199); needs the third-party executables."""
    def need(path):
        if not os.path.exists(path):
            raise RuntimeError('external solver {} is not installed (bin3C bundles it under external/)'.format(path))
        return path

    if method == 'mcl':
        ofile = os.path.join(work_dir, '{}.mcl'.format(base_name))
        edge_file = write_edges(g, work_dir, base_name)
        with open(os.path.join(work_dir, 'mcl.log'), 'w+') as stdout:
            subprocess.check_call([need(app_path('external', 'mcl')), edge_file, '--abc', '-I', '1.2', '-o', ofile],
                                  stdout=stdout, stderr=subprocess.STDOUT)
        return read_mcl(ofile)
    if method == 'infomap':
        edge_file = write_edges(g, work_dir, base_name)
        with open(os.path.join(work_dir, 'infomap.log'), 'w+') as stdout:
            subprocess.check_call([need(app_path('external', 'Infomap')), '-u', '-v', '-z', '-i', 'link-list', '-s',
                                   str(seed), '-N', '10', edge_file, work_dir], stdout=stdout, stderr=subprocess.STDOUT)
        return read_tree(os.path.join(work_dir, '{}.tree'.format(base_name)))
    if method == 'simap':
        ofile = os.path.join(work_dir, '{}.simap'.format(base_name))
        edge_file = write_edges(g, work_dir, base_name)
        with open(os.path.join(work_dir, 'simap.log'), 'w+') as stdout:
            subprocess.check_call(['java', '-jar', need(app_path('external', 'simap-1.0.0.jar')), 'mdl', '-s', str(seed),
                                   '-i', '1e-5', '1e-3', '-a', '1e-5', '-g', edge_file, '-o', ofile],
                                  stdout=stdout, stderr=subprocess.STDOUT)
        return read_table(ofile)
    ofile = os.path.join(work_dir, '{}.slm'.format(base_name))
    edge_file = write_edges(g, work_dir, base_name, sep='\t')
    with open(os.path.join(work_dir, 'slm.log'), 'w+') as stdout:
        subprocess.check_call(['java', '-jar', need(app_path('external', 'ModularityOptimizer.jar')), edge_file, ofile,
                               '1', '2.0', '3', '10', '10', str(seed), '1'], stdout=stdout, stderr=subprocess.STDOUT)
    return read_table(ofile, seq_col=None, cl_col=0)


# the reference's private names for the wire-format functions (cluster.py:57, :76, :111, :139)
_read_mcl, _read_table, _read_tree, _write_edges = read_mcl, read_table, read_tree, write_edges


def _select_clusters(clustering, wanted, min_extent, ordered_only):
    """The cluster ids of `wanted` that pass the enabled criteria, ascending; every criterion logs how many are left and
    raises when none is (the messages are the reference's, cluster.py:348-362)."""
    criteria = []
    if min_extent:
        criteria.append(('minimum extent', lambda k: clustering[k]['extent'] >= min_extent,
                         'No clusters passed min_extent criterion of >= {}'.format(min_extent)))
    if ordered_only:
        criteria.append(('ordered-only', lambda k: 'order' in clustering[k], 'No clusters passed ordered-only criterion'))
    left = list(wanted)
    for label, passes, complaint in criteria:
        left = [k for k in left if passes(k)]
        logger.info('Clusters passing {} criterion: {}'.format(label, len(left)))
        if not left:
            raise NoRemainingClustersException(complaint)
    return sorted(left)


def enable_clusters(contact_map, clustering, cl_list=None, ordered_only=True, min_extent=None):
    """
    Unmask the sequences of the given clusters (all of them for cl_list=None) in the contact map and put them in cluster
    order: each cluster's determined order and orientation with ordered_only, else its sequence ids as they come
    (cluster.py:327-387).  Clusters under `min_extent` (default: the map's own) or, with ordered_only, without an order are
    dropped.

    :return: the ids of the clusters that remain, ascending
    """
    chosen = _select_clusters(clustering, clustering.keys() if cl_list is None else cl_list,
                              contact_map.min_extent if min_extent is None else min_extent, ordered_only)
    placement = np.hstack([clustering[k]['order'] if ordered_only else SeqOrder.asindex(clustering[k]['seq_ids']) for k in chosen])
    if placement.shape[0] == 0:
        raise NoRemainingClustersException('No requested cluster contained ordering information')
    logger.info('Total number of sequences in the clustering: {}'.format(placement.shape[0]))
    shown = np.zeros(contact_map.order.mask_vector().shape, dtype=np.bool_)
    shown[placement['index']] = True
    shown &= contact_map.get_primary_acceptance_mask()
    logger.info('After joining with active sequence mask map: {}'.format(int(shown.sum())))
    contact_map.order.set_mask_only(shown)
    contact_map.order.set_order_and_orientation(placement, implicit_excl=True)
    return chosen


def plot_clusters(contact_map, fname, clustering, cl_list=None, simple=True, permute=False, max_image_size=None,
                  ordered_only=False, min_extent=None, use_taxo=False, flatten=False, **kwargs):
    """
    The heat-map of the clustered map with one tick per cluster, labelled by cluster name (or taxon with use_taxo)
    (cluster.py:390-449).  The raster is permuted and block-reduced on the device (``max_image_size``).
    """
    logger.info('Plotting heatmap of complete solution' if cl_list is None
                else 'Plotting heatmap for {} specified clusters'.format(len(cl_list)))
    assert simple or contact_map.bin_size is None, 'extent (binned) maps are outside the accelerated path'
    if not contact_map.has_processed_map():
        contact_map.prepare_seq_map(norm=True, bisto=True)
    shown = enable_clusters(contact_map, clustering, cl_list=cl_list, ordered_only=ordered_only, min_extent=min_extent)
    sizes = np.array([len(clustering[k]['seq_ids']) for k in shown], dtype=np.int64)
    label_key = 'taxon' if use_taxo else 'name'
    contact_map.plot(fname, permute=permute, simple=simple, tick_locs=np.concatenate(([0], np.cumsum(sizes))),
                     tick_labs=[clustering[k][label_key] for k in shown], max_image_size=max_image_size, flatten=flatten, **kwargs)


def write_mcl(contact_map, fname, clustering):
    """clustering.mcl (cluster.py:521-548): one line of sequence names per cluster; filtered-out sequences become singletons."""
    seq_info = contact_map.seq_info
    lost = np.ones(contact_map.total_seq, dtype=np.bool_)
    cl_soln = {}
    for k, v in clustering.items():
        lost[v['seq_ids']] = False
        cl_soln[k] = [seq_info[ix].name for ix in np.sort(v['seq_ids'])]
    for n, ix in enumerate(np.flatnonzero(lost), len(cl_soln)):
        cl_soln[n] = [seq_info[ix].name]
    with open(fname, 'w') as outh:
        for k in sorted(cl_soln.keys()):
            outh.write(' '.join(cl_soln[k]))
            outh.write('\n')
