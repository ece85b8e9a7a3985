#!/usr/bin/env python3
"""
bin3C command line, hot-path edition: the ``cluster`` sub-command of the reference (bin3C.py:79-95, 167-211)
with the same options, log format and output files for the accelerated path -- clustering.mcl,
clustering.p.gz, bin3C.log -- running mask -> normalise -> KR -> Louvain on the MI355X.

``mkmap`` (BAM/FASTA ingest, bin3C.py:60-74) is outside the accelerated path; ``mksynth`` writes a
synthetic contact_map.p.gz of the same object type so that the cluster stage can be exercised end to end.
The cluster heat-map (bin3C.py:208-211, cluster_plot.png) is drawn from a raster that is permuted and
block-reduced on the device; reports and per-cluster FASTA (bin3C.py:199-206) are downstream I/O of the
reference and are not produced here.
"""
import argparse
import logging
import os
import sys

from bin3c_amd.mzd.cluster import cluster_map, plot_clusters, write_mcl
from bin3c_amd.mzd.exceptions import ApplicationException
from bin3c_amd.mzd.io_utils import load_object, save_object
from bin3c_amd.mzd.utils import make_dir, make_random_seed

__version__ = '0.1.1-mi355x'

RUNTIME_DEFAULTS = {'min_reflen': 1000, 'min_signal': 5, 'max_image': 4000, 'min_extent': 50000, 'min_mapq': 60,
                    'strong': 10}


def ifelse(arg, default):
    return default if arg is None else arg


def build_parser():
    glob = argparse.ArgumentParser(add_help=False)
    glob.add_argument('-V', '--version', default=False, action='store_true', help='Show the application version')
    glob.add_argument('-v', '--verbose', default=False, action='store_true', help='Verbose output')
    glob.add_argument('--clobber', default=False, action='store_true', help='Clobber existing files')
    glob.add_argument('--log', help='Log file path [OUTDIR/bin3C.log]')
    glob.add_argument('--max-image', type=int, help='Maximum image size for plots [4000]')
    glob.add_argument('--min-extent', type=int, help='Minimum cluster extent used in output [50000]')
    glob.add_argument('--min-reflen', type=int, help='Minimum acceptable reference length [1000]')
    glob.add_argument('--min-signal', type=int, help='Minimum acceptable signal [5]')

    parser = argparse.ArgumentParser(description='bin3C: a Hi-C based metagenome deconvolution tool (MI355X hot path)')
    sub = parser.add_subparsers(title='commands', dest='command', description='Valid commands',
                                help='choose an analysis stage for further options')
    cmd_cluster = sub.add_parser('cluster', parents=[glob], description='Cluster an existing contact map into genome bins.')
    cmd_cluster.add_argument('-s', '--seed', default=None, help='Random seed')
    cmd_cluster.add_argument('--no-report', default=False, action='store_true', help='Accepted for compatibility')
    cmd_cluster.add_argument('--no-spades', default=False, action='store_true', help='Accepted for compatibility')
    cmd_cluster.add_argument('--no-plot', default=False, action='store_true', help='Do not generate a clustered heatmap')
    cmd_cluster.add_argument('--no-fasta', default=False, action='store_true', help='Accepted for compatibility')
    cmd_cluster.add_argument('--only-large', default=False, action='store_true', help='Accepted for compatibility')
    cmd_cluster.add_argument('--algo', default='louvain', choices=['louvain', 'infomap', 'mcl', 'slm', 'simap'],
                             help='Clustering algorithm to apply [louvain on the GPU]')
    cmd_cluster.add_argument('--fasta', default=None, help='Accepted for compatibility')
    cmd_cluster.add_argument('MAP', help='Contact map')
    cmd_cluster.add_argument('OUTDIR', help='Output directory')

    cmd_synth = sub.add_parser('mksynth', parents=[glob], description='Write a synthetic contact map (stands in for mkmap).')
    cmd_synth.add_argument('-n', '--contigs', type=int, default=1000)
    cmd_synth.add_argument('-e', '--edges', type=int, default=50000)
    cmd_synth.add_argument('-s', '--seed', type=int, default=1)
    cmd_synth.add_argument('OUTDIR', help='Output directory')
    return parser


def setup_logging(args):
    logging.captureWarnings(True)
    root = logging.getLogger('')
    root.setLevel(logging.DEBUG)
    formatter = logging.Formatter(fmt='%(levelname)-8s | %(asctime)s | %(name)7s | %(message)s')
    ch = logging.StreamHandler()
    ch.setLevel(logging.DEBUG if args.verbose else logging.INFO)
    ch.setFormatter(formatter)
    root.addHandler(ch)
    log_path = args.log if args.log is not None else os.path.join(args.OUTDIR, 'bin3C.log')
    fh = logging.FileHandler(log_path, mode='a')
    fh.setLevel(logging.DEBUG)
    fh.setFormatter(formatter)
    root.addHandler(fh)
    return logging.getLogger('main')


def main(argv=None):
    os.environ.setdefault('B3C_PIN', '1')      # a command-line run owns its process: stay on the CPUs of the GPU's NUMA node
    parser = build_parser()
    args = parser.parse_args(argv)
    if getattr(args, 'version', False):
        print('bin3C v{}'.format(__version__))
        return 0
    if args.command is None:
        parser.print_help()
        return 1
    try:
        make_dir(args.OUTDIR, args.clobber)
    except IOError as e:
        print('Error: {}'.format(e))
        return 1
    logger = setup_logging(args)
    logger.debug('bin3C v{}'.format(__version__))
    logger.debug(sys.version.replace('\n', ' '))
    logger.debug('Command line: {}'.format(' '.join(sys.argv)))
    try:
        if args.command == 'mksynth':
            from bin3c_amd.mzd.contact_map import ContactMap
            from bin3c_amd.synth import make_synth
            sm = make_synth(args.contigs, args.edges, args.seed)
            cm = ContactMap.from_arrays(sm.to_scipy(), sm.lengths, sm.sites,
                                        min_len=ifelse(args.min_reflen, RUNTIME_DEFAULTS['min_reflen']),
                                        min_sig=ifelse(args.min_signal, RUNTIME_DEFAULTS['min_signal']),
                                        min_extent=ifelse(args.min_extent, RUNTIME_DEFAULTS['min_extent']))
            logger.info('Saving contact map instance')
            save_object(os.path.join(args.OUTDIR, 'contact_map.p'), cm)
        elif args.command == 'cluster':
            if not args.seed:
                args.seed = make_random_seed()
                logger.info('Generated random seed: {}'.format(args.seed))
            else:
                logger.info('User set random seed: {}'.format(args.seed))
            logger.info('Loading existing contact map from: {}'.format(args.MAP))
            cm = load_object(args.MAP)
            cm.min_extent = ifelse(args.min_extent, RUNTIME_DEFAULTS['min_extent'])
            if args.min_signal is not None or args.min_reflen is not None:
                min_reflen = ifelse(args.min_reflen, RUNTIME_DEFAULTS['min_reflen'])
                min_signal = ifelse(args.min_signal, RUNTIME_DEFAULTS['min_signal'])
                cm.min_len = min_reflen
                cm.min_sig = min_signal
                cm.set_primary_acceptance_mask(min_sig=min_signal, min_len=min_reflen, update=True)
            clustering = cluster_map(cm, method=args.algo, seed=args.seed, work_dir=args.OUTDIR)
            write_mcl(cm, os.path.join(args.OUTDIR, 'clustering.mcl'), clustering)
            save_object(os.path.join(args.OUTDIR, 'clustering.p'), clustering)
            logger.info('Wrote {} clusters'.format(len(clustering)))
            if not args.no_plot:
                # the entire clustering (bin3C.py:208-211)
                plot_clusters(cm, os.path.join(args.OUTDIR, 'cluster_plot.png'), clustering,
                              max_image_size=ifelse(args.max_image, RUNTIME_DEFAULTS['max_image']),
                              ordered_only=False, simple=False, permute=True)
    except ApplicationException as ex:
        logger.error(ex.message)
        return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())
