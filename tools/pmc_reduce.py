"""Mean per ACTIVE launch of the counters tools/pmc_kernel.sh collected.  usage: tools/pmc_reduce.py gpurun_out/<tag>"""
import collections, csv, glob, statistics, sys
for f in sorted(glob.glob(sys.argv[1] + '/p*/*/*counter_collection.csv')):
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        by[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    ds = list(by.values())
    keys = sorted(ds[0].keys())
    ref = max(keys, key=lambda k: max(d[k] for d in ds))
    act = [d for d in ds if d[ref] > 0.5 * max(x[ref] for x in ds)]
    print('# %s: %d launches, %d active' % (f.split('/')[-3], len(ds), len(act)))
    for k in keys:
        print('%-45s %14.0f' % (k, statistics.mean(d[k] for d in act)))
