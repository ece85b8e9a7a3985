"""Per-LAUNCH counters of one kernel from the passes tools/pmc_kr.sh collected (the k-th dispatch of the kernel is the same launch in
every pass: the launch order of a step is deterministic).  usage: tools/pmc_by_launch.py gpurun_out/<tag> '<kernel name prefix>' [first n]"""
import collections, csv, glob, sys
src, kern = sys.argv[1], sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 100
cols = collections.OrderedDict()
for f in sorted(glob.glob(src + '/p*/*/*counter_collection.csv')):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        nm = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
        if not nm.startswith(kern):
            continue
        d = by.setdefault(int(r['Dispatch_Id']), {})
        d[r['Counter_Name']] = float(r['Counter_Value'])
        d['_dur_us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    for k, (_id, d) in enumerate(sorted(by.items())):
        for name, v in d.items():
            cols.setdefault(name if name != '_dur_us' else 'dur_us', {})[k] = v
names = list(cols.keys())
n = max(len(v) for v in cols.values())
print('%4s ' % 'k' + ' '.join('%14s' % nm[:14] for nm in names))
for k in range(min(n, first)):
    print('%4d ' % k + ' '.join('%14.0f' % cols[nm].get(k, float('nan')) for nm in names))
