"""Summarise rocprofv3 --pmc passes: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:40]
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, ctrs in acc.items():
    print(name)
    for c, v in sorted(ctrs.items()):
        print('   %-32s n=%4d mean=%.6g' % (c, len(v), sum(v) / len(v)))
