"""Development tool: idle time between consecutive kernels in a rocprofv3 --kernel-trace CSV (per step of bench.py)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:36]) for r in rows)
idx = [i for i, e in enumerate(ev) if e[2].startswith('k_heaviest')]
for a, b in list(zip(idx, idx[1:] + [len(ev)]))[-2:]:
    seg = ev[a:b]
    busy = sum(e[1] - e[0] for e in seg)
    wall = seg[-1][1] - seg[0][0]
    print('step: wall %.2f ms, kernels %.2f ms, idle %.2f ms, %d kernels' % (wall / 1e6, busy / 1e6, (wall - busy) / 1e6, len(seg)))
    gaps = collections.defaultdict(lambda: [0, 0.0])
    for (s0, e0, n0), (s1, e1, n1) in zip(seg, seg[1:]):
        g = gaps[(n0[:14], n1[:14])]
        g[0] += 1
        g[1] += (s1 - e0) / 1e3
    for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print('   %-14s -> %-14s x%-4d %8.1f us total, %6.1f us each' % (k[0], k[1], c, t, t / c))
