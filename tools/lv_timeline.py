#!/usr/bin/env python3
"""GPU box helper: reduce a rocprofv3 --kernel-trace CSV of ONE bench step to a per-sweep table of the Louvain stage
(sweeps are delimited by k_lv_tot2, levels by k_lv_init) and a per-stage table of everything else.
usage: python tools/lv_timeline.py <kernel_trace.csv> [step index from the end, default 1]"""
import csv
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    s = m.group(1) if m else name
    if 'rocprim' in s or 'hipcub' in s:
        k = re.search(r'(radix_sort\w*|scan\w*|reduce_by_key\w*|onesweep\w*|histogram\w*|lookback\w*|transform\w*)', name)
        s = 'rocprim:' + (k.group(1) if k else 'x')
    return s[:44]


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    recs = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows))
    # steps: each begins with the locality order's first kernel (k_heaviest)
    starts = [i for i, r in enumerate(recs) if r[2].startswith('k_heaviest')]
    # two k_heaviest-class launches may exist per step; keep those further than 20 ms apart
    steps = []
    for i in starts:
        if not steps or recs[i][0] - recs[steps[-1]][0] > 20e6:
            steps.append(i)
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lo = steps[-which]
    hi = steps[-which + 1] if which > 1 else len(recs)
    step = recs[lo:hi]
    t0 = step[0][0]
    print('step: %d kernels, wall %.2f ms, busy %.2f ms' % (len(step), (step[-1][1] - t0) / 1e6, sum(e - s for s, e, _ in step) / 1e6))
    # Louvain: from the first k_lv_init
    li = next((i for i, r in enumerate(step) if r[2].startswith('k_lv_init')), None)
    pre = step[:li] if li is not None else step
    agg = OrderedDict()
    for s, e, k in pre:
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += e - s
    print('--- before Louvain: wall %.2f ms' % ((pre[-1][1] - t0) / 1e6))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print('   %-46s x%-5d %9.3f ms' % (k, c, t / 1e6))
    if li is None:
        return
    lv = step[li:]
    level = -1
    sweep = 0
    cur = OrderedDict()
    w0 = lv[0][0]

    def flush(tag, w1):
        nonlocal cur, w0
        if cur:
            tot = sum(t for c, t in cur.values())
            print('%s wall %.3f ms busy %.3f ms' % (tag, (w1 - w0) / 1e6, tot / 1e6))
            for k, (c, t) in sorted(cur.items(), key=lambda kv: -kv[1][1])[:14]:
                print('      %-46s x%-4d %8.3f ms  avg %7.1f us' % (k, c, t / 1e6, t / c / 1e3))
        cur = OrderedDict()
        w0 = w1

    for s, e, k in lv:
        if k.startswith('k_lv_init'):
            flush('L%d tail (aggregation etc.)' % level if level >= 0 else 'pre', s)
            level += 1
            sweep = 0
        a = cur.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += e - s
        if k.startswith('k_lv_tot2'):
            flush('L%d sweep %d' % (level, sweep), e)
            sweep += 1
    flush('L%d tail' % level, lv[-1][1])
    print('Louvain wall %.2f ms' % ((lv[-1][1] - lv[0][0]) / 1e6))


if __name__ == '__main__':
    main()
