"""Reduce rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, one counter per pass) to profiles/: the counter rows without the
mangled signatures (profiles/<round>_<key>_FETCH_SIZE.csv, ..._WRITE_SIZE.csv) and the records of profiles/traffic.json that
bench.py reads -- HBM bytes per launch of the SpMV (all launches of one product) and of the Louvain scan's first sweep, corrected as
MI355X_MICROARCH.md's HBM section prescribes, with the check on a streaming kernel of known size taken in the same passes.

usage: python tools/traffic_reduce.py <gpurun_out/dir with pmc_fetch/ pmc_write/ and a bench json> <round, e.g. r05> <key>
       key: the name bench.py looks the workload up by -- 'C3' (the contract workload, SURVEY 8(d)'s generator), 'C3_variant' (the
       rank^-0.5 genome-size variant), 'C5' ...; the sizes (n, nnz) are read from the bench line found in the directory
"""
import csv
import glob
import json
import os
import re
import sys

src, rnd, KEY = sys.argv[1], sys.argv[2], sys.argv[3]
cfg = KEY.split('_')[0].lower()            # the bench line of the SAME config carries the sizes (a directory holds several)
bench = (sorted(glob.glob('%s/bench_%s.json' % (src, cfg))) or sorted(glob.glob('%s/bench_%s*.json' % (src, cfg)))
         or sorted(glob.glob('%s/bench*.json' % src) + glob.glob('%s/diag.json' % src)))
line = [l for l in open(bench[0]) if l.startswith('{')][-1]
wl = json.loads(line)['config']['workload']
NNZ, N = int(re.search(r'nnz=(\d+)', wl).group(1)), int(re.search(r'n=(\d+)', wl).group(1))
tag = KEY.lower()

for d, c in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    f = glob.glob('%s/pmc_%s/*/*counter_collection.csv' % (src, d))[0]
    with open('profiles/%s_%s_%s.csv' % (rnd, tag, c), 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['Dispatch_Id', 'Kernel', 'Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'Scratch_Size', 'VGPR_Count',
                    'SGPR_Count', 'Counter_Name', 'Counter_Value', 'Duration_ns'])
        for r in csv.DictReader(open(f)):
            nm = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            w.writerow([r['Dispatch_Id'], nm, r['Grid_Size'], r['Workgroup_Size'], r['LDS_Block_Size'], r['Scratch_Size'],
                        r['VGPR_Count'], r['SGPR_Count'], r['Counter_Name'], r['Counter_Value'],
                        int(r['End_Timestamp']) - int(r['Start_Timestamp'])])


def rows_of(counter, kern):
    return [r for r in csv.DictReader(open('profiles/%s_%s_%s.csv' % (rnd, tag, counter))) if r['Kernel'].startswith(kern)]


def active(counter, kern):
    """mean counter value (bytes; the counters are in KB) over the launches that did work (>= half the longest)"""
    rows = rows_of(counter, kern)
    if not rows:
        return 0.0, 0, 0
    mx = max(int(r['Duration_ns']) for r in rows)
    a = [float(r['Counter_Value']) for r in rows if int(r['Duration_ns']) >= 0.5 * mx]
    return sum(a) / len(a) * 1024.0, len(a), len(rows)


prev = json.load(open('profiles/traffic.json')) if os.path.exists('profiles/traffic.json') else {}
out = dict(prev)
kernels = sorted({r['Kernel'].split('<')[0] for r in csv.DictReader(open('profiles/%s_%s_FETCH_SIZE.csv' % (rnd, tag)))})
fp, _, _ = active('FETCH_SIZE', 'kr_prep_counts')
wp, _, _ = active('WRITE_SIZE', 'kr_prep_counts')
correction = ('MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies the 128-byte fabric requests of coalesced reads at 64 bytes, so it is '
              'doubled; WRITE_SIZE is exact.')
if fp:
    correction += (' Checked in the same two passes on a pure streaming kernel of known size: kr_prep_counts reads 8*nnz+8*n = %.0f MB and '
                   'reports %.0f MB (x%.3f), writes 2*nnz+8*n = %.0f MB and reports %.0f MB.' % (
                       (8 * NNZ + 8 * N) / 1e6, fp / 1e6, fp / (8 * NNZ + 8 * N), (2 * NNZ + 8 * N) / 1e6, wp / 1e6))
source = ('profiles/%s_%s_FETCH_SIZE.csv, profiles/%s_%s_WRITE_SIZE.csv (rocprofv3 --pmc, one counter per pass, bench.py --steps 1 --warmup 0 on: %s)'
          % (rnd, tag, rnd, tag, wl))

# one SpMV = all launches of one product: kr_kb_tails + kr_kb_tcd2 (+ kr_kb_epi for the tiles without a dictionary, + kr_kb_hub),
# or the CSR row kernel
det, tot, na, nt = {}, 0.0, 0, 0
parts = [k for k in ('kr_kb_tails', 'kr_kb_tcd2', 'kr_kb_epi', 'kr_kb_hub', 'kr_kb_vec', 'kr_kb_pipe') if k in kernels]
for k in parts:
    f_, a_, t_ = active('FETCH_SIZE', k)
    w_, _, _ = active('WRITE_SIZE', k)
    det[k] = {'FETCH_SIZE_raw_bytes_per_launch': f_, 'WRITE_SIZE_raw_bytes_per_launch': w_}
    tot += 2.0 * f_ + w_
    if k in ('kr_kb_tails', 'kr_kb_vec', 'kr_kb_pipe') or (k == 'kr_kb_tcd2' and not na):
        na, nt = a_, t_          # (launches that did work: the tail pass where it runs -- kr_kb_tcd2 may be a no-op list of tiles)
if parts:
    name = ' + '.join(parts)
    out[KEY] = dict(det, kernel=name + (' (one SpMV: tails in column order in fixed point, dictionary rows + fused KR epilogue)'
                                        if 'kr_kb_tails' in parts else ' (one SpMV)'),
                    hbm_bytes_per_launch=tot, algorithmic_bytes_per_launch=12.0 * NNZ + 20.0 * N, launches_counted=na, launches_total=nt,
                    correction=correction, source=source)
    print(json.dumps(out[KEY], indent=1))

# the Louvain neighbour scan: its launches in launch order (k_lv_first_g, then k_lv_decide_g); the FIRST sweep of level 0 reads every
# row (n_sub = 4 launches): its HBM bytes per launch are what bench.py's roofline_other.traffic carries
def scan_rows(counter):
    rows = [r for r in csv.DictReader(open('profiles/%s_%s_%s.csv' % (rnd, tag, counter)))
            if r['Kernel'].startswith('k_lv_decide_g<64, 4') or r['Kernel'].startswith('k_lv_first_g') or r['Kernel'].startswith('k_lv_scan')]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return rows


lf, lw = scan_rows('FETCH_SIZE'), scan_rows('WRITE_SIZE')
if lf and len(lf) == len(lw):
    first = [2.0 * float(a['Counter_Value']) * 1024.0 + float(b['Counter_Value']) * 1024.0 for a, b in zip(lf[:4], lw[:4])]
    out[KEY + '_k_lv_scan'] = {
        'kernel': 'k_lv_decide (k_lv_first_g + k_lv_decide_g: Louvain delta-Q neighbour scan, one launch per sub-round)',
        'first_sweep_hbm_bytes_per_launch': sum(first) / len(first),
        'first_sweep_hbm_bytes': first,
        'FETCH_SIZE_raw_bytes_first_12_launches': [float(r['Counter_Value']) * 1024.0 for r in lf[:12]],
        'WRITE_SIZE_raw_bytes_first_12_launches': [float(r['Counter_Value']) * 1024.0 for r in lw[:12]],
        'duration_ns_first_12_launches': [int(r['Duration_ns']) for r in lf[:12]],
        'algorithmic_bytes_per_launch': (16.0 * NNZ + 28.0 * N) / 4.0,
        'note': 'hbm bytes = 2 x FETCH_SIZE + WRITE_SIZE per launch (see correction); SURVEY 8(d) charges 16 B per stored entry + 28 B per node and sweep',
        'correction': correction, 'source': source}
    print(json.dumps(out[KEY + '_k_lv_scan'], indent=1))
json.dump(out, open('profiles/traffic.json', 'w'), indent=1)
