"""Reduce the --pmc passes of tools/profile_round.sh to profiles/: the counter rows without the mangled signatures
(profiles/<round>_c3_FETCH_SIZE.csv, ..._WRITE_SIZE.csv) and profiles/traffic.json (bytes per launch of the dominant
kernel, corrected as MI355X_MICROARCH.md's HBM section prescribes, with the check on two streaming kernels of known size).
usage: python tools/traffic_reduce.py gpurun_out/<tag> <round, e.g. r03> [C3|C5]
(C5: tools/run_c5.sh's passes; only the SpMV kernels are collected there, so the calibration kernels are skipped)"""
import csv, glob, json, re, sys

src, rnd = sys.argv[1], sys.argv[2]
CFG = sys.argv[3] if len(sys.argv) > 3 else 'C3'
NNZ, N = 200903927, 1000000            # C3 (bench.py prints them in config.workload)
if CFG != 'C3':
    line = [l for l in open(glob.glob('%s/bench_%s.json' % (src, CFG.lower()))[0]) if l.startswith('{')][-1]
    wl = json.loads(line)['config']['workload']
    NNZ, N = int(re.search(r'nnz=(\d+)', wl).group(1)), int(re.search(r'n=(\d+)', wl).group(1))

for tag, c in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    f = glob.glob('%s/pmc_%s/*/*counter_collection.csv' % (src, tag))[0]
    with open('profiles/%s_%s_%s.csv' % (rnd, CFG.lower(), c), 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['Dispatch_Id', 'Kernel', 'Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'Scratch_Size', 'VGPR_Count',
                    'SGPR_Count', 'Counter_Name', 'Counter_Value', 'Duration_ns'])
        for r in csv.DictReader(open(f)):
            nm = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            w.writerow([r['Dispatch_Id'], nm, r['Grid_Size'], r['Workgroup_Size'], r['LDS_Block_Size'], r['Scratch_Size'],
                        r['VGPR_Count'], r['SGPR_Count'], r['Counter_Name'], r['Counter_Value'],
                        int(r['End_Timestamp']) - int(r['Start_Timestamp'])])


def active(counter, kern):
    """mean counter value (bytes; the counters are in KB) over the launches that did work (>= half the longest)"""
    rows = [r for r in csv.DictReader(open('profiles/%s_%s_%s.csv' % (rnd, CFG.lower(), counter))) if r['Kernel'].startswith(kern)]
    if not rows:
        return 0.0, 0, 0
    mx = max(int(r['Duration_ns']) for r in rows)
    a = [float(r['Counter_Value']) for r in rows if int(r['Duration_ns']) >= 0.5 * mx]
    return sum(a) / len(a) * 1024.0, len(a), len(rows)


import os
prev = {}
if os.path.exists('profiles/traffic.json'):
    prev = json.load(open('profiles/traffic.json'))
kernels = sorted({r['Kernel'].split('<')[0] for r in csv.DictReader(open('profiles/%s_%s_FETCH_SIZE.csv' % (rnd, CFG.lower())))})
if CFG != 'C3':
    # power-law map: one SpMV = kr_kb_tails + kr_kb_tcd2 + kr_kb_hub
    tot, det = 0.0, {}
    for k in ('kr_kb_tails', 'kr_kb_tcd2', 'kr_kb_hub'):
        if k not in kernels:
            continue
        f_, na, nt = active('FETCH_SIZE', k)
        w_, _, _ = active('WRITE_SIZE', k)
        det[k] = {'FETCH_SIZE_raw_bytes_per_launch': f_, 'WRITE_SIZE_raw_bytes_per_launch': w_}
        tot += 2.0 * f_ + w_
    out = dict(prev)
    out[CFG] = dict(det, kernel='kr_kb_tails + kr_kb_tcd2' + (' + kr_kb_hub' if 'kr_kb_hub' in det else '') + ' (one SpMV)',
                    hbm_bytes_per_launch=tot, launches_counted=na, launches_total=nt,
                    algorithmic_bytes_per_launch=12.0 * NNZ + 20.0 * N,
                    correction='FETCH_SIZE doubled (128-byte fabric requests tallied at 64 bytes), WRITE_SIZE as reported: the '
                               'correction calibrated on C3 in the same round (see C3)',
                    source='profiles/%s_%s_FETCH_SIZE.csv, profiles/%s_%s_WRITE_SIZE.csv (rocprofv3 --pmc, one counter per pass, '
                           'bench.py --config %s --steps 1 --warmup 0)' % (rnd, CFG.lower(), rnd, CFG.lower(), CFG))
    json.dump(out, open('profiles/traffic.json', 'w'), indent=1)
    print(json.dumps(out[CFG], indent=1))
    sys.exit(0)
fp, _, _ = active('FETCH_SIZE', 'kr_prep_counts')
wp, _, _ = active('WRITE_SIZE', 'kr_prep_counts')
fq, _, _ = active('FETCH_SIZE', 'k_quantise')
wq, _, _ = active('WRITE_SIZE', 'k_quantise')
correction = ('MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies the 128-byte fabric requests of coalesced reads at 64 '
              'bytes, so it is doubled; WRITE_SIZE is exact. Checked in the same two passes on two pure streaming kernels of '
              'known size: kr_prep_counts reads 8*nnz+8*n = %.0f MB and reports %.0f MB (x%.3f), writes 2*nnz+8*n = %.0f MB '
              'and reports %.0f MB%s.' % ((8 * NNZ + 8 * N) / 1e6, fp / 1e6, fp / (8 * NNZ + 8 * N), (2 * NNZ + 8 * N) / 1e6, wp / 1e6,
                                         '; k_quantise reads and writes 8 bytes per sub-map entry and reports %.0f MB fetched / %.0f MB '
                                         'written (x%.3f)' % (fq / 1e6, wq / 1e6, fq / wq) if wq else
                                         ' (the second calibration kernel of rounds 1-2, k_quantise, no longer exists: its work moved into k_lv_init)'))
source = ('profiles/%s_c3_FETCH_SIZE.csv, profiles/%s_c3_WRITE_SIZE.csv (rocprofv3 --pmc, one counter per pass, '
          'bench.py --config C3 --steps 1 --warmup 0)' % (rnd, rnd))
out = dict(prev)
if 'kr_kb_tails' in kernels:
    # one SpMV = kr_kb_tails + kr_kb_tcd2: the sum of the two launches
    ft, na, nt = active('FETCH_SIZE', 'kr_kb_tails')
    wt, _, _ = active('WRITE_SIZE', 'kr_kb_tails')
    fd, _, _ = active('FETCH_SIZE', 'kr_kb_tcd2')
    wd, _, _ = active('WRITE_SIZE', 'kr_kb_tcd2')
    if 'C3' in prev and prev['C3'].get('kernel', '').startswith('kr_kb_vec'):
        out['C3_kr_kb_vec'] = prev['C3']            # the CSR row kernel's record (earlier passes), cited by DESIGN.md
    out['C3'] = {
        'kernel': 'kr_kb_tails + kr_kb_tcd2 (one SpMV: tails in column order in fixed point, dictionary rows + fused KR epilogue)',
        'kr_kb_tails': {'FETCH_SIZE_raw_bytes_per_launch': ft, 'WRITE_SIZE_raw_bytes_per_launch': wt},
        'kr_kb_tcd2': {'FETCH_SIZE_raw_bytes_per_launch': fd, 'WRITE_SIZE_raw_bytes_per_launch': wd},
        'hbm_bytes_per_launch': 2.0 * (ft + fd) + wt + wd,
        'algorithmic_bytes_per_launch': 12.0 * NNZ + 20.0 * N,      # bench.py uses a record only for the map it was taken on
        'launches_counted': na, 'launches_total': nt, 'correction': correction, 'source': source}
    # the Louvain neighbour scan beside it: raw FETCH_SIZE of the launches of the first three sweeps of level 0 (unpruned: every row
    # is read), in launch order -- k_lv_first_g, then eleven launches of k_lv_decide_g
    lv = [r for r in csv.DictReader(open('profiles/%s_c3_FETCH_SIZE.csv' % rnd)) if r['Kernel'].startswith('k_lv_decide_g') or r['Kernel'].startswith('k_lv_first_g')]
    lv.sort(key=lambda r: int(r['Dispatch_Id']))
    if lv:
        out['C3_k_lv_scan'] = {
            'kernel': 'k_lv_first_g + k_lv_decide_g (Louvain delta-Q neighbour scan, one launch per sub-round)',
            'FETCH_SIZE_raw_bytes_first_12_launches': [float(r['Counter_Value']) * 1024.0 for r in lv[:12]],
            'duration_ns_first_12_launches': [int(r['Duration_ns']) for r in lv[:12]],
            'algorithmic_bytes_per_launch': (16.0 * NNZ + 28.0 * N) / 4.0,
            'note': 'raw counter values (FETCH_SIZE tallies 128-byte fabric requests at 64 bytes: double them for bytes); round 3: '
                    '2.18 / 2.02 / 1.80 GB raw in the three general sub-rounds of sweep 1',
            'source': source}
    json.dump(out, open('profiles/traffic.json', 'w'), indent=1)
    print(json.dumps(out['C3'], indent=1))
    if lv:
        print(json.dumps(out['C3_k_lv_scan'], indent=1))
    sys.exit(0)
fv, na, nt = active('FETCH_SIZE', 'kr_kb_vec')
wv, _, _ = active('WRITE_SIZE', 'kr_kb_vec')
fp, _, _ = active('FETCH_SIZE', 'kr_prep_counts')
wp, _, _ = active('WRITE_SIZE', 'kr_prep_counts')
fq, _, _ = active('FETCH_SIZE', 'k_quantise')
wq, _, _ = active('WRITE_SIZE', 'k_quantise')
out = {'C3': {
    'kernel': 'kr_kb_vec<8,8,uint16> (CSR SpMV on raw 16-bit counts, 4 entries per lane and load, fused KR epilogue)',
    'FETCH_SIZE_raw_bytes_per_launch': fv, 'WRITE_SIZE_raw_bytes_per_launch': wv,
    'hbm_bytes_per_launch': 2.0 * fv + wv,
    'launches_counted': na, 'launches_total': nt,
    'correction': 'MI355X_MICROARCH.md (HBM section): FETCH_SIZE tallies the 128-byte fabric requests of coalesced reads at 64 '
                  'bytes, so it is doubled; WRITE_SIZE is exact. Checked in the same two passes on two pure streaming kernels of '
                  'known size: kr_prep_counts reads 8*nnz+8*n = %.0f MB and reports %.0f MB (x%.3f), writes 2*nnz+8*n = %.0f MB '
                  'and reports %.0f MB; k_quantise reads and writes 8 bytes per sub-map entry and reports %.0f MB fetched / '
                  '%.0f MB written (x%.3f).' % ((8 * NNZ + 8 * N) / 1e6, fp / 1e6, fp / (8 * NNZ + 8 * N), (2 * NNZ + 8 * N) / 1e6,
                                                wp / 1e6, fq / 1e6, wq / 1e6, fq / wq),
    'reading': 'streams by construction 6*nnz + 36*n = %.0f MB; the rest of the corrected %.0f MB are 128-byte operand lines '
               'fetched for gathers that miss the XCD L2 (the 8 MB operand vector against 4 MB of L2): %.1f M line fetches for '
               '%.0f M gathered operands' % ((6 * NNZ + 36 * N) / 1e6, (2 * fv + wv) / 1e6,
                                             (2 * fv - (6 * NNZ + 28 * N)) / 128 / 1e6, NNZ / 1e6),
    'source': 'profiles/%s_c3_FETCH_SIZE.csv, profiles/%s_c3_WRITE_SIZE.csv (rocprofv3 --pmc, one counter per pass, '
              'bench.py --config C3 --steps 1 --warmup 0)' % (rnd, rnd)}}
json.dump(out, open('profiles/traffic.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
