import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'):
        continue
    d = json.loads(line)
    c = d['config']
    print('%s  %.2f ms/step  %.3g edges/s' % (c['workload'][:3], d['ms_per_step'], d['value']))
    print('   stage_ms', {k: round(v, 2) for k, v in c['stage_ms'].items()})
    print('   kernel_ms', {k: round(v, 2) for k, v in c['kernel_ms'].items()}, 'spmv/step', c['kr_spmv_per_step'], 'sweeps', c['louvain_sweeps_per_step'])
    r = d['roofline']
    print('   roofline %s: %.0f GB/s (%.1f%%) avg %.3f ms, copy %s GB/s' % (r['kernel'][:12], r['achieved'], 100 * r['frac'], r['avg_launch_ms'], r.get('copy_gbs') and round(r['copy_gbs'])))
    o = d.get('roofline_other')
    if o:
        print('   other    %s: %.0f GB/s (%.1f%%) avg %.3f ms' % (o['kernel'][:12], o['achieved'], 100 * o['frac'], o['avg_launch_ms']))
    if 'cpu_baseline' in d:
        print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:100])
    sv = d.get('variant_workload')
    if sv:
        print('   VARIANT leg: %.2f ms/step  %.3g edges/s  stage_ms %s  spmv/step %s sweeps %s levels %s' % (
            sv['ms_per_step'], sv['value'], {k: round(v, 2) for k, v in sv['stage_ms'].items()}, sv['kr_spmv_per_step'],
            sv['louvain_sweeps_per_step'], sv['louvain_levels']))
