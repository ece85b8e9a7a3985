"""Resource usage (VGPRs, scratch, LDS, occupancy) of every kernel of libb3c: hipcc -Rpass-analysis=kernel-resource-usage.

    python tools/resource_usage.py [file.hip ...] [--scratch-only] [--all]

tools/profile_round.sh writes its output to profiles/<round>_resource_usage.txt, so that a stack object that lands in
scratch (round 2: a 128-byte KRState copy in every lane of kr_kb_tcd2) is seen in the round it appears.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'bin3c_amd', 'csrc')


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    res = []
    for o in out:
        o = o.replace('(anonymous namespace)::', '')
        o = re.sub(r'^void ', '', o)
        depth, cut = 0, len(o)
        for i, ch in enumerate(o):              # cut the argument list: the first '(' outside template brackets
            if ch == '<':
                depth += 1
            elif ch == '>':
                depth -= 1
            elif ch == '(' and depth == 0:
                cut = i
                break
        res.append(o[:cut])
    return res


def usage(src):
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', os.path.join(ROOT, 'include'), '-I', CSRC,
           '-Wno-unused-result', '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null']
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split('\n'):
        m = re.search(r'remark:\s+([A-Za-z][A-Za-z \[\]/]*?): (\S+)', line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == 'Function Name':
            cur = {'name': v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    names = demangle([r['name'] for r in rows])
    for r, nm in zip(rows, names):
        r['name'] = nm
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    scratch_only = '--scratch-only' in sys.argv
    files = args or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    print('{:<78} {:>5} {:>5} {:>7} {:>7} {:>4}'.format('kernel', 'VGPR', 'AGPR', 'scratch', 'LDS', 'occ'))
    bad = 0
    lib_only = '--all' not in sys.argv      # libb3c's own kernels; --all adds the rocPRIM / hipCUB template instantiations (sorts, scans)
    skipped = 0
    for f in files:
        for r in usage(f):
            sc = int(r.get('ScratchSize [bytes/lane]', '0'))
            third_party = r['name'].startswith(('rocprim::', 'hipcub::', 'thrust::')) or '::detail::' in r['name']
            if lib_only and third_party:
                skipped += 1
                continue
            bad += sc > 0
            if scratch_only and sc == 0:
                continue
            print('{:<78} {:>5} {:>5} {:>7} {:>7} {:>4}'.format(r['name'][:78], r.get('VGPRs', '?'), r.get('AGPRs', '?'), sc,
                                                              r.get('LDS Size [bytes/block]', '?'), r.get('Occupancy [waves/SIMD]', '?')))
    print('# kernels with scratch: {}{}'.format(bad, ' ({} rocPRIM / hipCUB instantiations not listed: --all)'.format(skipped) if lib_only else ''))


if __name__ == '__main__':
    main()
