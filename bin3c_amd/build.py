"""Compile libb3c.so (HIP, gfx950) in-tree.  hipcc cross-compiles without a GPU present."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIB = os.path.join(HERE, 'libb3c.so')
SOURCES = ['api.hip', 'csr_ops.hip', 'kr.hip', 'louvain.hip', 'reorder.hip', 'accumulate.hip', 'downsample.hip', 'dist.hip']


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, 'b3c.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> bin3c_amd/libb3c.so; returns the path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, 'build', src.replace('.hip', '.o'))
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', INCLUDE, '-I', CSRC,
               '-Wno-unused-result', '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on {}'.format(src))
        if verbose and out:
            sys.stdout.write(out.decode())
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl']
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build_lib(force='--force' in sys.argv, verbose=True))
