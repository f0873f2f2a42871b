"""Build libcat_hip.so (gfx950) in-tree with hipcc.  No cmake / torch extension machinery: the library is a
plain C-ABI shared object (include/cat_hip.h) that ctypes loads."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libcat_hip.so')
SOURCES = ['runtime.hip', 'conv_igemm.hip', 'conv_smallco.hip', 'conv_pk.hip', 'conv_ksum.hip', 'conv_q.hip', 'conv_split.hip', 'conv_twgrad.hip', 'conv_pwgrad.hip', 'block_norm.hip', 'dwconv.hip', 'norm.hip', 'elementwise.hip', 'ka_loss.hip', 'spade.hip', 'eval_ops.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


DIAG_LIB = os.path.join(LIBDIR, 'libcat_hip_diag.so')


def build(force=False, verbose=True, diag=False):
    """diag=True: the DIAGNOSTIC build (-DCAT_DIAG, common.h kDiag) into lib/libcat_hip_diag.so -- per-phase shader clocks and the
    ablation switches that make results wrong by design; only tools/debug/* load it (CAT_LIB=diag).  Never built by build() / the driver."""
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, 'common.h'), os.path.join(HERE, '..', 'include', 'cat_hip.h')]
    hipcc = _hipcc()
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace('.hip', '.diag.o' if diag else '.o'))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + (['-DCAT_DIAG'] if diag else []) + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    lib = DIAG_LIB if diag else LIB
    if force or jobs or _stale(lib, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, diag='--diag' in sys.argv))
