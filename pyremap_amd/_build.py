"""
Builds ``libremap_hip.so`` (the C-ABI HIP library, ``include/remap_hip.h``)
in-tree with hipcc for gfx950.  Used by ``__graft_entry__.build()`` and, as a
convenience, on first use when the library is missing but hipcc is present.
"""
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, 'csrc')
INCLUDE = os.path.join(_REPO, 'include')
LIB_DIR = os.path.join(_PKG, '_lib')
LIB_PATH = os.environ.get('REMAP_HIP_LIB') or \
    os.path.join(LIB_DIR, 'libremap_hip.so')
SOURCES = ['remap_spmm.hip', 'remap_csr.hip', 'remap_schedule.hip',
           'remap_plan.hip']
ARCH = 'gfx950'


def find_hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'),
                 '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    return None


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES]


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = sources() + [os.path.join(INCLUDE, 'remap_hip.h')] + \
        [os.path.join(CSRC, f) for f in os.listdir(CSRC)
         if f.endswith('.h')]
    return any(os.path.getmtime(d) > built for d in deps)


def build_library(force=False, verbose=False):
    """
    ``hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared``.

    ``-ffp-contract=off`` keeps the multiply and the add of the accumulation
    separate, which is what makes the default mode bit-identical to scipy's
    ``csr_matvecs`` (the REMAP_FLAG_FMA kernels call fma explicitly).
    """
    if not force and not is_stale():
        return LIB_PATH
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError('hipcc not found: cannot build libremap_hip.so')
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc, '-O3', '-std=c++17', f'--offload-arch={ARCH}',
           '-ffp-contract=off', '-fPIC', '-shared',
           f'-I{INCLUDE}', f'-I{CSRC}', '-o', LIB_PATH] + sources()
    if verbose:
        print(' '.join(cmd))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f'hipcc failed:\n{proc.stdout}')
    return LIB_PATH
