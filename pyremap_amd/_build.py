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
           'remap_plan.hip', 'remap_shard.hip']
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
    deps = sources() + [os.path.join(INCLUDE, 'remap_hip.h'),
                        os.path.join(CSRC, 'libremap_hip.map')] + \
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
    # one hipcc per source, side by side (remap_spmm.hip alone takes over a
    # minute: hundreds of kernel instantiations), then one link
    obj_dir = os.path.join(LIB_DIR, 'obj')
    os.makedirs(obj_dir, exist_ok=True)
    # -fvisibility=hidden: the library's dynamic symbols are the REMAP_API
    # entry points of include/remap_hip.h and nothing else (no mangled
    # remap::... helpers, no kernel stubs)
    common = [hipcc, '-O3', '-std=c++17', f'--offload-arch={ARCH}',
              '-ffp-contract=off', '-fPIC', '-fvisibility=hidden',
              f'-I{INCLUDE}', f'-I{CSRC}']
    headers = [os.path.join(INCLUDE, 'remap_hip.h')] + \
        [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    newest_header = max(os.path.getmtime(h) for h in headers)
    jobs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + '.o')
        fresh = os.path.exists(obj) and not force and \
            os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header)
        if fresh:
            continue
        cmd = common + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        jobs.append((src, subprocess.Popen(
            cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
            text=True)))
    failed = []
    for src, proc in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            failed.append(f'{src}:\n{out}')
    if failed:
        raise RuntimeError('hipcc failed:\n' + '\n'.join(failed))
    objs = [os.path.join(obj_dir, os.path.basename(src) + '.o')
            for src in sources()]
    cmd = [hipcc, f'--offload-arch={ARCH}', '-fPIC', '-shared',
           '-Wl,--version-script=' + os.path.join(CSRC, 'libremap_hip.map'),
           '-o', LIB_PATH] + objs
    if verbose:
        print(' '.join(cmd))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f'hipcc (link) failed:\n{proc.stdout}')
    return LIB_PATH
