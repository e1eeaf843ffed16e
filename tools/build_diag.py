#!/usr/bin/env python3
"""
Build the DIAGNOSTIC variants of libremap_hip.so (never the product):

    python tools/build_diag.py diag     -DREMAP_DIAG    tune[6] (no stores /
                                        gather from 1024 rows) and tune[7]
                                        (LDS occupancy throttle) switches
    python tools/build_diag.py stamps   -DREMAP_STAMPS  in-kernel s_memtime
                                        stamps (tools/stamps.py)
    python tools/build_diag.py plain    -DREMAP_PLAIN_STORES  write-back instead of
                                        non-temporal Y stores (an A/B)
    python tools/build_diag.py xaux2    -DREMAP_X_AUX=2  cache policy of the X
                                        loads (1 sc0, 2 nt, 16 sc1, 17, 3 ...)
    python tools/build_diag.py ceiling  tools/hbm_ceiling.hip -> executable

Outputs go to tools/_build/ (git-ignored; they still travel to the GPU box).
Use with REMAP_HIP_LIB=tools/_build/libremap_hip_<variant>.so.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pyremap_amd import _build  # noqa: E402

OUT = os.path.join(ROOT, 'tools', '_build')


def run(cmd):
    print(' '.join(cmd))
    subprocess.run(cmd, check=True)


def main():
    what = sys.argv[1:] or ['diag', 'stamps', 'ceiling']
    os.makedirs(OUT, exist_ok=True)
    hipcc = _build.find_hipcc()
    for w in what:
        if w == 'ceiling':
            run([hipcc, '-O3', '-std=c++17', f'--offload-arch={_build.ARCH}',
                 '-o', os.path.join(OUT, 'hbm_ceiling'),
                 os.path.join(ROOT, 'tools', 'hbm_ceiling.hip')])
            continue
        if w.startswith('xaux'):   # cache policy of the X loads (A/B)
            define = f'-DREMAP_X_AUX={int(w[4:])}'
        else:
            define = {'diag': '-DREMAP_DIAG', 'stamps': '-DREMAP_STAMPS',
                      'plain': '-DREMAP_PLAIN_STORES'}[w]
        run([hipcc, '-O3', '-std=c++17', f'--offload-arch={_build.ARCH}',
             '-ffp-contract=off', '-fPIC', '-shared', define,
             f'-I{_build.INCLUDE}', f'-I{_build.CSRC}', '-o',
             os.path.join(OUT, f'libremap_hip_{w}.so')] + _build.sources())


if __name__ == '__main__':
    main()
