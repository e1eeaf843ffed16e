#!/usr/bin/env python3
"""
Register budget of every kernel in libremap_hip.so's device code: compile
remap_spmm.hip to gfx950 assembly and print .vgpr_count / .sgpr_count and the
waves per SIMD they allow (MI355X_MICROARCH.md, register files: 512 VGPRs per
lane per SIMD, granule 8).

    python tools/kernel_regs.py [filter] [-D...]

KERNEL_REGS=--fail-on-spill: exit 1 if any (listed) kernel spills VGPRs --
tools/round_check.sh runs it so (a spill is scratch traffic in a kernel the
dispatcher may pick).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pyremap_amd import _build  # noqa: E402


def demangle(names):
    for tool in ('c++filt', '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'):
        try:
            out = subprocess.run([tool] + names, capture_output=True,
                                 text=True, check=True).stdout
            return out.strip().split('\n')
        except (OSError, subprocess.CalledProcessError):
            continue
    return names


def main():
    args = sys.argv[1:]
    defs = [a for a in args if a.startswith('-D')]
    flt = [a for a in args if not a.startswith('-D')]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'spmm.s')
        subprocess.run(
            [_build.find_hipcc(), '-O3', '-std=c++17',
             f'--offload-arch={_build.ARCH}', '-ffp-contract=off', '-fPIC',
             f'-I{_build.INCLUDE}', f'-I{_build.CSRC}', '-S',
             '--cuda-device-only', '-o', out] + defs +
            [os.path.join(_build.CSRC, 'remap_spmm.hip')],
            check=True, stderr=subprocess.DEVNULL)
        s = open(out).read()
        if '--keep' in os.environ.get('KERNEL_REGS', ''):
            open('/tmp/spmm.s', 'w').write(s)
    names = re.findall(r'^\s+\.name:\s+(_Z\S+)$', s, re.M)
    vg = re.findall(r'^\s+\.vgpr_count:\s+(\d+)$', s, re.M)
    sg = re.findall(r'^\s+\.sgpr_count:\s+(\d+)$', s, re.M)
    sp = re.findall(r'^\s+\.vgpr_spill_count:\s+(\d+)$', s, re.M)
    spilled = []
    for n, v, g, x in zip(demangle(names), vg, sg, sp):
        n = n.replace('remap::(anonymous namespace)::', '')
        n = n.replace('void ', '').split('(')[0]
        if flt and not any(f in n for f in flt):
            continue
        alloc = (int(v) + 7) // 8 * 8
        waves = min(8, 512 // max(alloc, 8))
        print(f'{n:64s} vgpr {v:>4} sgpr {g:>4} spill {x:>3} '
              f'waves/SIMD {waves}')
        if int(x):
            spilled.append(n)
    if '--fail-on-spill' in os.environ.get('KERNEL_REGS', '') and spilled:
        print(f'FAIL: {len(spilled)} kernel(s) spill VGPRs: '
              f'{spilled[:4]}')
        sys.exit(1)


if __name__ == '__main__':
    main()
