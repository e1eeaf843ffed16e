#!/usr/bin/env python3
"""
The DPP read hazard of gfx9 / CDNA: a DPP instruction must not read a VGPR a
VALU instruction wrote less than 2 wait states before it (nor follow a
`v_cmpx` / other VALU write of EXEC by less than 5).  The library issues its
DPP operations (`v_add_u32_dpp`, `v_mov_b64_dpp` with `row_newbcast`:
csrc/spmm_strip.h `strip_addr` / `strip_weight`, used by families 8 and 11)
through inline asm, which LLVM's hazard recogniser does not look into -- so
nothing but register allocation keeps the distance.  This tool compiles
remap_spmm.hip to gfx950 assembly and checks every DPP site; a compiler bump
that moves a writer next to one fails `tools/round_check.sh` (and this
script) instead of silently computing garbage.

    python tools/dpp_hazard_scan.py [file.s]      exit code 1 on a hazard
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pyremap_amd import _build  # noqa: E402

REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def wait_states(ins):
    """Wait states an instruction puts between its neighbours."""
    if ins.startswith('s_nop'):
        return int(ins.split()[1], 0) + 1
    return 1


def scan(lines):
    """[(line number, dpp instruction, offending instruction)]"""
    ins = []   # (line number, text) of instructions, labels as None
    for n, raw in enumerate(lines, 1):
        t = raw.split(';')[0].strip()
        if not t or t.startswith('.') or t.startswith('//'):
            continue
        if t.endswith(':'):
            ins.append((n, None))       # a label: another path may join
            continue
        ins.append((n, t))
    bad = []
    sites = 0
    for k, (n, t) in enumerate(ins):
        if t is None or '_dpp' not in t.split()[0]:
            continue
        sites += 1
        ops = t.split(None, 1)[1]
        parts = [p.strip() for p in ops.split(',')]
        # sources: everything behind the destination, modifiers aside
        src = regs(','.join(p.split(' row_')[0] for p in parts[1:]))
        dist = 0
        j = k - 1
        while j >= 0 and dist < 5:
            m, u = ins[j]
            if u is None:
                j -= 1          # (conservative: look through the label)
                continue
            op = u.split()[0]
            if op.startswith('v_') and dist < 2:
                dst = regs(u.split(None, 1)[1].split(',')[0]) \
                    if ' ' in u else set()
                if dst & src:
                    bad.append((n, t, f'line {m}: {u}'))
            if op.startswith('v_cmpx') or (
                    op.startswith('v_') and re.search(r'\bexec\b',
                                                     u.split(',')[0])):
                bad.append((n, t, f'line {m}: {u} (EXEC)'))
            dist += wait_states(u)
            j -= 1
    return sites, bad


def main():
    if len(sys.argv) > 1:
        lines = open(sys.argv[1]).read().splitlines()
    else:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, 'spmm.s')
            subprocess.run(
                [_build.find_hipcc(), '-O3', '-std=c++17',
                 f'--offload-arch={_build.ARCH}', '-ffp-contract=off',
                 '-fPIC', f'-I{_build.INCLUDE}', f'-I{_build.CSRC}', '-S',
                 '--cuda-device-only', '-o', out,
                 os.path.join(_build.CSRC, 'remap_spmm.hip')],
                check=True, stderr=subprocess.DEVNULL)
            lines = open(out).read().splitlines()
    sites, bad = scan(lines)
    print(f'{sites} DPP sites, {len(bad)} within the hazard window of a '
          f'VALU writer')
    for n, t, why in bad[:20]:
        print(f'  line {n}: {t}\n      <- {why}')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
