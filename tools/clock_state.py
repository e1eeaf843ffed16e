#!/usr/bin/env python3
"""
Which clock do the kernels run at?  A series of launches of a workload, the
shader clock measured right behind every series (`remap_clock_probe`: one
wave, 100 x Delta s_memtime / Delta s_memrealtime MHz), after idle periods of
different lengths -- config 4's LDS patch kernel is instruction-issue-bound
and its time follows the clock (5.2 or 6.4 ms); config 3's row-group kernel
waits on memory and does not.  GPU box only.

    python tools/clock_state.py [config4 config3 ...]

With the stamps build (`python tools/build_diag.py stamps`, then
REMAP_HIP_LIB=tools/_build/libremap_hip_stamps.so) and --in-kernel the clock
is ALSO measured inside the kernels: one wave per workgroup reads s_memtime /
s_memrealtime at its first and last instruction (csrc/spmm_device.h:
REMAP_CLOCK_BEGIN / _END); MHz = 100 x sum(cycles) / sum(ticks) over the
series' workgroups.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyremap_amd import engine, synthetic  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    in_kernel = '--in-kernel' in sys.argv
    names = [a for a in sys.argv[1:] if not a.startswith('--')]
    for name in (names or ['config4', 'config3']):
        cfg = synthetic.CONFIGS[name]
        m = synthetic.make_config(name, device=dev, locality='mesh')
        plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b,
                                              m.n_a, m.n_b, index_base=1,
                                              device=dev)
        sched = plan.auto_schedule(m.dst_dims)
        K = cfg['K']
        x = torch.randn((m.n_a, K), device=dev, dtype=torch.float64)
        y = torch.empty((m.n_b, K), device=dev, dtype=torch.float64)
        # (stamps build: the kernels add their cycle / tick counts to slots
        # 6 and 7 of the buffer handed over as the byte mask, which that
        # build does not write)
        stamps = torch.zeros((m.n_b, K), device=dev, dtype=torch.uint8) \
            if in_kernel else None

        def series(n):
            if in_kernel:
                stamps[0, :64].zero_()
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                engine.apply_strided(
                    plan, x, y, n_batch=1, k_inner=K, x_row_stride=K,
                    x_batch_stride=0, y_row_stride=K, y_batch_stride=0,
                    mode=engine.MODE_FRACB, mask_out=stamps)
            b.record()
            mhz = engine.clock_probe(dev)
            torch.cuda.synchronize()
            inside = None
            if in_kernel:
                c = stamps[0, :64].view(torch.int64).cpu()
                inside = 100.0 * int(c[6]) / max(int(c[7]), 1)
            return a.elapsed_time(b) / n, mhz(), inside
        n = 12 if name in ('config4', 'config5', 'headline') else 100
        series(3)
        for idle in (0.0, 0.0, 0.05, 0.5, 2.0, 0.0, 0.0):
            time.sleep(idle)
            ms, mhz, inside = series(n)
            print(json.dumps(dict(
                workload=name, family=sched['family'], idle_before_s=idle,
                launches=n, ms_per_launch=round(ms, 4),
                clock_mhz_behind_the_series=round(mhz, 1),
                clock_mhz_inside_the_kernels=None if inside is None else
                round(inside, 1))), flush=True)
        del x, y, plan, stamps
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
