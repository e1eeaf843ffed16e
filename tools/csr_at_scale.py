#!/usr/bin/env python3
"""
`remap_csr_from_coo` against the oracle on a BASELINE mapping too large for
the test suite's time budget (GPU box):

    python tools/csr_at_scale.py config5      # 95 M triplets -> 78 M entries

Row pointers and column indices equal, every weight bit for bit (the
oracle: oracle/remap_oracle.c::oracle_coo_to_csr, the C restatement of
scipy's coo -> csr, remap_numpy.py:134-137).  Prints one line for
profiles/r06_analysis.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import oracle  # noqa: E402
from pyremap_amd import engine, synthetic  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'config5'
dev = torch.device('cuda', 0)
m = synthetic.make_config(name, device=dev, locality='mesh')
t0 = time.perf_counter()
plan = engine.RemapPlan.from_triplets(m.row, m.col, m.S, m.frac_b, m.n_a,
                                      m.n_b, index_base=1, device=dev)
torch.cuda.synchronize()
t_dev = time.perf_counter() - t0
rowptr, col, val = plan.to_host_csr()
mm = m.numpy()
t0 = time.perf_counter()
ref = oracle.coo_to_csr(mm['row'] - 1, mm['col'] - 1, mm['S'], m.n_b, m.n_a)
t_cpu = time.perf_counter() - t0
ok = (ref.nnz == plan.nnz and np.array_equal(rowptr, ref.indptr) and
      np.array_equal(col, ref.indices) and
      np.array_equal(val.view(np.int64), ref.data.view(np.int64)))
print(f'{name}: {mm["row"].size} triplets -> {plan.nnz} entries '
      f'({mm["row"].size - plan.nnz} duplicates merged); device '
      f'{t_dev:.2f} s, oracle {t_cpu:.1f} s; indptr / indices / data '
      f'bitwise equal: {ok}')
sys.exit(0 if ok else 1)
