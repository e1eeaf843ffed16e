#!/usr/bin/env python3
"""
One mapping, several MI355X: destination rows sharded over the GPUs, each
shard reading only the source rows it references (`X[unique(col[shard])]`),
no reduction collective -- the two ways `pyremap_amd` offers it.

    # one process, N GPUs (peer copies over xGMI):
    python examples/multi_gpu.py

    # one process per GPU (RCCL over xGMI), the same calls as collectives:
    torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 8 \
        examples/multi_gpu.py

On a one-GPU box the first form lists that GPU twice (results are the same,
bit for bit, whatever the number of devices).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import pyremap_amd as pyremap  # noqa: E402
from pyremap_amd import synthetic  # noqa: E402


def main():
    under_torchrun = 'RANK' in os.environ and int(os.environ['WORLD_SIZE']) > 1
    if under_torchrun:
        import torch.distributed as dist
        local = int(os.environ['LOCAL_RANK']) % torch.cuda.device_count()
        torch.cuda.set_device(local)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group(
            'nccl', device_id=torch.device('cuda', local))
        rank = dist.get_rank()
    else:
        rank = 0

    # a mapping file as ESMF would write it: 20 000-cell mesh (numbered as
    # an MPAS mesh numbers its cells) -> 1-degree grid, conservative
    m = synthetic.conservative_map(20000, (180, 360), 2, 6, seed=1,
                                   locality='mesh')
    map_path = f'/tmp/map_example_{os.getpid()}.nc'
    m.save(map_path)
    rng = np.random.default_rng(0)
    src = pyremap.MpasCellMeshDescriptor(mesh_name='toy', lat=rng.random(
        m.n_a), lon=rng.random(m.n_a))
    dst = pyremap.get_lat_lon_descriptor(dlon=1.0, dlat=1.0)
    ds = pyremap.Dataset({
        'temperature': pyremap.DataArray(
            np.where(rng.random((4, m.n_a, 32)) < 0.1, np.nan,
                     rng.standard_normal((4, m.n_a, 32))),
            dims=('Time', 'nCells', 'nVertLevels')),
        'ssh': pyremap.DataArray(rng.standard_normal((4, m.n_a)),
                                 dims=('Time', 'nCells'))})

    one = pyremap.Remapper(map_filename=map_path, src_descriptor=src,
                           dst_descriptor=dst)
    reference = one.remap_numpy(ds, renormalization_threshold=0.01)

    if under_torchrun:
        many = pyremap.Remapper(map_filename=map_path, src_descriptor=src,
                                dst_descriptor=dst)
        many.use_process_group(src=0)          # rank 0's data is remapped
        how = f'{int(os.environ["WORLD_SIZE"])} ranks'
    else:
        n = max(torch.cuda.device_count(), 2)
        devices = [f'cuda:{i % torch.cuda.device_count()}' for i in range(n)]
        many = pyremap.Remapper(map_filename=map_path, src_descriptor=src,
                                dst_descriptor=dst, devices=devices)
        how = f'devices {devices}'
    out = many.remap_numpy(ds, renormalization_threshold=0.01)
    for name in reference.data_vars:
        a, b = reference[name].values, out[name].values
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.array_equal(a[~np.isnan(a)].view(np.int64),
                              b[~np.isnan(b)].view(np.int64)), name
    if rank == 0:
        print(f'{how}: {list(out.data_vars)} identical to one GPU, '
              f'bit for bit; shards hold '
              f'{100 * many._matrix.packed_fraction():.0f} % of a broadcast '
              f'of the source rows each')
        os.remove(map_path)
    if under_torchrun:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
