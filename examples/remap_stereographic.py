#!/usr/bin/env python
"""
Remap every variable of a file on an Antarctic stereographic grid to a grid of
the same extent and another resolution -- the job of pyremap's
``examples/remap_stereographic.py``, without ESMF: the weights are generated
analytically (``map_tool='analytic'``) and applied on the GPU.

    python examples/remap_stereographic.py -i in.nc -o out.nc -r 20 \
        [-m bilinear|neareststod|conserve]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pyremap_amd import ProjectionGridDescriptor, Remapper  # noqa: E402
from pyremap_amd.io.netcdf import open_dataset  # noqa: E402
from pyremap_amd.polar import (  # noqa: E402
    get_antarctic_stereographic_projection,
)


def grid_name(x, y):
    dx = int((x[1] - x[0]) / 1000.0)
    lx = int((x[-1] - x[0]) / 1000.0)
    ly = int((y[-1] - y[0]) / 1000.0)
    return f'{lx}x{ly}km_{dx}km_Antarctic_stereo'


def main(argv=None):
    parser = argparse.ArgumentParser(
        description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    parser.add_argument('-i', dest='in_filename', required=True)
    parser.add_argument('-o', dest='out_filename', required=True)
    parser.add_argument('-r', dest='resolution', required=True, type=float,
                        help='output resolution in km')
    parser.add_argument('-m', dest='method', default='bilinear',
                        choices=['bilinear', 'neareststod', 'conserve'])
    args = parser.parse_args(argv)

    ds_in = open_dataset(args.in_filename)
    x = np.asarray(ds_in['x'].values, dtype=float)
    y = np.asarray(ds_in['y'].values, dtype=float)
    projection = get_antarctic_stereographic_projection()

    remapper = Remapper(method=args.method, map_tool='analytic')
    remapper.src_descriptor = ProjectionGridDescriptor.create(
        projection, x, y, grid_name(x, y))
    res = args.resolution * 1e3
    x_out = x[0] + res * np.arange(int((x[-1] - x[0]) / res + 0.5) + 1)
    y_out = y[0] + res * np.arange(int((y[-1] - y[0]) / res + 0.5) + 1)
    remapper.dst_descriptor = ProjectionGridDescriptor.create(
        projection, x_out, y_out, grid_name(x_out, y_out))

    remapper.build_map()            # map_<src>_to_<dst>_analytic<method>.nc
    remapper.ncremap(args.in_filename, args.out_filename, overwrite=True,
                     renormalize=0.01)
    print(f'{remapper.map_filename}: {len(x)} x {len(y)} -> '
          f'{len(x_out)} x {len(y_out)} cells, kernel schedule '
          f'{remapper.schedule["family"]}; wrote {args.out_filename}')
    return remapper


if __name__ == '__main__':
    main()
