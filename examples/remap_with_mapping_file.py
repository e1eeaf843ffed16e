#!/usr/bin/env python
"""
Apply an existing mapping file (ESMF_RegridWeightGen / mbtempest / ncremap
output; NetCDF-3 or NetCDF-4) to a file on an MPAS mesh -- what
``Remapper.ncremap`` / ``remap_numpy`` of pyremap do -- on the GPU.

    python examples/remap_with_mapping_file.py -m map_oEC60to30v3_to_0.5x0.5degree_aave.nc \
        -i timeSeriesStatsMonthly.nc -o out.nc --dlon 0.5 --dlat 0.5 \
        [-v timeMonthly_avg_activeTracers_temperature ...] [--renormalize 0.01]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pyremap_amd import (  # noqa: E402
    MpasCellMeshDescriptor,
    Remapper,
    get_lat_lon_descriptor,
)
from pyremap_amd.io.mapfile import read_mapping  # noqa: E402


def main(argv=None):
    parser = argparse.ArgumentParser(
        description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    parser.add_argument('-m', dest='map_filename', required=True)
    parser.add_argument('-i', dest='in_filename', required=True)
    parser.add_argument('-o', dest='out_filename', required=True)
    parser.add_argument('--dlon', type=float, required=True)
    parser.add_argument('--dlat', type=float, required=True)
    parser.add_argument('-v', dest='variables', nargs='*', default=None)
    parser.add_argument('--renormalize', type=float, default=None)
    parser.add_argument('--mesh-name', default='mpas')
    args = parser.parse_args(argv)

    n_cells = read_mapping(args.map_filename).n_a
    remapper = Remapper(map_filename=args.map_filename)
    # only the mesh size is needed to apply weights; pass a mesh file
    # (MpasCellMeshDescriptor(filename)) to carry lat/lon coordinates along
    remapper.src_descriptor = MpasCellMeshDescriptor(
        mesh_name=args.mesh_name, size=n_cells)
    remapper.dst_descriptor = get_lat_lon_descriptor(args.dlon, args.dlat)
    remapper.ncremap(args.in_filename, args.out_filename,
                     variable_list=args.variables, overwrite=True,
                     renormalize=args.renormalize, replace_mpas_fill=True)
    print(f'{args.in_filename} -> {args.out_filename} with '
          f'{args.map_filename} ({remapper.schedule["family"]} kernels)')
    return remapper


if __name__ == '__main__':
    main()
