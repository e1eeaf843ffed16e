#!/usr/bin/env python
"""
The workflow of pyremap's ``examples/make_mpas_to_lat_lon_mapping.py`` without
ESMF or NCO: build a bilinear mapping file from an MPAS mesh to a global
lat-lon grid (``map_tool='analytic'``: ESMF's weights, reproduced by
``pyremap_amd.weights``), then remap a file from the mesh file to file
(``ncremap``) and in memory (``remap_numpy``), both on the GPU.

    python examples/make_mpas_to_lat_lon_mapping.py --mesh ocean.QU.240km.nc \
        --mesh-name oQU240 -i timeSeriesStatsMonthly.nc --res 0.5 \
        [--type cell|edge|vertex] [-v timeMonthly_avg_ssh ...] [-o OUT_DIR]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pyremap_amd import Remapper, get_lat_lon_descriptor  # noqa: E402
from pyremap_amd.io.netcdf import open_dataset, write_netcdf  # noqa: E402


def main(argv=None):
    parser = argparse.ArgumentParser(
        description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    parser.add_argument('--mesh', required=True, help='MPAS mesh file')
    parser.add_argument('--mesh-name', required=True)
    parser.add_argument('--type', default='cell',
                        choices=['cell', 'edge', 'vertex'])
    parser.add_argument('-i', dest='in_filename', required=True,
                        help='a file with fields on that mesh')
    parser.add_argument('--res', type=float, default=0.5,
                        help='resolution of the lat-lon grid in degrees')
    parser.add_argument('-v', dest='variables', nargs='*', default=None)
    parser.add_argument('-o', dest='out_dir', default='.')
    parser.add_argument('--renormalize', type=float, default=0.01)
    args = parser.parse_args(argv)

    # the mapping file lands under the default name
    # (map_<src>_to_<dst>_analyticbilin.nc) in the output directory
    os.makedirs(args.out_dir, exist_ok=True)
    here = os.getcwd()
    os.chdir(args.out_dir)
    try:
        remapper = Remapper(ntasks=1, method='bilinear', map_tool='analytic',
                            use_tmp=False)
        remapper.src_from_mpas(filename=os.path.join(here, args.mesh)
                               if not os.path.isabs(args.mesh) else args.mesh,
                               mesh_name=args.mesh_name, mesh_type=args.type)
        remapper.dst_descriptor = get_lat_lon_descriptor(dlon=args.res,
                                                         dlat=args.res)
        remapper.build_map()
        dst_name = remapper.dst_descriptor.mesh_name
        in_filename = args.in_filename if os.path.isabs(args.in_filename) \
            else os.path.join(here, args.in_filename)
        # file -> file
        out_file = f'remapped_{dst_name}_file.nc'
        remapper.ncremap(in_filename, out_file, variable_list=args.variables,
                         overwrite=True, renormalize=args.renormalize,
                         replace_mpas_fill=True)
        # the same in memory
        ds = open_dataset(in_filename, variables=args.variables)
        out_array = f'remapped_{dst_name}_array.nc'
        write_netcdf(remapper.remap_numpy(ds, args.renormalize), out_array)
        print(f'{remapper.map_filename}: {args.mesh_name} {args.type}s -> '
              f'{dst_name}; {out_file}, {out_array} '
              f'({remapper.schedule["family"]} kernels)')
    finally:
        os.chdir(here)
    return remapper


if __name__ == '__main__':
    main()
