"""
``Remapper``: the public class of the reference
(``pyremap/remapper/remapper.py:7-532``) with the same constructor, public
attributes and method names, whose ``remap_numpy()`` / ``ncremap()`` apply
the weights on an MI355X instead of through scipy / the NCO subprocess.

Added on top of the reference surface: ``remap`` / ``remap_file`` aliases
(the pyremap-1.x names BASELINE.json uses), ``device`` / ``engine_flags``
knobs and :meth:`from_triplets` for mapping data that is already in memory.
"""
from pyremap_amd.remapper.remap_numpy import (
    _load_mapping,
    _remap_numpy,
    _remap_numpy_array,
)
from pyremap_amd.remapper.setup import _setup_remapper


class Remapper:
    """
    A class for remapping fields using a given mapping file.  The weights
    and indices are loaded once -- sorted into a device-resident CSR -- and
    reused for every field mapped between the same two grids.

    Attributes are those of the reference (``remapper.py:119-137``):
    ``ntasks, src_grid_info, dst_grid_info, map_filename, method, use_tmp,
    expand_dist, expand_factor, src_scrip_filename, dst_scrip_filename,
    format, src_descriptor, dst_descriptor, map_tool, esmf_path, moab_path,
    parallel_exec`` plus

    device : torch.device or str or None
        the GPU that holds the weights (default: the current HIP device)
    devices : list of devices or None
        ONE process, several GPUs: destination rows sharded over them, each
        shard's source rows carried over by peer copies, results assembled on
        ``devices[0]`` -- same calls, same results (bit for bit) as with one
        device.  One process per GPU (torchrun): :meth:`use_process_group`.
    engine_flags : int
        ``pyremap_amd.engine.FLAG_*`` bits; 0 = bit-identical to scipy
    """

    def __init__(
        self,
        ntasks=1,
        map_filename=None,
        method='bilinear',
        src_descriptor=None,
        dst_descriptor=None,
        map_tool='esmf',
        parallel_exec='mpirun',
        use_tmp=True,
        device=None,
        devices=None,
    ):
        self.ntasks = ntasks
        self.src_grid_info = dict()
        self.dst_grid_info = dict()
        self.map_filename = map_filename
        self.method = method
        self.use_tmp = use_tmp
        self.expand_dist = None
        self.expand_factor = None
        self.src_scrip_filename = 'src_mesh.nc'
        self.dst_scrip_filename = 'dst_mesh.nc'
        self.format = 'NETCDF3_64BIT_DATA'
        self.src_descriptor = src_descriptor
        self.dst_descriptor = dst_descriptor
        self.map_tool = map_tool
        self.esmf_path = None
        self.moab_path = None
        self.parallel_exec = parallel_exec
        self.device = device
        #: several GPUs driven from this one process: the destination rows
        #: are sharded over them (pyremap_amd.parallel.MultiDeviceRemap);
        #: fields are handed over and results returned on devices[0].
        #: EXPERIMENTAL: bit-for-bit against one device with the same GPU
        #: listed N times (tests/test_gpu_multi.py); the peer copies and the
        #: cross-device stream ordering have not run on two physical GPUs
        #: (tests/test_gpu_multi.py::test_two_physical_gpus_* are skipped on
        #: one-GPU boxes)
        self.devices = list(devices) if devices else None
        if self.devices and device is None:
            self.device = self.devices[0]
        #: set by use_process_group(): remap_numpy / ncremap as collectives
        self._process_group = None
        self.engine_flags = 0
        #: what RemapPlan.auto_schedule chose for the loaded mapping
        self.schedule = None
        self._ds_map = None
        self._matrix = None
        self._mapping_override = None

    # -- grid definitions (remapper.py:139-421): they only record info ------
    def src_from_lon_lat(self, filename, mesh_name=None, lon_var='lon',
                         lat_var='lat', regional=None):
        self.src_grid_info = _lon_lat_info(filename, mesh_name, lon_var,
                                           lat_var, regional)

    def dst_from_lon_lat(self, filename, mesh_name=None, lon_var='lon',
                         lat_var='lat', regional=None):
        self.dst_grid_info = _lon_lat_info(filename, mesh_name, lon_var,
                                           lat_var, regional)

    def dst_global_lon_lat(self, dlon, dlat, lon_min=-180.0, mesh_name=None):
        info = {'type': 'lon-lat', 'dlon': dlon, 'dlat': dlat,
                'lon_min': lon_min}
        if mesh_name is not None:
            info['name'] = mesh_name
        self.dst_grid_info = info

    def src_from_proj(self, filename, mesh_name, x_var='x', y_var='y',
                      proj_attr=None, proj_str=None):
        self.src_grid_info = _proj_info(filename, mesh_name, x_var, y_var,
                                        proj_attr, proj_str)

    def dst_from_proj(self, filename, mesh_name, x_var='x', y_var='y',
                      proj_attr=None, proj_str=None):
        self.dst_grid_info = _proj_info(filename, mesh_name, x_var, y_var,
                                        proj_attr, proj_str)

    def dst_from_points(self, filename, mesh_name, lon_var='lon',
                        lat_var='lat'):
        self.dst_grid_info = {'type': 'points', 'filename': filename,
                              'name': mesh_name, 'lon': lon_var,
                              'lat': lat_var}

    def src_from_mpas(self, filename, mesh_name, mesh_type='cell'):
        self.src_grid_info = {'type': 'mpas', 'filename': filename,
                              'name': mesh_name,
                              'mpas_mesh_type': mesh_type}

    def dst_from_mpas(self, filename, mesh_name, mesh_type='cell'):
        self.dst_grid_info = {'type': 'mpas', 'filename': filename,
                              'name': mesh_name,
                              'mpas_mesh_type': mesh_type}

    # -- weights --------------------------------------------------------------
    def build_map(self, logger=None):
        """
        The reference shells out to ``ESMF_RegridWeightGen`` / ``mbtempest``
        here (``build_map.py:8-91``); neither exists on the GPU images and
        weight generation for unstructured meshes is outside this engine's
        scope.  ``map_tool='analytic'`` (an addition) covers what has a closed
        form: ``conserve`` / ``bilinear`` / ``neareststod`` between two
        lat-lon grids, or two grids of one projection, and ``bilinear`` from
        an MPAS mesh (cells, edges or vertices, given by its mesh file) to
        anything -- ESMF's weights, reproduced (:mod:`pyremap_amd.weights`).  The file is written to
        ``map_filename`` (default name as in ``setup.py:29-42``).
        """
        from pyremap_amd.remapper.setup import _setup_remapper
        if self.map_tool != 'analytic':
            raise NotImplementedError(
                'pyremap_amd applies existing mapping files on the GPU; '
                'build the mapping file with ESMF_RegridWeightGen / mbtempest '
                '(e.g. through pyremap) and pass it as map_filename, or use '
                "map_tool='analytic' for lat-lon / projection grid pairs and "
                "bilinear maps from an MPAS mesh")
        _setup_remapper(self)
        from pyremap_amd.weights import write_weights
        if logger is not None:
            logger.info(f'analytic {self.method} weights -> '
                        f'{self.map_filename}')
        write_weights(self.map_filename, self.src_descriptor,
                      self.dst_descriptor, self.method)
        self._ds_map = None
        self._matrix = None

    @classmethod
    def from_triplets(cls, row, col, S, frac_b, src_descriptor,
                      dst_descriptor, index_base=1, device=None,
                      map_filename='<memory>'):
        """
        A Remapper whose mapping data is already in memory (what a mapping
        file holds: 1-based ``row``/``col``, ``S``, ``frac_b``).
        """
        import numpy as np

        from pyremap_amd.io.mapfile import MappingFile
        remapper = cls(map_filename=map_filename,
                       src_descriptor=src_descriptor,
                       dst_descriptor=dst_descriptor, device=device)
        row = np.asarray(row)
        col = np.asarray(col)
        if index_base != 1:
            row = row + (1 - index_base)
            col = col + (1 - index_base)
        n_a = int(np.prod(src_descriptor.dim_sizes))
        n_b = int(np.prod(dst_descriptor.dim_sizes))
        remapper._mapping_override = MappingFile(
            n_a, n_b, list(src_descriptor.dim_sizes)[::-1],
            list(dst_descriptor.dim_sizes)[::-1], row, col, S, frac_b)
        return remapper

    def use_process_group(self, group=None, src=0):
        """
        Make ``remap_numpy`` / ``remap_array`` / ``ncremap`` COLLECTIVE calls
        over an initialised ``torch.distributed`` process group (one process
        per GPU; backend ``nccl`` = RCCL over xGMI): every rank makes the
        same call, the data of rank ``src`` is remapped (the other ranks'
        arrays only supply shapes), each rank computes its share of the
        destination rows from the source rows it references, and every rank
        returns the full result; ``ncremap`` writes the file on ``src`` only.
        Call before the first remap (the mapping is sharded when loaded).
        """
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('use_process_group() needs an initialised '
                               'torch.distributed process group')
        if self._ds_map is not None:
            raise RuntimeError('use_process_group() must precede the first '
                               'remap: the mapping is already loaded')
        self._process_group = (group, int(src))

    def load_mapping(self):
        """Read and upload the weights now instead of at the first remap."""
        if self.map_filename is None:
            raise ValueError('No mapping file has been defined')
        _load_mapping(self)
        return self._matrix

    # -- application ----------------------------------------------------------
    def ncremap(self, in_filename, out_filename, variable_list=None,
                overwrite=False, renormalize=None, logger=None,
                replace_mpas_fill=False, parallel_exec=None):
        """
        File -> file remapping with the signature of the reference
        (``remapper.py:434-506``).  The reference runs the NCO ``ncremap``
        executable; here the file is read, remapped on the GPU and written
        back (``pyremap_amd/remapper/remap_file.py``).  ``logger`` receives a
        one-line summary; ``parallel_exec`` is accepted and ignored (there is
        no subprocess to launch).
        """
        from pyremap_amd.remapper.remap_file import _remap_file
        _setup_remapper(self)
        _remap_file(self, in_filename, out_filename, variable_list,
                    overwrite, renormalize, logger, replace_mpas_fill)

    def remap_numpy(self, ds, renormalization_threshold=None):
        """
        Given a source data set, returns a remapped version of the data set,
        possibly masked and renormalized (``remapper.py:508-532``).

        ``ds`` is an ``xarray.Dataset`` / ``DataArray`` (when xarray is
        installed) or a :class:`pyremap_amd.Dataset` / ``DataArray``.
        """
        return _remap_numpy(self, ds, renormalization_threshold)

    def remap_array(self, field, remap_axes, renormalization_threshold=None):
        """
        Array-level entry (the reference's private ``_remap_numpy_array``):
        numpy in -> ``numpy.ma.MaskedArray`` out; device tensor in -> device
        tensor out, nothing leaves the GPU.
        """
        if self.map_filename is None:
            raise ValueError('No mapping file has been defined')
        _load_mapping(self)
        return _remap_numpy_array(self, field, remap_axes,
                                  renormalization_threshold)

    # pyremap-1.x names used by BASELINE.json's north_star
    remap = remap_numpy
    remap_file = ncremap


def _lon_lat_info(filename, mesh_name, lon_var, lat_var, regional):
    info = {'type': 'lon-lat', 'filename': filename, 'lon': lon_var,
            'lat': lat_var}
    if mesh_name is not None:
        info['name'] = mesh_name
    if regional is not None:
        info['regional'] = regional
    return info


def _proj_info(filename, mesh_name, x_var, y_var, proj_attr, proj_str):
    info = {'type': 'proj', 'filename': filename, 'name': mesh_name,
            'x': x_var, 'y': y_var}
    if proj_attr is not None:
        info['proj_attr'] = proj_attr
    elif proj_str is not None:
        info['proj_str'] = proj_str
    else:
        raise ValueError('Must provide one of "proj_attr" or "proj_str".')
    return info
