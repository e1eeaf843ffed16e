"""
``pyremap.utility`` under its reference name: ``write_netcdf`` with the
reference's signature (``pyremap/utility.py:8-72``) on top of this package's
own writer.  ``check_call`` (the subprocess helper for ESMF / NCO) has no
counterpart: nothing here shells out.
"""
from pyremap_amd.io import netcdf as _netcdf


def write_netcdf(ds, filename, format, engine=None, logger=None,
                 fillvalues=None):
    """
    Write a Dataset with netCDF4-style fill values: numeric variables that
    hold NaNs get ``_FillValue`` = the default fill value of their type, all
    other variables none (``utility.py:38-51``).

    ``format``: ``NETCDF4`` / ``NETCDF4_CLASSIC`` (this package's HDF5
    writer) or a classic format (``NETCDF3_CLASSIC``, ``NETCDF3_64BIT``,
    ``NETCDF3_64BIT_DATA``; the reference writes the last one through
    NetCDF-4 + ``ncks -5``, :53-72, here it is written directly).  ``engine``
    is accepted for compatibility and ignored.
    """
    if fillvalues is not None:
        import numpy as np
        fillvalues = {np.dtype(k).str[1:]: v for k, v in fillvalues.items()
                      if not str(k).startswith('S')}
    _netcdf.write_netcdf(ds, filename, format=format, fillvalues=fillvalues)
    if logger is not None:
        logger.info(f'wrote {filename} ({format})')
