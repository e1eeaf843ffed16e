"""
Minimal labelled-array containers with the slice of the ``xarray`` interface
that the remapping path touches.

The reference takes and returns ``xarray.Dataset`` / ``xarray.DataArray``
(``pyremap/remapper/remap_numpy.py:19-69,150-220``).  ``xarray`` is not
installed in the build container or on the MI355X boxes, so
:class:`DataArray` and :class:`Dataset` here implement exactly the members
that path uses -- ``dims, sizes, shape, values, coords, attrs, name,
data_vars, __getitem__, drop_vars, map, from_dict`` -- with xarray's
semantics (masked arrays become NaN on construction, ``Dataset.map`` rebuilds
the coordinates from the mapped arrays, ...).  When real ``xarray`` is
importable the :class:`~pyremap_amd.remapper.Remapper` accepts and returns
its objects instead (see ``pyremap_amd/remapper/remap_numpy.py``).

``oracle/make_goldens.py`` installs this module as the ``xarray`` the
reference's own ``_remap_numpy`` imports, so the behaviour of these classes is
exercised by the reference code itself when the goldens are generated.
"""
from collections import OrderedDict

import numpy as np


class LazyValues:
    """
    The values of a variable that are PRODUCED ON DEMAND and not retained:
    a variable of a file that has not been read yet, or the remapped form of
    one that has not been computed yet.  ``shape`` / ``dtype`` are known up
    front; :meth:`load` reads (computes) and returns the array -- every call
    does so again, so consumers take it once.  :meth:`prefetch` (optional)
    starts the work early without waiting for it.  This is what lets
    ``Remapper.ncremap`` stream a file variable by variable -- as NCO does,
    ``pyremap/remapper/ncremap.py:117-145`` -- with bounded host memory.
    """

    def __init__(self, shape, dtype, load, prefetch=None):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._load = load
        self._prefetch = prefetch
        self._started = None

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    def prefetch(self):
        """Start producing the values (no-op when there is nothing to do
        ahead of time); the next :meth:`load` picks the work up."""
        if self._prefetch is not None and self._started is None:
            self._started = self._prefetch()

    def load(self):
        if self._started is not None:
            finish, self._started = self._started, None
            return np.asarray(finish())
        return np.asarray(self._load())

    def copy(self):
        return self

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.load(), dtype=dtype)


def _as_data(data):
    """xarray's ``as_compatible_data``: a masked array becomes NaN-filled."""
    if isinstance(data, LazyValues):
        return data
    if isinstance(data, np.ma.MaskedArray):
        mask = np.ma.getmaskarray(data)
        if mask.any():
            raw = np.ma.getdata(data)
            if raw.dtype.kind == 'f':
                out = np.array(raw, copy=True)
            else:
                out = raw.astype(np.float64)
            out[mask] = np.nan
            return out
        return np.asarray(np.ma.getdata(data))
    return np.asarray(data)


class DataArray:
    """A named N-D array with dimension names, coordinates and attributes."""

    def __init__(self, data, coords=None, dims=None, name=None, attrs=None):
        self._data = _as_data(data)
        if dims is None:
            dims = tuple(f'dim_{i}' for i in range(self._data.ndim))
        if isinstance(dims, str):
            dims = (dims,)
        self.dims = tuple(dims)
        if len(self.dims) != self._data.ndim:
            raise ValueError(
                f'different number of dimensions on data and dims: '
                f'{self._data.ndim} vs {len(self.dims)}')
        self.name = name
        self.attrs = OrderedDict(attrs) if attrs is not None else \
            OrderedDict()
        self.encoding = {}
        self.coords = OrderedDict()
        if coords is not None:
            for cname, cval in coords.items():
                self.coords[cname] = _as_coord(cname, cval)

    def __getattr__(self, name):
        # ``da.units`` / ``da.lat``: attributes first, then coordinates
        if name.startswith('_'):
            raise AttributeError(name)
        attrs = self.__dict__.get('attrs', {})
        if name in attrs:
            return attrs[name]
        coords = self.__dict__.get('coords', {})
        if name in coords:
            return coords[name]
        raise AttributeError(
            f"'DataArray' object has no attribute '{name}'")

    # -- array-like members ------------------------------------------------
    @property
    def values(self):
        if isinstance(self._data, LazyValues):
            return self._data.load()
        return self._data

    @property
    def data(self):
        return self.values

    @property
    def is_lazy(self):
        """Are the values produced on demand (:class:`LazyValues`)?"""
        return isinstance(self._data, LazyValues)

    @property
    def shape(self):
        return self._data.shape

    @property
    def ndim(self):
        return self._data.ndim

    @property
    def dtype(self):
        return self._data.dtype

    @property
    def sizes(self):
        return OrderedDict(zip(self.dims, self._data.shape))

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._data, dtype=dtype)

    def __repr__(self):
        return (f'<pyremap_amd.DataArray {self.name!r} '
                f'{dict(self.sizes)} {self.dtype}>')

    def copy(self):
        return DataArray(self._data.copy(), coords=self.coords,
                         dims=self.dims, name=self.name, attrs=self.attrs)

    def _bare(self):
        """The same variable without coordinates (as stored in a Dataset)."""
        return DataArray(self._data, dims=self.dims, name=self.name,
                         attrs=self.attrs)

    def to_dict(self):
        return {
            'coords': {k: {'dims': v.dims, 'data': v.values,
                           'attrs': dict(v.attrs)}
                       for k, v in self.coords.items()},
            'attrs': dict(self.attrs),
            'dims': self.dims,
            'data': self.values,
            'name': self.name,
        }

    @classmethod
    def from_dict(cls, d):
        """``xarray.DataArray.from_dict`` (used at remap_numpy.py:218)."""
        coords = None
        if 'coords' in d:
            coords = OrderedDict()
            for cname, cdict in d['coords'].items():
                coords[cname] = DataArray(
                    cdict['data'], dims=cdict['dims'], name=cname,
                    attrs=cdict.get('attrs'))
        return cls(d['data'], coords=coords, dims=d.get('dims'),
                   name=d.get('name'), attrs=d.get('attrs'))


def _as_coord(name, value):
    if isinstance(value, DataArray):
        return value._bare() if value.coords else value
    if isinstance(value, tuple):
        dims, data = value[0], value[1]
        attrs = value[2] if len(value) > 2 else None
        return DataArray(data, dims=dims, name=name, attrs=attrs)
    if isinstance(value, dict):
        return DataArray(value['data'], dims=value['dims'], name=name,
                         attrs=value.get('attrs'))
    return DataArray(np.asarray(value), dims=(name,), name=name)


class _DataVars:
    """Read-only mapping over the non-coordinate variables of a Dataset."""

    def __init__(self, ds):
        self._ds = ds

    def __iter__(self):
        return iter([k for k in self._ds._vars if k not in
                     self._ds._coord_names])

    def __len__(self):
        return len(list(iter(self)))

    def __contains__(self, key):
        return key in self._ds._vars and key not in self._ds._coord_names

    def __getitem__(self, key):
        if key not in self:
            raise KeyError(key)
        return self._ds[key]

    def keys(self):
        return list(iter(self))

    def items(self):
        return [(k, self._ds[k]) for k in self]

    def values(self):
        return [self._ds[k] for k in self]


class _Coords(_DataVars):
    def __iter__(self):
        return iter([k for k in self._ds._vars if k in
                     self._ds._coord_names])

    def __contains__(self, key):
        return key in self._ds._vars and key in self._ds._coord_names


class Dataset:
    """A dict of aligned :class:`DataArray` variables plus global attrs."""

    def __init__(self, data_vars=None, coords=None, attrs=None):
        self._vars = OrderedDict()
        self._coord_names = set()
        self.encoding = {}
        self.attrs = OrderedDict(attrs) if attrs is not None else \
            OrderedDict()
        if data_vars is not None:
            for name, value in data_vars.items():
                self[name] = value
        if coords is not None:
            for name, value in coords.items():
                self._set_coord(name, _as_coord(name, value))

    # -- construction ------------------------------------------------------
    def _check_sizes(self, name, da):
        sizes = self.sizes
        for dim, size in zip(da.dims, da.shape):
            if dim in sizes and sizes[dim] != size:
                raise ValueError(
                    f'conflicting sizes for dimension {dim!r}: length '
                    f'{size} on {name!r} and length {sizes[dim]} on the '
                    f'dataset')

    def _set_coord(self, name, da):
        self._check_sizes(name, da)
        bare = da._bare()
        bare.name = name
        self._vars[name] = bare
        self._coord_names.add(name)

    def __setitem__(self, name, value):
        if isinstance(value, DataArray):
            da = value
        elif isinstance(value, tuple):
            dims, data = value[0], value[1]
            attrs = value[2] if len(value) > 2 else None
            da = DataArray(data, dims=dims, attrs=attrs)
        else:
            da = DataArray(value)
        self._check_sizes(name, da)
        # coordinates carried by the array become dataset coordinates
        for cname, cval in da.coords.items():
            if cname not in self._vars:
                self._set_coord(cname, cval)
        bare = da._bare()
        bare.name = name
        self._vars[name] = bare
        self._coord_names.discard(name)

    # -- access ------------------------------------------------------------
    @property
    def data_vars(self):
        return _DataVars(self)

    @property
    def coords(self):
        return _Coords(self)

    @property
    def variables(self):
        return self._vars

    @property
    def sizes(self):
        sizes = OrderedDict()
        for da in self._vars.values():
            for dim, size in zip(da.dims, da.shape):
                sizes.setdefault(dim, size)
        return sizes

    dims = sizes

    def __contains__(self, name):
        return name in self._vars

    def __iter__(self):
        return iter(self.data_vars)

    def __getitem__(self, name):
        if name not in self._vars:
            raise KeyError(name)
        var = self._vars[name]
        coords = OrderedDict()
        for cname in self._vars:
            if cname in self._coord_names and cname != name:
                cvar = self._vars[cname]
                if all(dim in var.dims for dim in cvar.dims):
                    coords[cname] = cvar
        if name in self._coord_names and var.dims == (name,):
            coords[name] = var
        # (var._data, not var.values: a lazy variable stays lazy)
        da = DataArray(var._data, coords=coords, dims=var.dims, name=name)
        # as in xarray, the variable's attrs / encoding are shared, so
        # ``ds['lat'].attrs['units'] = ...`` sticks
        da.attrs = var.attrs
        da.encoding = var.encoding
        return da

    def __getattr__(self, name):
        # ``ds.x`` / ``ds.title``: variables first, then global attributes
        if name.startswith('_'):
            raise AttributeError(name)
        if name in self.__dict__.get('_vars', ()):
            return self[name]
        attrs = self.__dict__.get('attrs', {})
        if name in attrs:
            return attrs[name]
        raise AttributeError(
            f"'Dataset' object has no attribute '{name}'")

    def __repr__(self):
        lines = [f'<pyremap_amd.Dataset {dict(self.sizes)}>']
        for name in self.coords:
            v = self._vars[name]
            lines.append(f'  * {name} {v.dims} {v.dtype}')
        for name in self.data_vars:
            v = self._vars[name]
            lines.append(f'    {name} {v.dims} {v.dtype}')
        return '\n'.join(lines)

    def copy(self):
        out = Dataset(attrs=self.attrs)
        for name, var in self._vars.items():
            out._vars[name] = var
        out._coord_names = set(self._coord_names)
        out.encoding = dict(self.encoding)
        return out

    def drop_vars(self, names):
        if isinstance(names, str):
            names = [names]
        out = self.copy()
        for name in names:
            if name not in out._vars:
                raise ValueError(f'variable {name!r} not in the dataset')
            del out._vars[name]
            out._coord_names.discard(name)
        return out

    def map(self, func, keep_attrs=None, args=(), **kwargs):
        """
        ``xarray.Dataset.map``: apply ``func`` to every data variable and
        assemble a new Dataset from the results (coordinates come from the
        returned arrays).  Used at ``remap_numpy.py:48-55``.
        """
        results = OrderedDict()
        for name in self.data_vars:
            res = func(self[name], *args, **kwargs)
            if not isinstance(res, DataArray):
                src = self._vars[name]
                res = DataArray(res, dims=src.dims, name=name)
            if keep_attrs:
                res.attrs = OrderedDict(self._vars[name].attrs)
            results[name] = res
        return Dataset(results, attrs=self.attrs if keep_attrs else None)
