"""
Polar stereographic projection on the WGS84 ellipsoid, for images without
pyproj (the reference builds its comparison grids with
``pyproj.Proj('+proj=stere +lat_ts=... +lat_0=+-90 +lon_0=0 +ellps=WGS84')``,
``pyremap/polar.py:18-49``).  Formulas: Snyder, "Map Projections -- A Working
Manual" (USGS PP 1395), eqs. 21-33/34, 21-39/40, 15-9, 7-9 (polar aspect,
standard parallel ``lat_ts``).

``ProjectionGridDescriptor`` accepts either a ``pyproj.Proj`` or an object with
this class's ``inverse(x, y) -> (lon, lat)`` method.
"""
import numpy as np

WGS84_A = 6378137.0
WGS84_F = 1.0 / 298.257223563


class PolarStereographic:
    def __init__(self, lat_ts, lat_0, lon_0=0.0, a=WGS84_A, f=WGS84_F,
                 x_0=0.0, y_0=0.0):
        if abs(abs(lat_0) - 90.0) > 1e-12:
            raise ValueError('only the polar aspect (lat_0 = +-90) is '
                             'implemented')
        self.south = lat_0 < 0.0
        self.lat_ts = float(lat_ts)
        self.lat_0 = float(lat_0)
        self.lon_0 = float(lon_0)
        self.a = float(a)
        self.e = float(np.sqrt(f * (2.0 - f)))
        self.x_0 = float(x_0)
        self.y_0 = float(y_0)
        phi_c = np.radians(abs(self.lat_ts))
        e = self.e
        if abs(phi_c - 0.5 * np.pi) < 1e-14:
            # true scale at the pole (k_0 = 1)
            self._scale = 2.0 * self.a / np.sqrt(
                (1.0 + e) ** (1.0 + e) * (1.0 - e) ** (1.0 - e))
        else:
            m_c = np.cos(phi_c) / np.sqrt(1.0 - (e * np.sin(phi_c)) ** 2)
            self._scale = self.a * m_c / self._t(phi_c)

    @property
    def srs(self):
        return (f'+proj=stere +lat_ts={self.lat_ts} +lat_0={self.lat_0} '
                f'+lon_0={self.lon_0} +k_0=1.0 +x_0={self.x_0} '
                f'+y_0={self.y_0} +ellps=WGS84')

    def _t(self, phi):
        e = self.e
        s = np.sin(phi)
        return np.tan(0.25 * np.pi - 0.5 * phi) / \
            ((1.0 - e * s) / (1.0 + e * s)) ** (0.5 * e)

    def forward(self, lon, lat):
        """(lon, lat) in degrees -> (x, y) in metres."""
        lon = np.asarray(lon, dtype=np.float64)
        lat = np.asarray(lat, dtype=np.float64)
        sign = -1.0 if self.south else 1.0
        phi = np.radians(sign * lat)
        dlam = np.radians(sign * (lon - self.lon_0))
        rho = self._scale * self._t(phi)
        x = sign * rho * np.sin(dlam)
        y = -sign * rho * np.cos(dlam)
        return x + self.x_0, y + self.y_0

    def inverse(self, x, y):
        """(x, y) in metres -> (lon, lat) in degrees."""
        x = np.asarray(x, dtype=np.float64) - self.x_0
        y = np.asarray(y, dtype=np.float64) - self.y_0
        sign = -1.0 if self.south else 1.0
        xn, yn = sign * x, sign * y
        rho = np.hypot(xn, yn)
        t = rho / self._scale
        e = self.e
        phi = 0.5 * np.pi - 2.0 * np.arctan(t)
        for _ in range(12):
            s = np.sin(phi)
            phi = 0.5 * np.pi - 2.0 * np.arctan(
                t * ((1.0 - e * s) / (1.0 + e * s)) ** (0.5 * e))
        lam = np.arctan2(xn, -yn)
        lon = self.lon_0 + sign * np.degrees(lam)
        lon = np.where(rho == 0.0, self.lon_0, lon)
        # (-180, 180], the range pyproj reports
        lon = (lon + 180.0) % 360.0 - 180.0
        lon = np.where(lon == -180.0, 180.0, lon)
        lat = sign * np.degrees(phi)
        return lon, lat


def projection_from_string(proj_str):
    """
    A projection object for a PROJ string (``remapper/descriptor.py:150-165``
    hands the string to ``pyproj.Proj``).  With pyproj installed that is what
    is returned; without it the polar stereographic aspect on WGS84
    (``+proj=stere +lat_0=+-90 ...``, what the polar grids of MPAS analysis
    use) is parsed into a :class:`PolarStereographic`; anything else needs
    pyproj.
    """
    try:
        import pyproj
        return pyproj.Proj(proj_str)
    except ImportError:
        pass
    params = {}
    for token in str(proj_str).split():
        if token.startswith('+'):
            key, _, value = token[1:].partition('=')
            params[key] = value
    ellps_ok = params.get('ellps', params.get('datum', 'WGS84')) == 'WGS84'
    if params.get('proj') == 'stere' and ellps_ok and \
            abs(abs(float(params.get('lat_0', 0.0))) - 90.0) < 1e-12 and \
            float(params.get('k_0', params.get('k', 1.0))) == 1.0:
        lat_0 = float(params['lat_0'])
        return PolarStereographic(
            lat_ts=float(params.get('lat_ts', lat_0)), lat_0=lat_0,
            lon_0=float(params.get('lon_0', 0.0)),
            x_0=float(params.get('x_0', 0.0)),
            y_0=float(params.get('y_0', 0.0)))
    raise NotImplementedError(
        f'projection {proj_str!r}: without pyproj only polar stereographic '
        f'projections on WGS84 are available')


def antarctic_stereographic():
    """``polar.py:39-49``: lat_ts = -71, lat_0 = -90, lon_0 = 0 (EPSG:3031)."""
    return PolarStereographic(lat_ts=-71.0, lat_0=-90.0, lon_0=0.0)


def arctic_stereographic():
    """``polar.py:18-36``: lat_ts = 75, lat_0 = 90, lon_0 = 0."""
    return PolarStereographic(lat_ts=75.0, lat_0=90.0, lon_0=0.0)


def project_to_lat_lon(projection, x, y):
    """x, y -> (lat, lon) in degrees through a ``pyproj.Proj`` or a
    :class:`PolarStereographic`; (None, None) if neither is usable."""
    if projection is None:
        return None, None
    if hasattr(projection, 'inverse'):
        lon, lat = projection.inverse(x, y)
        return lat, lon
    try:
        import pyproj
    except ImportError:
        return None, None
    lat_lon = pyproj.Proj(proj='latlong', datum='WGS84')
    transformer = pyproj.Transformer.from_proj(projection, lat_lon)
    lon, lat = transformer.transform(x, y)
    return lat, lon
