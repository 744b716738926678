"""Unit value types with the reference's API (utils/units.py:26-373): Distance (stored in
metres), Velocity (m/s), Energy (Wh), Power (W).  Host glue only -- device state is raw
float32 arrays in the units named in include/ble_abi.h."""
import datetime as dt

_METERS_PER_FOOT = 0.3048


class Distance:
  def __init__(self, *, m=0.0, meters=0.0, km=0.0, kilometers=0.0, feet=0.0):
    self._d = m + meters + (km + kilometers) * 1000.0 + feet * _METERS_PER_FOOT

  m = property(lambda self: self._d)
  meters = property(lambda self: self._d)
  km = property(lambda self: self._d / 1000.0)
  kilometers = property(lambda self: self._d / 1000.0)
  feet = property(lambda self: self._d / _METERS_PER_FOOT)

  def __add__(self, o): return Distance(m=self._d + _dist(o)._d)
  def __sub__(self, o): return Distance(m=self._d - _dist(o)._d)
  def __mul__(self, o): return Distance(m=self._d * _num(o))
  __rmul__ = __mul__

  def __truediv__(self, o):
    if isinstance(o, (int, float)): return Distance(m=self._d / o)
    if isinstance(o, dt.timedelta): return Velocity(mps=self._d / o.total_seconds())
    if isinstance(o, Distance): return self._d / o._d
    raise NotImplementedError(f'Cannot divide distance by {type(o)}')

  def __eq__(self, o): return abs(self._d - o._d) < 1e-9
  def __lt__(self, o): return self._d < o._d
  def __le__(self, o): return self._d <= o._d
  def __gt__(self, o): return self._d > o._d
  def __ge__(self, o): return self._d >= o._d
  def __repr__(self): return f'Distance(m={self._d})'


def _dist(o):
  if not isinstance(o, Distance): raise NotImplementedError(f'Cannot combine Distance and {type(o)}')
  return o


def _num(o):
  if not isinstance(o, (int, float)): raise NotImplementedError(f'Cannot multiply by {type(o)}')
  return o


class Velocity:
  def __init__(self, *, mps=0.0, meters_per_second=0.0, kmph=0.0, kilometers_per_hour=0.0):
    self._v = mps + meters_per_second + (kmph + kilometers_per_hour) * 1000 / 3600

  mps = property(lambda self: self._v)
  meters_per_second = property(lambda self: self._v)
  kmph = property(lambda self: self._v * 3600 / 1000)
  kilometers_per_hour = property(lambda self: self._v * 3600 / 1000)

  def __add__(self, o): return Velocity(mps=self._v + o._v)
  def __sub__(self, o): return Velocity(mps=self._v - o._v)

  def __mul__(self, o):
    if isinstance(o, dt.timedelta): return Distance(m=self._v * o.total_seconds())
    raise NotImplementedError(f'Cannot multiply velocity with {type(o)}')
  __rmul__ = __mul__

  def __truediv__(self, o): return Velocity(mps=self._v / _num(o))
  def __eq__(self, o): return abs(self._v - o._v) < 1e-9
  def __repr__(self): return f'{self._v} m/s'


class Energy:
  def __init__(self, *, watt_hours=0.0): self._wh = watt_hours
  watt_hours = property(lambda self: self._wh)
  def __add__(self, o): return Energy(watt_hours=self._wh + o._wh)
  def __sub__(self, o): return Energy(watt_hours=self._wh - o._wh)
  def __truediv__(self, o): return self._wh / o._wh
  def __mul__(self, o): return Energy(watt_hours=self._wh * _num(o))
  __rmul__ = __mul__
  def __gt__(self, o): return self._wh > o._wh
  def __ge__(self, o): return self._wh >= o._wh
  def __eq__(self, o): return self._wh == o._wh
  def __repr__(self): return f'Energy(watt_hours={self._wh})'


class Power:
  def __init__(self, *, watts=0.0): self._w = watts
  watts = property(lambda self: self._w)
  def __add__(self, o): return Power(watts=self._w + o._w)
  def __sub__(self, o): return Power(watts=self._w - o._w)

  def __mul__(self, o):
    if isinstance(o, dt.timedelta): return Energy(watt_hours=self._w * timedelta_to_hours(o))
    raise NotImplementedError(f'Cannot multiply Power with {type(o)}')
  __rmul__ = __mul__
  def __gt__(self, o): return self._w > o._w
  def __eq__(self, o): return self._w == o._w
  def __repr__(self): return f'Power(watts={self._w})'


def distance_to_degrees(d): return d.km / 111.0


def relative_distance(x, y): return Distance(m=(x.m * x.m + y.m * y.m) ** 0.5)


def seconds_to_hours(s): return s / 3600.0


def timedelta_to_hours(d): return seconds_to_hours(d.total_seconds())


def datetime(year, month, day, hour=0, minute=0, second=0, microsecond=0, tzinfo=dt.timezone.utc, *, fold=0):
  return dt.datetime(year, month, day, hour, minute, second, microsecond, tzinfo, fold=fold)


def datetime_from_timestamp(timestamp):
  return dt.datetime.fromtimestamp(timestamp, tz=dt.timezone.utc)
