"""env/balloon/control.py:21-25 of the reference."""
import enum


class AltitudeControlCommand(enum.IntEnum):
  DOWN = 0
  STAY = 1
  UP = 2
