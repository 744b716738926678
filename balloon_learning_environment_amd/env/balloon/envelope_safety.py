"""env/balloon/envelope_safety.py:40-157 of the reference on the transition's device function (`ble_probe_safety_f32`,
layer 1)."""
from balloon_learning_environment_amd.env.balloon import _probes, control

BUFFER = 250.0                                       # :34 [Pa]
CRITICAL_BUFFER = 150.0                              # :35
HYSTERESIS = 50.0                                    # :36
MAX_SUPERPRESSURE = 2380.0                           # the reference vehicle's envelope (balloon.py:160)

NOMINAL, LOW_CRITICAL, LOW, HIGH, HIGH_CRITICAL = range(5)      # the FSM byte (ble_state_f32.env_fsm)


class EnvelopeSafetyLayer:                           # :93-157
  def __init__(self, max_superpressure: float):
    self._max_superpressure = float(max_superpressure)        # (ABI 5: the device layer takes it as an input)
    self.state_code = NOMINAL

  def get_action(self, action, superpressure: float) -> control.AltitudeControlCommand:
    eff, self.state_code, _ = _probes.safety(1, int(action), superpressure, self.state_code,
                                             max_superpressure=None if self._max_superpressure == MAX_SUPERPRESSURE else self._max_superpressure)
    return control.AltitudeControlCommand(eff)

  @property
  def navigation_is_paused(self) -> bool:            # :139-140
    return self.state_code != NOMINAL
