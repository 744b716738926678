"""MI355X-native vectorised Balloon Learning Environment transition.

Native object: `VecBalloonArena` / `vec_state.VecSimulator` (N environments, device tensors,
one HIP kernel launch per agent step).  Reference-shaped facades: `BalloonArena`,
`BalloonEnv`.  The compute path is libble_hip.so only (see _lib.py); importing this
package does not load it, using an arena / simulator does and fails loudly if it is absent.
"""
__version__ = '0.1.0'
