/*
 * ble_abi.h -- C ABI of libble_hip.so, the MI355X (gfx950) vectorised Balloon Learning
 * Environment transition.
 *
 * The reference (google/balloon-learning-environment, pure Python) has no FFI for this
 * path; the seams it does have are Python classes.  Each entry point below replaces the
 * arithmetic behind one of those seams, for N environments at once, and is what a
 * ctypes binding inside the reference would call (INTEGRATION.md shows the stub).
 * Paths are relative to /root/reference/balloon_learning_environment/.
 *
 * Conventions
 *  - All array pointers are DEVICE pointers (HIP), caller-owned, struct-of-arrays,
 *    length n unless stated.  No torch types, no C++ types.
 *  - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *    default stream).  The caller keeps the buffers alive until the stream has passed
 *    the call, and synchronises before reading results on the host.
 *  - Return value: BLE_OK (0) or a negative BLE_E_* code for host-side argument / launch
 *    errors.  The library never throws, aborts or asserts on the device.  Conditions on
 *    which the reference raises *inside* the arithmetic (range checks) are OR-ed into
 *    the device word `err_flags` (BLE_FLAG_*), which the host mirror turns back into the
 *    reference's exceptions.
 *  - Re-entrant: no global mutable state; different streams may run concurrently on
 *    disjoint buffers.
 *  - Units and encodings follow the reference: metres, Pa, K, mol, Wh, W, kg/s, seconds;
 *    actions 0=DOWN 1=STAY 2=UP (env/balloon/control.py:21-25); status 0=OK
 *    1=OUT_OF_POWER 2=BURST 3=ZEROPRESSURE (env/balloon/balloon.py:66-70).
 */
#ifndef BLE_ABI_H_
#define BLE_ABI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: ble_state_f32 gained the optional episode_cache; ble_wind_noise_f32 the optional harmonic_cache; ble_gp_history_f32 gained chol_stride and the carried slab grew to 7620 doubles (packed Cholesky L -> Lt D Lt^T +
 *    drop vector + zeta / d): a caller built against version 1 allocates 7260 doubles per environment. */
/* 3: ble_step_n_f32 gained `noise` (ble_noise_gen: the wind-noise generator evaluated inside the fused rollout). */
/* 4: ble_set_step_form, ble_probe_latlng_f64, the shard forms ble_reset_at_f32 / ble_wind_noise_at_f32 / ble_noise_gen.env_offset. */
/* 5: ble_state_f32 gained the optional `vehicle` (ble_vehicle: BalloonState's flight-vehicle constants and
 *    power_safety_layer_enabled as run-time inputs); ble_noise_primitive_version(); ble_set_step_form(2) is refused by the product
 *    library (the two-wavefront form is an experiment build); ble_step_n_f32 rejects a negative ble_noise_gen.env_offset. */
#define BLE_ABI_VERSION 5

/* Version of the wind-noise PRIMITIVE's bit pattern (csrc/ble_noise.h::simplex4 and the hash / draw streams under it).  The primitive
 * is this library's own (the reference's opensimplex==0.3 noise4d is absent and unpinned), so its values are defined by this
 * repository -- and whenever they change, for whatever reason, this number is bumped: a noise seed recorded against version k flies
 * the same wind only on a library that reports k.  The host mirror refuses a library whose version differs from the one its oracle
 * (oracle/noise_oracle.py::PRIMITIVE_VERSION) and the committed fixture (tests/golden/f14_wind_noise.npz::noise_primitive_version)
 * were made with; checkpoints carry it.  1: rounds 2-4.  2: round 5 (explicit FMAs in a corner's sums, offsets by select). */
#define BLE_NOISE_PRIMITIVE_VERSION 2

/* return codes */
#define BLE_OK 0
#define BLE_E_INVALID_ARG (-1) /* NULL required pointer, n < 0, substeps outside 1 .. BLE_MAX_SUBSTEPS ... */
#define BLE_E_LAUNCH (-2)      /* hipLaunchKernel / hipGetLastError failed */
#define BLE_E_NO_DEVICE (-3)   /* no HIP device visible */

/* bits OR-ed into *err_flags by the kernels (where the reference raises) */
#define BLE_FLAG_PRESSURE_RANGE 1u /* standard_atmosphere.py:126-127 assert */
#define BLE_FLAG_ABSORPTIVITY 2u   /* thermal.py:142-145 ValueError */
#define BLE_FLAG_SOLAR_RANGE 4u    /* solar.py:190-197 ValueError */
#define BLE_FLAG_POWER_TABLE 16u   /* power_table.py:24 assert */
#define BLE_FLAG_NONFINITE 32u     /* a state value became NaN/Inf (no reference analogue) */
#define BLE_FLAG_GP_WINDOW 64u     /* > 120 observations inside the WindGP's 6 h window (steps < 180 s): oldest dropped */
#define BLE_FLAG_PRESSURE_SEARCH 128u /* pressure_range_builder.py:104-108,180-182 ValueError */
#define BLE_FLAG_DAY_CYCLE 256u    /* features.py:432-437 ZeroDivisionError: the next sunrise exactly one day after the next sunset
                                      (polar night); the two day-cycle features of that environment are NaN */

/* wind grid geometry: generative/vae.py:30-38,77-93 (FieldShape defaults) */
#define BLE_GRID_NX 21 /* x (lat axis of the grid), -500..500 km step 50 */
#define BLE_GRID_NY 21 /* y, -500..500 km step 50 */
#define BLE_GRID_NP 10 /* pressure, 5000..14000 Pa step 1000 */
#define BLE_GRID_NT 9  /* time, 0..48 h step 6 */
#define BLE_GRID_FLOATS (BLE_GRID_NX * BLE_GRID_NY * BLE_GRID_NP * BLE_GRID_NT * 2) /* 79380 */
#define BLE_COUNT_SLOTS 64 /* width of the live-environment counter, see ble_step_f32 */

/*
 * BalloonState's flight-vehicle constants (balloon.py:156-172: dataclass FIELDS with defaults), mols_lift_gas (:183, a state
 * field no transition changes) and power_safety_layer_enabled (:200, read by simulate_step :305).  A HOST struct of doubles --
 * the reference's are Python floats and several defaults (0.0199, 183.7, 3058.56) are not float32 numbers.  One vehicle per
 * call: every environment of a batch flies the same one (the reference builds one BalloonState per balloon; batches of
 * different vehicles are different calls).  ble_vehicle_default() fills in the reference's defaults.
 */
typedef struct ble_vehicle {
  double envelope_volume_base;         /* [m^3]   1804      balloon.py:157 */
  double envelope_volume_dv_pressure;  /* [m^3/Pa] 0.0199   :158 */
  double envelope_mass;                /* [kg]    68.5      :159 */
  double envelope_max_superpressure;   /* [Pa]    2380      :160 (burst threshold and EnvelopeSafetyLayer's argument, :204-205) */
  double envelope_cod;                 /* [.]     0.25      :163 */
  double payload_mass;                 /* [kg]    92.5      :166 */
  double nighttime_power_load_w;       /* [W]     183.7     :168 */
  double daytime_power_load_w;         /* [W]     120.4     :169 */
  double acs_valve_hole_diameter_m;    /* [m]     0.04      :171 */
  double battery_capacity_wh;          /* [Wh]    3058.56   :173 */
  double mols_lift_gas;                /* [mol]   6830      :183 */
  int32_t power_safety_layer_enabled;  /* bool    1         :200 */
  int32_t reserved_;                   /* 0 */
} ble_vehicle;

/*
 * Per-environment simulator state, struct of device arrays.
 * Replaces: BalloonState (env/balloon/balloon.py:73-250), the three safety-layer
 * objects it owns (altitude_safety.py:63-111, envelope_safety.py:93-157,
 * power_safety.py:26-126) and Atmosphere's per-episode alpha
 * (standard_atmosphere.py:76-87).
 * The flight-vehicle constants: `vehicle` below (ABI 5).  NULL -- every BASELINE configuration -- selects kernels in which the
 * reference's defaults are compile-time constants (csrc/ble_physics.h::VehicleDefault); a non-NULL vehicle selects a second
 * instantiation of the same lane functions that reads them from scalar registers (csrc/ble_physics.h::VehicleRt).
 */
typedef struct ble_state_f32 {
  /* mutable, read+written by ble_step_f32 (balloon.py:175-195) */
  float* x;                    /* [m]  units.Distance x, W->E offset from the station */
  float* y;                    /* [m] */
  float* pressure;             /* [Pa] */
  float* ambient_temperature;  /* [K] */
  float* internal_temperature; /* [K] */
  float* envelope_volume;      /* [m^3] */
  float* superpressure;        /* [Pa] */
  float* mols_air;             /* [mol] */
  float* battery_charge;       /* [Wh] */
  /* derived, written by ble_step_f32 (balloon.py:189-193) */
  float* acs_power;      /* [W] */
  float* acs_mass_flow;  /* [kg/s] */
  float* solar_charging; /* [W] */
  float* power_load;     /* [W] */
  /* per-episode constants, read only */
  const float* center_lat_deg;     /* BalloonState.center_latlng */
  const float* center_lng_deg;
  const float* upwelling_infrared; /* [W/m^2] balloon.py:208 */
  const float* alpha;              /* Atmosphere lapse-rate mix, standard_atmosphere.py:82-84 */
  const int64_t* start_unix;       /* date_time when time_elapsed == 0, UTC seconds */
  /* clocks: date_time = start_unix + time_elapsed_s (balloon.py:546-547) */
  int32_t* time_elapsed_s;
  /* PowerSafetyLayer._sunrise_with_hysteresis / ._sunset, seconds relative to start_unix */
  int32_t* sunrise_h_rel;
  int32_t* sunset_rel;
  /* discrete state */
  uint8_t* status;       /* BalloonStatus */
  uint8_t* last_command; /* raw action of the last step, balloon.py:286 */
  uint8_t* alt_fsm;      /* 0 NOMINAL 1 LOW 2 VERY_LOW         (altitude_safety.py:40-44) */
  uint8_t* env_fsm;      /* 0 NOMINAL 1 LOW_CRITICAL 2 LOW 3 HIGH 4 HIGH_CRITICAL (envelope_safety.py:45-50) */
  uint8_t* power_paused; /* PowerSafetyLayer.navigation_is_paused */
  /* OPTIONAL (may be NULL), opaque: [BLE_EPISODE_CACHE_ROWS][n] doubles, zero-initialised by the caller.  What the
   * transition derives from the per-episode constants alone (the atmosphere's two transition pressures -- two pows --,
   * sin / cos of the centre latitude, the earth-IR heat per unit area), keyed by the bit patterns of (alpha,
   * center_lat_deg, upwelling_infrared).  ble_reset_f32 fills it; ble_step_f32 / ble_step_n_f32 read it and, where an
   * entry does not match the constants in `st` (edited by hand, or never reset on the device), recompute and store it --
   * so it can never go stale.  NULL: recomputed by every launch (0.8 us per launch).  Entries are functions of the episode's
   * constants alone, not of the vehicle. */
  double* episode_cache;
  /* OPTIONAL (may be NULL), ABI 5: HOST pointer to the vehicle every environment of this call flies, read on the host when the
   * call is made (not retained).  NULL = the reference's defaults.  Honoured by ble_step_f32 / ble_step_n_f32 (one lane per
   * environment whatever the batch size), ble_reset_f32 / ble_reset_at_f32 (the cold start) and ble_observe_f32 (battery state of
   * charge, excess energy, the reachable pressure range).  A vehicle with a non-positive volume base, dV/dp, capacity, drag
   * coefficient or maximum superpressure <= 300 Pa (envelope_safety.py's bands would overlap) is rejected with BLE_E_INVALID_ARG. */
  const ble_vehicle* vehicle;
} ble_state_f32;
#define BLE_EPISODE_CACHE_ROWS 7
#define BLE_MAX_SUBSTEPS 60
/* Up to this many environments ble_step_f32 / ble_step_n_f32 -- with or without a wind-noise generator -- run the
 * four-wavefronts-per-environment form of the transition (csrc/ble_step_split.h: 4 x n / 64 waves -- one per SIMD up to
 * 16 384 environments, two up to 32 768), above it the one-lane-per-environment kernel (one wave per SIMD at 65 536).
 * The forms are bit-identical; ble_set_step_form() forces one. */
#define BLE_SPLIT_MAX_ENVS 32768

int ble_abi_version(void);

/* BLE_NOISE_PRIMITIVE_VERSION of the loaded library (above). */
int ble_noise_primitive_version(void);

/* Fills *v with the reference's defaults (balloon.py:156-173,183,200); returns BLE_OK or BLE_E_INVALID_ARG (v == NULL). */
int ble_vehicle_default(ble_vehicle* v);

/* hipError_t (as int) of the calling thread's most recent launch through this library; 0 = success.
 * Diagnostic companion of BLE_E_LAUNCH. */
int ble_last_hip_error(void);

/* Which form of the transition kernel ble_step_f32 / ble_step_n_f32 launch: 0 = automatic (by batch size, above), 1 = one
 * lane per environment, 4 = four wavefronts per environment.  (2 = two wavefronts per environment exists in experiment builds
 * only -- profiles/build_variant.sh -DBLE_WITH_PAIR_FORM; the product library answers BLE_E_INVALID_ARG since ABI 5: the form was
 * never selected and measured slower at every batch size.)
 * Process-global, thread-safe; takes effect with the next launch.  Returns the previous setting (>= 0) or
 * BLE_E_INVALID_ARG.  The initial value is 0, or what BLE_STEP_SPLIT (0 -> one lane, 1 / 4, 2) in the process environment
 * says when the library first looks at it -- once, not per launch.  (ABI 4; ABI 3 re-read the variable on every launch.) */
int ble_set_step_form(int waves_per_env);

/* Number of visible HIP devices (>= 0) or BLE_E_NO_DEVICE. */
int ble_device_count(void);

/*
 * One agent step (180 s = `substeps` x 10 s) for n environments.  `substeps` = time_delta / stride of
 * Balloon.simulate_step, 1 .. BLE_MAX_SUBSTEPS: the reference's 18, and up to 10 minutes per step, are held to
 * the parity bar (tests/test_gpu_parity.py); beyond that the per-step solar interpolation and the float32
 * accumulators of the state drift past 1e-5 (measured at 120), so longer steps are refused, not approximated.
 * Replaces BalloonArena.step (env/balloon_arena.py:184-202) up to, not including, the
 * feature constructor:
 *     wind = WindField.get_ground_truth(x, y, pressure, time_elapsed)   wind_field.py:125-145
 *          = GridBasedWindField.get_forecast(...) + noise               grid_based_wind_field.py:70-94
 *     Balloon.simulate_step(wind, atmosphere, action, 3 min, 10 s)     balloon.py:263-328
 * and BalloonEnv.step's reward / terminal (env/balloon_env.py:172-186,
 * perciatelli_reward_function :44-102).
 *
 *   st            state, mutated in place
 *   action        n bytes, 0 DOWN / 1 STAY / 2 UP (control.py:22-26).  Not range-checked on the device (the reference's
 *                 AltitudeControlCommand(3) raises on the host, and so does this package's single-environment facade):
 *                 any other value flies like STAY and is stored in last_command as given
 *   wind_grid     BLE_GRID_FLOATS floats, row-major (x, y, pressure, time, uv) = the
 *                 reference's `field` ndarray (21,21,10,9,2)
 *   grid_env_stride  0: one grid shared by all envs; otherwise env i reads
 *                 wind_grid + i * grid_env_stride (floats) -- per-env forecasts
 *   noise_uv      optional n x 2 additive wind noise [m/s] (the SimplexWindNoise term,
 *                 simplex_wind_noise.py; NULL = 0)
 *   reward        n floats out;  terminal  n bytes out (status != OK after the step)
 *   effective_action  optional n bytes out: the action after the three safety layers
 *   err_flags     optional device uint32, BLE_FLAG_* OR-ed in
 *   active_count  optional device uint64[BLE_COUNT_SLOTS]: the number of envs that were
 *                 actually stepped (status == OK on entry) is ADDED, spread over the slots
 *                 (one same-address atomic per wave would serialise 1 024 waves for ~11 us);
 *                 the caller sums the slots
 * Envs whose status != OK on entry are skipped: state untouched, reward 0, terminal 1
 * (the reference raises AssertionError, balloon.py:288-290; the host mirror does too).
 */
int ble_step_f32(const ble_state_f32* st, const uint8_t* action, const float* wind_grid,
                 int64_t grid_env_stride, const float* noise_uv, float* reward, uint8_t* terminal,
                 uint8_t* effective_action, uint32_t* err_flags, unsigned long long* active_count,
                 int64_t n, int substeps, void* stream);

/*
 * The wind-noise generator of a fused rollout: the arguments of ble_wind_noise_f32 (below) that do not change from
 * step to step.  One noise field per (seed, environment index, episode[i]).
 */
typedef struct ble_noise_gen {
  unsigned long long seed;
  const uint32_t* episode;  /* optional device uint32[n]: the per-environment episode counters ble_reset_f32 maintains (NULL = 0) */
  uint32_t* harmonic_cache; /* optional, as for ble_wind_noise_f32: [BLE_NOISE_CACHE_ROWS][n] words */
  int64_t env_offset;       /* ABI 4: index of this call's environment 0 in the GLOBAL batch (0 on one GPU; the shard's first
                               environment on a rank of a sharded run).  The noise field of environment i is keyed by
                               (seed, env_offset + i, episode[i]): a sharded batch flies the fields the unsharded one does */
} ble_noise_gen;

/*
 * `n_steps` consecutive agent steps in ONE kernel launch: the state stays in registers
 * between the steps (loaded once, stored once); per step only the action is read and
 * reward / terminal are written.  action / reward / terminal are [n_steps][n] row-major;
 * active_count, if given, is [n_steps][BLE_COUNT_SLOTS].  Same semantics per step as
 * ble_step_f32.
 *   noise   NULL: every step flies in the forecast (WindField.get_forecast; noise term 0, SURVEY 8(d)'s bench
 *           definition).  Otherwise the reference's WindField.get_ground_truth (wind_field.py:125-145): at every
 *           step the SimplexWindNoise term is evaluated inside the kernel at the pre-step (x, y, pressure, elapsed)
 *           with the generator `noise` describes -- bit for bit what n_steps rounds of ble_wind_noise_f32(mode 0)
 *           followed by ble_step_f32(noise_uv) produce (tests/test_gpu_parity.py).
 */
int ble_step_n_f32(const ble_state_f32* st, const uint8_t* action, const float* wind_grid,
                   int64_t grid_env_stride, const ble_noise_gen* noise, float* reward, uint8_t* terminal,
                   uint32_t* err_flags, unsigned long long* active_count, int64_t n, int substeps, int n_steps,
                   void* stream);

/*
 * Episode reset on the device for the environments with mask[i] != 0 (mask NULL = all).
 * Replaces BalloonArena.reset's balloon part (env/balloon_arena.py:161-182,228-268):
 *   sample != 0  draw alpha, start time, position, centre lat/lng, pressure, upwelling IR with the
 *                reference's distributions (utils/sampling.py:37-152) from a Philox4x32-10 stream
 *                keyed by (seed, env index, episode[i]); episode[i] (optional device uint32[n]) is
 *                then incremented.  (The reference's JAX threefry streams are not reproduced.)
 *   sample == 0  keep x, y, pressure, center_lat/lng_deg, upwelling_infrared, alpha, start_unix.
 * then stable_init.cold_start_to_stable_params (env/balloon/stable_init.py:132-157),
 * PowerSafetyLayer.__init__'s sunrise/sunset search (env/balloon/power_safety.py:40-48 ->
 * env/balloon/solar.py:432-483), battery 2905.6 Wh, clocks 0, FSMs NOMINAL, status OK.
 * Writes the per-episode "constants" of `st` too when sample != 0 (they are const only to
 * ble_step_f32).  The wind field is reset by the caller (new grid pointer / contents).
 */
int ble_reset_f32(const ble_state_f32* st, const uint8_t* mask, unsigned long long seed,
                  uint32_t* episode, int sample, uint32_t* err_flags, int64_t n, void* stream);
/* The same for a SHARD of a larger batch (ABI 4): environment i of this call is environment env_offset + i of the global
 * batch and draws from the Philox stream (seed, env_offset + i, episode[i]) -- the union of the shards' resets is the reset
 * of the unsharded batch, whatever the sharding.  ble_reset_f32 is this with env_offset = 0. */
int ble_reset_at_f32(const ble_state_f32* st, const uint8_t* mask, unsigned long long seed,
                     uint32_t* episode, int sample, uint32_t* err_flags, int64_t env_offset, int64_t n, void* stream);

/*
 * GridBasedWindField.get_forecast (grid_based_wind_field.py:70-94,145-187) for n query
 * points: clamp, time boomerang, float32 query packing, 16-corner interpolation.
 */
int ble_forecast_f32(const float* wind_grid, int64_t grid_env_stride, const float* x_m,
                     const float* y_m, const float* pressure, const int32_t* elapsed_s, float* u,
                     float* v, int64_t n, void* stream);

/*
 * GridBasedWindField.get_forecast_column (grid_based_wind_field.py:96-132): for each of
 * n (x, y, elapsed) columns, the forecast at `n_levels` shared pressure levels.
 * out_uv is [n][n_levels][2].
 */
int ble_forecast_column_f32(const float* wind_grid, int64_t grid_env_stride, const float* x_m,
                            const float* y_m, const int32_t* elapsed_s, const float* levels_pa,
                            int n_levels, float* out_uv, int64_t n, void* stream);

/*
 * Observation for n environments: PerciatelliFeatureConstructor.observe + get_features
 * (env/features.py:301-330,400-581) with its WindGP (env/wind_gp.py:90-241: Matern nu = 0.5,
 * refit on the observations of the last 6 h) and get_pressure_range
 * (env/balloon/pressure_range_builder.py:203-275).  One workgroup per environment.
 *
 *   noise_uv     optional [n][2]: measured wind minus forecast at the balloon (the reference's
 *                WindGP.observe error term, wind_gp.py:118-123); NULL = 0
 *   reset_mask   optional [n]: != 0 starts a new episode's history (the reference builds a new
 *                feature constructor in BalloonArena.reset, balloon_arena.py:171-177)
 *   hist         per-env ring of the last BLE_GP_CAPACITY observations, caller-allocated device
 *                memory, zero-initialised `count`
 *   append       1: observe() then get_features(); 0: get_features() on the existing history
 *   obs          [n][BLE_OBS_DIM] float32 out
 */
#define BLE_OBS_DIM 1099
#define BLE_GP_CAPACITY 128
#define BLE_GP_CHOL_STRIDE 7620 /* 120 * 121 / 2 doubles (packed factor) + 120 (the drop vector of the next slide) + 2 * 120 (zeta_u / d, zeta_v / d) */
typedef struct ble_gp_history_f32 {
  float* xyp;         /* [n][BLE_GP_CAPACITY][3]  x m, y m, pressure Pa */
  int32_t* elapsed_s; /* [n][BLE_GP_CAPACITY]     time_elapsed of the observation */
  float* err_uv;      /* [n][BLE_GP_CAPACITY][2]  measured - forecast, m/s */
  int32_t* count;     /* [n] observations appended this episode; ring slot = count % BLE_GP_CAPACITY */
  double* chol;       /* optional [n][BLE_GP_CHOL_STRIDE]: the current window's K + noise = Lt D Lt^T (unit-lower Lt,
                         d on the diagonal, packed lower triangle, 7260 doubles) followed by the drop vector
                         p = L22^-1 l21 of the next slide (120 doubles) and zeta / d for the two error components
                         (zeta = Lt^-1 y, 2 x 120 doubles), carried from call to call so that the
                         per-step refit of the reference (wind_gp.py:186-188) becomes an O(n^2) slide;
                         opaque to the caller; NULL = refit in LDS every call */
  int32_t* n_chol;    /* [n] rows of `chol` in use (required when chol != NULL), zero-initialised */
  int64_t chol_stride; /* doubles between the slabs of consecutive environments in `chol`: >= BLE_GP_CHOL_STRIDE AND EVEN
                          when chol != NULL (the kernel moves the slab as 16-byte double2 pairs, so every slab must start
                          16-byte aligned: `chol` itself 16-byte aligned, the stride a multiple of two doubles).  A smaller
                          or odd value is rejected with BLE_E_INVALID_ARG -- the kernel would write past the caller's
                          allocation or fault on a misaligned pair; ignored when chol == NULL */
} ble_gp_history_f32;
int ble_observe_f32(const ble_state_f32* st, const float* wind_grid, int64_t grid_env_stride,
                    const float* noise_uv, const uint8_t* reset_mask, const ble_gp_history_f32* hist,
                    int append, float* obs, uint32_t* err_flags, int64_t n, void* stream);
/* ... for a forecast that is NOT a grid (ABI 5).  The reference's feature constructor takes any wind_field.WindField
 * (features.py:290-299) and asks it for the column above the balloon, forecast.get_forecast_column(x, y, 181 levels, elapsed)
 * (features.py:499-503 -> wind_gp.py:218-222) -- e.g. the SimpleStaticWindField of its unit tests (wind_field.py:149-184), a step
 * function of pressure that no (21, 21, 10, 9) grid reproduces.
 *   forecast_levels  optional [n][181][2] float32: the caller's forecast (u, v) [m/s] at the levels 5 000 + 50 k Pa, k = 0 .. 180, at each
 *                    environment's position and time.  NULL: the column comes from wind_grid, as in ble_observe_f32 (which is this
 *                    call with NULL).  wind_grid must be a valid grid either way (its column is then computed and not used). */
int ble_observe_forecast_f32(const ble_state_f32* st, const float* wind_grid, int64_t grid_env_stride, const float* forecast_levels,
                             const float* noise_uv, const uint8_t* reset_mask, const ble_gp_history_f32* hist,
                             int append, float* obs, uint32_t* err_flags, int64_t n, void* stream);

/*
 * Tail of the wind-field VAE decoder (generative/vae.py:149-186, Decoder.__call__ after the
 * last Dense layer): n sets of 7 x 7 x 90 flow fields -> half-pixel linear resize to 23 x 23
 * -> central differences -> n wind grids [21][21][10][9][2] (grid_env_stride = 79 380 floats).
 * The four Dense layers before it are plain GEMMs (rocBLAS/hipBLASLt through torch.matmul).
 * n < 2^31 per call (one workgroup per grid).
 */
int ble_decode_flow_fields_f32(const float* flow, float* wind_grid, int64_t n, void* stream);

/*
 * SimplexWindNoise.get_wind_noise (env/simplex_wind_noise.py:214-259) for n environments at their
 * positions: noise_uv [n][2] in m/s, to be passed as `noise_uv` to ble_step_f32 / ble_observe_f32.
 * Five harmonics per component with the reference's weights and spacings; generator seeds and
 * offsets per (seed, env, episode[i]).  The 4-D noise primitive is NOT opensimplex 0.3's (absent,
 * unpinned): see csrc/ble_noise.h.  mode 1 is a test probe of the raw primitive.
 *   harmonic_cache  optional (may be NULL), opaque: [BLE_NOISE_CACHE_ROWS][n] 32-bit words, zero-initialised by the caller.
 *                   The reference draws a harmonic's generator seed and offsets once per reset and keeps them in its
 *                   NoisyWindHarmonic objects (simplex_wind_noise.py:97-114); this is where they are kept here, keyed by
 *                   (seed, episode[i]) -- an entry drawn for another key is redrawn (50 Philox draws) and stored.
 *                   NULL: redrawn by every call.  Same values either way.
 */
#define BLE_NOISE_CACHE_ROWS 53
int ble_wind_noise_f32(const float* x_m, const float* y_m, const float* pressure, const int32_t* elapsed_s,
                       unsigned long long seed, const uint32_t* episode, int mode, uint32_t* harmonic_cache,
                       float* noise_uv, int64_t n, void* stream);
/* ... for a shard whose environment 0 is environment env_offset of the global batch (ABI 4; see ble_reset_at_f32). */
int ble_wind_noise_at_f32(const float* x_m, const float* y_m, const float* pressure, const int32_t* elapsed_s,
                          unsigned long long seed, const uint32_t* episode, int mode, uint32_t* harmonic_cache,
                          float* noise_uv, int64_t env_offset, int64_t n, void* stream);

/*
 * Rows of the struct-of-arrays state as records (ABI 5): out[count][BLE_ROW_DOUBLES] doubles, row r = environment first + r, the 26
 * per-environment members of ble_state_f32 in the struct's order (x ... power_paused), every value converted exactly (float32, int32,
 * int64 seconds < 2^53 and bytes are all doubles).  What a host consumer that wants ONE balloon as an object -- BalloonArena.
 * get_balloon_state / get_simulator_state (env/balloon_arena.py:204-226), the evaluation loop's per-step read (eval/eval_lib.py:163) --
 * copies back in one transfer instead of 26.
 */
#define BLE_ROW_DOUBLES 26
int ble_state_rows_f64(const ble_state_f32* st, int64_t first, int64_t count, double* out, int64_t n, void* stream);

/* power_table.lookup (env/balloon/power_table.py:21-38). watts out as float. */
int ble_power_table_f32(const float* pressure_ratio, const float* state_of_charge, float* watts,
                        uint32_t* err_flags, int64_t n, void* stream);

/*
 * Function-level probes: run exactly the device functions ble_step_f32 uses, one lane
 * per element, so that each reference function can be parity-tested on its own.
 */
/* Atmosphere.at_pressure (standard_atmosphere.py:122-154): height [m], temperature [K] */
int ble_probe_atmosphere_f32(const float* alpha, const float* pressure, float* height,
                             float* temperature, uint32_t* err_flags, int64_t n, void* stream);
/* Atmosphere.at_height (standard_atmosphere.py:89-120): pressure [Pa] and temperature [K] at heights [m], float64 (ABI 5; until then the
 * host mirror inverted ble_probe_atmosphere_f32 by bracketing its float32 outputs: 1e-7).  Heights outside [-610 m, 84 852 m) set
 * BLE_FLAG_PRESSURE_RANGE (the reference asserts, :94-95). */
int ble_probe_atmosphere_at_height_f64(const float* alpha, const double* height_m, double* pressure, double* temperature,
                                       uint32_t* err_flags, int64_t n, void* stream);
/* solar_calculator at BalloonState.latlng (solar.py:43-174, spherical_geometry.py:44-76):
 * sin/cos of the refraction-corrected elevation, elevation [deg] and flux [W/m^2] */
int ble_probe_solar_f32(const float* center_lat_deg, const float* center_lng_deg, const float* x_m,
                        const float* y_m, const int64_t* unix_s, float* el_deg, float* flux,
                        int64_t n, void* stream);
/* BalloonState.latlng (balloon.py:217-220, spherical_geometry.py:44-76): latitude / longitude [deg, float64] of the point
 * (x, y) metres east / north of the centre -- the function the observation and the exact solar chain evaluate (ABI 4) */
int ble_probe_latlng_f64(const float* center_lat_deg, const float* center_lng_deg, const float* x_m, const float* y_m,
                         double* lat_deg, double* lng_deg, int64_t n, void* stream);
/* solar_atmospheric_attenuation + solar_power (solar.py:177-209,515-536) from el [deg] */
int ble_probe_solar_power_f32(const float* el_deg, const float* pressure, float* attenuation,
                              float* power_w, int64_t n, void* stream);
/* thermal.d_balloon_temperature_dt (thermal.py:175-230) */
int ble_probe_thermal_f32(const float* volume, const float* t_int, const float* t_amb,
                          const float* pressure, const float* el_deg, const float* flux,
                          const float* upwelling_ir, float* dtdt, uint32_t* err_flags, int64_t n,
                          void* stream);
/* calculate_superpressure_and_volume (balloon.py:552-609) */
int ble_probe_sp_volume_f32(const float* mols_air, const float* t_int, const float* pressure,
                            float* volume, float* superpressure, int64_t n, void* stream);
/* ... with the function's own vehicle arguments (mols_lift_gas, envelope_volume_base, envelope_volume_dv_pressure: balloon.py:552-558)
 * taken from `vehicle` (ABI 5; NULL = the defaults), and d_balloon_temperature_dt with its balloon_mass argument (thermal.py:175-181) =
 * vehicle->envelope_mass */
int ble_probe_sp_volume_vehicle_f32(const ble_vehicle* vehicle, const float* mols_air, const float* t_int, const float* pressure,
                                    float* volume, float* superpressure, int64_t n, void* stream);
int ble_probe_thermal_vehicle_f32(const ble_vehicle* vehicle, const float* volume, const float* t_int, const float* t_amb,
                                  const float* pressure, const float* el_deg, const float* flux,
                                  const float* upwelling_ir, float* dtdt, uint32_t* err_flags, int64_t n,
                                  void* stream);
/* acs.get_most_efficient_power / get_fan_efficiency / get_mass_flow (acs.py:44-68) */
int ble_probe_acs_f32(const float* pressure_ratio, float* power_w, float* efficiency,
                      float* mass_flow, int64_t n, void* stream);

/* The three safety layers of the transition one at a time, stateful across calls through `fsm` (in/out, one byte per
 * element; a fresh layer = 0):
 *   layer 0  AltitudeSafetyLayer.get_action (altitude_safety.py:63-111): value = pressure [Pa], alpha = the
 *            atmosphere's lapse-rate blend; fsm 0 NOMINAL 1 LOW 2 VERY_LOW;
 *   layer 1  EnvelopeSafetyLayer.get_action (envelope_safety.py:109-157, max superpressure 2380 Pa): value =
 *            superpressure [Pa]; fsm 0 NOMINAL 1 LOW_CRITICAL 2 LOW 3 HIGH 4 HIGH_CRITICAL;
 *   layer 2  PowerSafetyLayer.get_action (power_safety.py:52-126): value = battery charge [Wh]; clocks [n][3] =
 *            (now, sunrise + 30 min, sunset) in seconds from a common epoch -- the layer moves the two events on
 *            by whole days, in place; fsm = navigation_is_paused; night_load_w / capacity_wh as the reference's
 *            arguments (the transition passes 183.7 W and 3058.56 Wh).
 * effective_action [n] = the layer's answer to action [n] (0 DOWN 1 STAY 2 UP). */
int ble_probe_safety_f32(int layer, const uint8_t* action, const float* value, const float* alpha,
                         int32_t* clocks, double night_load_w, double capacity_wh, uint8_t* fsm,
                         uint8_t* effective_action, uint32_t* err_flags, int64_t n, void* stream);
/* ABI 5: for layer 1 `alpha`, if not NULL, holds each element's maximum superpressure [Pa] (EnvelopeSafetyLayer.__init__'s argument,
 * envelope_safety.py:100-107); NULL = the reference vehicle's 2 380 Pa. */

/* The kernel's own fp64 primitives (reciprocal / rsqrt seeds and refinements, log, exp,
 * sincos), element-wise on device doubles.  op: 0 rcp seed, 1 rcp, 2 rsq seed, 3 rsqrt,
 * 4 sqrt, 5 log, 6 exp, 7 sin, 8 cos.  Test-only. */
int ble_probe_f64_prims(const double* x, double* y, int op, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BLE_ABI_H_ */
