// Instrumentation of ble_observe_kernel for PROFILING builds (never part of the product library):
//   profiles/build_variant.sh <out.so> -DBLE_OBS_INSTR_HEADER='"../../profiles/instr/ble_observe_instr.h"' [-DBLE_OBS_TIMING]
//                                      [-DBLE_OBS_SOLO] [-DBLE_OBS_PHASE_PROFILE]
// csrc/ble_observe.h calls the hooks below at fixed places; its product build defines all of them empty.
//   BLE_OBS_TIMING         in-kernel cycle marks (s_memtime), written into the tail of the observation vector by
//                          BLE_OBS_INSTR_END (profiles/time_observe.py decodes them)
//   BLE_OBS_SOLO           9000 doubles of extra LDS: one workgroup per CU
//   BLE_OBS_PHASE_PROFILE  `append` carries a stop code in bits 8..: the launch returns after that phase, so that
//                          per-phase instruction counts are differences of PMC runs (profiles/obs_phases.py)
#pragma once

#ifdef BLE_OBS_TIMING
#define BLE_OBS_INSTR_SHARED float role_t[4], sw1[5], role_t0[4], blk1[8];
#else
#define BLE_OBS_INSTR_SHARED
#endif

#ifdef BLE_OBS_SOLO
#define BLE_OBS_INSTR_SOLO()                                              \
  __shared__ double solo_pad[9000];                                       \
  if (threadIdx.x == 0 && n < 0) solo_pad[obs != nullptr] = 1.0;          \
  if (n < 0) obs[0] = (float)solo_pad[1];
#else
#define BLE_OBS_INSTR_SOLO()
#endif

#ifdef BLE_OBS_TIMING
#define BLE_OBS_INSTR_TIMERS()                                            \
  long long tmark[12]; int nmark = 0;                                     \
  long long tsub[5] = {0, 0, 0, 0, 0};                                    \
  long long tsw[6] = {0, 0, 0, 0, 0, 0};                                  \
  long long tblk[8] = {0, 0, 0, 0, 0, 0, 0, 0};                           \
  long long role_begin = 0;                                               \
  (void)tsub; (void)tsw; (void)tblk; (void)role_begin;
#define BLE_SW(i) do { tsw[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define BLE_MARK() do { tmark[nmark++] = (long long)__builtin_readcyclecounter(); } while (0)
#define BLE_SUB(i) do { tsub[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define BLE_BLK(i) do { if (kFirst) tblk[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define BLE_ROLE_ENTRY_DONE() do { if ((tid & 63) == 0) sh.role_t0[tid >> 6] = (float)((long long)__builtin_readcyclecounter() - tmark[0]); } while (0)
#define BLE_ROLE_BEGIN() do { role_begin = (long long)__builtin_readcyclecounter(); } while (0)
#define BLE_ROLE_END() do { if ((tid & 63) == 0) sh.role_t[tid >> 6] = (float)((long long)__builtin_readcyclecounter() - role_begin); } while (0)
#define BLE_OBS_INSTR_END()                                                                                              \
  do {                                                                                                                   \
    if (tid == 64) { for (int k = 0; k < 5; ++k) sh.sw1[k] = (float)(tsw[k] - tmark[5]); for (int k = 0; k < 8; ++k) sh.blk1[k] = (float)(tblk[k] - tmark[5]); } \
    __syncthreads();                                                                                                     \
    BLE_MARK();                                                                                                          \
    if (tid == 0) {                                                                                                      \
      for (int k = 0; k < 5; ++k) { out[kObsDim - 34 + k] = (float)(tsw[k] - tmark[5]); out[kObsDim - 29 + k] = sh.sw1[k]; } \
      for (int k = 1; k < nmark; ++k) out[kObsDim - 12 + k] = (float)(tmark[k] - tmark[k - 1]);                          \
      for (int k = 0; k < 3; ++k) out[kObsDim - 16 + k] = sh.role_t[k];                                                  \
      out[kObsDim - 13] = sh.role_t[3];                                                                                  \
      for (int k = 0; k < 4; ++k) out[kObsDim - 38 + k] = sh.role_t0[k];                                                 \
      for (int k = 0; k < 8; ++k) out[kObsDim - 46 + k] = sh.blk1[k];                                                    \
      for (int k = 0; k < 4; ++k) out[kObsDim - 20 + k] = (float)(tsub[k] - tmark[0]);                                   \
      out[kObsDim - 4] = (float)n_tiles; out[kObsDim - 3] = (float)n_reach; out[kObsDim - 2] = (float)(n_tiles > 8);     \
    }                                                                                                                    \
  } while (0)
#else
#define BLE_OBS_INSTR_TIMERS()
#define BLE_SW(i) do {} while (0)
#define BLE_MARK() do {} while (0)
#define BLE_SUB(i) do {} while (0)
#define BLE_BLK(i) do {} while (0)
#define BLE_ROLE_ENTRY_DONE() do {} while (0)
#define BLE_ROLE_BEGIN() do {} while (0)
#define BLE_ROLE_END() do {} while (0)
#define BLE_OBS_INSTR_END() do {} while (0)
#endif

#ifdef BLE_OBS_PHASE_PROFILE
#define BLE_OBS_INSTR_STOPCODE() const int stop_after = append >> 8; append &= 1;
#define BLE_STOP(k) do { if (stop_after == (k)) return; } while (0)
#else
#define BLE_OBS_INSTR_STOPCODE()
#define BLE_STOP(k) do {} while (0)
#endif

// (declarations: this one is a statement list, not a do-while)
#define BLE_OBS_INSTR_BEGIN() BLE_OBS_INSTR_SOLO() BLE_OBS_INSTR_TIMERS() BLE_OBS_INSTR_STOPCODE() (void)0
