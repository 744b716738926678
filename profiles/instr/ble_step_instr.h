// Per-wave clock marks and rare-path counters of ble_step_kernel for PROFILING builds (never part of the product library):
//   bash profiles/build_variant.sh step_timing '-DBLE_STEP_INSTR_HEADER="../../profiles/instr/ble_step_instr.h"'
// The profiling build re-uses the `active_count` argument as the mark buffer: int64 [workgroups][64], zeroed by the caller
// (the live-environment count is switched off).  profiles/step_wave_timing.py launches it and decodes the marks:
//   [0] wall clock (s_memrealtime, 100 MHz, chip-wide) at entry     [1] wall clock after the stores were acknowledged
//   [2..6] shader-clock (s_memtime) differences: state loads landed | ACS cubics built + barrier | per-episode constants |
//          the agent steps | stores issued + acknowledged;   [7] the whole wave
//   [8..39] shader-clock length of agent step k of the launch (k < 32)
//   [48..53] shader-clock sums over the launch of the sections of an agent step: atmosphere + safety layers | ephemeris |
//          wind blend | three solar nodes | the substep loop | status + reward   (BLE_STEP_TICK in ble_step_core.h)
//   [40..43] lanes that took a rare path, summed over the launch: 0 exact solar chain (a threshold within the fp32 floor),
//          1 layer transition crossed, 2 p and p +- 1 Pa straddle a transition, 3 atmosphere window above 21 km
// Marks 1 and 5 drain the memory counters (s_waitcnt 0) so that "loads landed" / "stores acknowledged" mean that.
#pragma once
#include <hip/hip_runtime.h>
__device__ unsigned long long* g_ble_step_dbg = nullptr;
#define BLE_STEP_COUNTS_LIVE 0
__device__ __forceinline__ void ble_step_tick(int i) {
  static __shared__ long long last_tick;
  const long long now_ = (long long)__builtin_readcyclecounter();
  if (i > 0 && g_ble_step_dbg && threadIdx.x == 0) g_ble_step_dbg[64 * (unsigned long long)blockIdx.x + 47 + i] += (unsigned long long)(now_ - last_tick);
  if (threadIdx.x == 0) last_tick = now_;
}
#define BLE_STEP_TICK(i) ble_step_tick(i)
#define BLE_STEP_EVENT(i) do { if (g_ble_step_dbg) atomicAdd(g_ble_step_dbg + 64 * (unsigned long long)blockIdx.x + 40 + (i), 1ull); } while (0)
#define BLE_STEP_INSTR_BEGIN() \
  long long smark[6] = {0, 0, 0, 0, 0, 0}; \
  if (threadIdx.x == 0 && blockIdx.x == 0) g_ble_step_dbg = active_count; \
  const long long swall0 = (long long)wall_clock64(); \
  smark[0] = (long long)__builtin_readcyclecounter(); \
  long long sstep = smark[0]
#define BLE_STEP_MARK(i) do { if ((i) == 1) __builtin_amdgcn_s_waitcnt(0); smark[i] = (long long)__builtin_readcyclecounter(); if ((i) == 3) sstep = smark[3]; } while (0)
#define BLE_STEP_STEP_DONE(k) \
  do { \
    const long long now_ = (long long)__builtin_readcyclecounter(); \
    if (threadIdx.x == 0 && active_count != nullptr && (k) < 32) active_count[64 * (long long)blockIdx.x + 8 + (k)] = (unsigned long long)(now_ - sstep); \
    sstep = now_; \
  } while (0)
#define BLE_STEP_INSTR_END() \
  do { \
    __builtin_amdgcn_s_waitcnt(0); \
    smark[5] = (long long)__builtin_readcyclecounter(); \
    if (threadIdx.x == 0 && active_count != nullptr) { \
      long long* d = reinterpret_cast<long long*>(active_count) + 64 * (long long)blockIdx.x; \
      d[0] = swall0; d[1] = (long long)wall_clock64(); \
      for (int q = 1; q < 6; ++q) d[1 + q] = smark[q] - smark[q - 1]; \
      d[7] = smark[5] - smark[0]; \
    } \
  } while (0)
