// launch.h -- kernel launch macro of the encoder kernels.
//
// Per-kernel timing (bench.py's roofline leg, ppasr_profile_*): while a profiling scope is active on the calling
// thread, every launch carries its own (start, stop) HIP events ATTACHED TO THE DISPATCH (hipExtLaunchKernelGGL), so
// hipEventElapsedTime(start, stop) is the kernel's own begin-to-end time -- the same two time stamps rocprofv3's kernel
// trace reads.  (Bracketing a launch with two hipEventRecord calls instead puts two extra barrier packets on the
// stream: 17-22 us per kernel on this stack, which made the live figure disagree with the rocprofv3 summary of the
// same command.)  Outside a scope the macro is a plain hipLaunchKernelGGL.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

namespace ppasr {
struct LaunchProf {
  // hands out the event pair of the next launch (`fn` = the kernel's host-side function pointer, which names it);
  // nullptr = no scope active
  void (*next)(void* ctx, const void* fn, hipEvent_t* start, hipEvent_t* stop) = nullptr;
  void* ctx = nullptr;
};
extern thread_local LaunchProf g_launch_prof;  // defined in capi.hip
}  // namespace ppasr

#define PPASR_LAUNCH(kernel, grid, block, lds, st, ...)                                       \
  do {                                                                                        \
    if (ppasr::g_launch_prof.next) {                                                          \
      hipEvent_t _ps = nullptr, _pe = nullptr;                                                \
      ppasr::g_launch_prof.next(ppasr::g_launch_prof.ctx, reinterpret_cast<const void*>(kernel), &_ps, &_pe); \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, _ps, _pe, 0, __VA_ARGS__);          \
    } else {                                                                                  \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                          \
    }                                                                                         \
  } while (0)
