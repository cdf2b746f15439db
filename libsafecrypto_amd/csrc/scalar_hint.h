// scalar_hint.h - keep_scalar(a, b): an opaque copy of two wave-uniform pointers that pins them in scalar registers at this point, so that the loads
// through them are not hoisted out of a persistent loop (they feed twiddle sets that must die before the loop body's big register arrays are live:
// dil_yaw_kernel, dil_cfin_kernel).  A header of its own because it is the one construct of those kernels with no meaning off the GPU: the CPU
// emulation (tests/emu/shim) supplies an empty one.
#pragma once
#include <stdint.h>

__device__ __forceinline__ void keep_scalar(const uint32_t *&a, const uint32_t *&b) { asm volatile("" : "+s"(a), "+s"(b)); }
