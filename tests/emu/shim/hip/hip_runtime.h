// Stand-in for <hip/hip_runtime.h> when a kernel header under libsafecrypto_amd/csrc/ is compiled as HOST code for the lane-by-lane emulation
// (tests/emu/wave_emu.h; test infrastructure only).  `__asm("llvm.amdgcn...")` labels on function declarations (csrc/modarith.h) are dropped so that
// the declarations bind to the host definitions in wave_emu.h.
#pragma once
#include "../../wave_emu.h"
#define __asm(x)
#ifdef EMU_HOST_API
#include "../../hip_host_emu.h"
#endif
