// Stand-in for libsafecrypto_amd/csrc/scalar_hint.h in the lane-by-lane emulation: pinning a pointer in scalar registers has no meaning on the CPU.
#pragma once
#include <stdint.h>
static inline void keep_scalar(const uint32_t *&, const uint32_t *&) {}
