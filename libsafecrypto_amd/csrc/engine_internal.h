// engine_internal.h - host-side structures shared by the translation units of libsc_hip.so.
#pragma once
#include "../../include/sc_hip.h"
#include "../../include/sc_hip_bench.h"
#include "modarith.h"

#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include <vector>
#include "cdt_guide.h"

namespace sc {
int fail(const std::string &m);      // records the message for sc_hip_last_error, returns SC_HIP_FAILURE
// set by the multi-device driver on its worker threads: the caller's buffers are pinned ONCE for all workers by the driver (multi.cpp); the per-device
// host entry must not register its slice as well - neighbouring slices share a page, and the workers would register and release overlapping pages
// concurrently
extern thread_local bool t_skip_host_pin;
void cdt_build_guide(const std::vector<uint64_t> &cdf, CdtGuide *g, std::vector<uint16_t> *guide, std::vector<uint64_t> *probe_table);
size_t cdt_lds_bytes(const sc_hip_cdt *c);      // table + padding + guide: the dynamic LDS of a kernel that samples from c
}

namespace sc {
// Copies between the caller's (pageable) memory and the device.  The HIP runtime moves a pageable buffer of more than 1 MiB by pinning the caller's
// pages in place for the duration of the copy ("HSA Copy Using Pinned resource" in its log) and anything smaller through its own pinned staging
// buffer.  The in-place path is the one a long-lived process cannot rely on here: with it, one run in eight of the GPU test suite (heap buffers of
// 4-16 MiB allocated, copied and freed in a loop) died with "Memory access fault by GPU ... on address <a host heap address>" inside a later copy; with
// every copy staged, none in forty (profiles/r02_pageable_copy_fault.txt).  So the library never hands the runtime a pageable
// piece above the staging threshold: large buffers it pins itself, explicitly and for exactly the call (hipHostRegister / hipHostUnregister in
// host_roundtrip and the multi-device driver), everything else goes through these helpers.
constexpr size_t HOST_PIECE = (size_t)1 << 20;
inline hipError_t copy_in(void *dev, const void *host, size_t bytes)
{
    for (size_t o = 0; o < bytes; o += HOST_PIECE) {
        const hipError_t e = hipMemcpy((char *)dev + o, (const char *)host + o, bytes - o < HOST_PIECE ? bytes - o : HOST_PIECE, hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
inline hipError_t copy_out(void *host, const void *dev, size_t bytes)
{
    for (size_t o = 0; o < bytes; o += HOST_PIECE) {
        const hipError_t e = hipMemcpy((char *)host + o, (const char *)dev + o, bytes - o < HOST_PIECE ? bytes - o : HOST_PIECE, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
// the same on a stream (for a pageable buffer the runtime stages each piece before it returns, so this orders with the stream but does not overlap)
inline hipError_t copy_in_async(void *dev, const void *host, size_t bytes, hipStream_t s, bool pinned)
{
    if (pinned) return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s);
    for (size_t o = 0; o < bytes; o += HOST_PIECE) {
        const hipError_t e = hipMemcpyAsync((char *)dev + o, (const char *)host + o, bytes - o < HOST_PIECE ? bytes - o : HOST_PIECE, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
inline hipError_t copy_out_async(void *host, const void *dev, size_t bytes, hipStream_t s, bool pinned)
{
    if (pinned) return hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
    for (size_t o = 0; o < bytes; o += HOST_PIECE) {
        const hipError_t e = hipMemcpyAsync((char *)host + o, (const char *)dev + o, bytes - o < HOST_PIECE ? bytes - o : HOST_PIECE, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
}

namespace sc {
// Device memory that may hold key material, masks, noise or CSPRNG output is zeroed before it goes back to the allocator, as the reference zeroes
// its temporaries (SC_MEMZERO(sc->temp, ...), dilithium.c:1964-1975, kyber_kem.c:1014) and keys on free: every free in this library is a wiped free.
inline hipError_t wipe_free(void *p)
{
    if (!p) return hipSuccess;
    size_t sz = 0;
    if (hipMemPtrGetInfo(p, &sz) == hipSuccess && sz) (void)hipMemset(p, 0, sz);
    else (void)hipGetLastError();
    return hipFree(p);
}
// the same for a buffer that stays allocated: zero the whole allocation on `stream` behind the work that used it (end of a sign / decapsulation /
// key generation call: per-call secrets do not outlive the call)
inline void wipe_async(void *p, hipStream_t stream)
{
    if (!p) return;
    size_t sz = 0;
    if (hipMemPtrGetInfo(p, &sz) == hipSuccess && sz) (void)hipMemsetAsync(p, 0, sz, stream);
    else (void)hipGetLastError();
}

// scoped device temporary (key loads, host-buffer variants): freed on every exit path
struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)wipe_free(p); }
    int get(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4) == hipSuccess ? 0 : fail("device allocation failed"); }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};
}

namespace sc {
// grow-only device workspace kept by a scheme handle: repeated calls at the same batch size allocate nothing
struct Workspace {
    char *p = nullptr;
    size_t cap = 0, used = 0;
    int reserve(size_t bytes)
    {
        used = 0;
        if (bytes <= cap) return 0;
        if (p) (void)wipe_free(p);
        p = nullptr; cap = 0;
        if (hipMalloc((void **)&p, bytes) != hipSuccess) return fail("device workspace allocation failed");
        cap = bytes;
        return 0;
    }
    template <typename T> T *take(size_t count)                  // 256-byte aligned slices of the reservation
    {
        T *r = reinterpret_cast<T *>(p + used);
        used += (count * sizeof(T) + 255) & ~(size_t)255;
        return r;
    }
    void release() { if (p) (void)wipe_free(p); p = nullptr; cap = used = 0; }
    void wipe(hipStream_t s) { if (p && used) (void)hipMemsetAsync(p, 0, used < cap ? used : cap, s); }      // per-call secrets: zeroed behind the call's work
    // the same for the slices taken so far only (take the secret slices first, call this, then take the public ones: zeroing costs bandwidth)
    size_t mark() const { return used; }
    void wipe_upto(size_t m, hipStream_t s) { if (p && m) (void)hipMemsetAsync(p, 0, m < cap ? m : cap, s); }
};
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return sc::fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct sc_hip_ctx {
    int device;
    int cus;
};

struct sc_hip_plan {
    sc_hip_ctx *ctx;
    int32_t q;
    int n, L;
    int32_t g;
    int mode;                     // MODE_U24 / MODE_U32 / MODE_MAD / MODE_SMAD / MODE_S32 in use
    bool lazy24, lazy32;
    bool mad_ok;                  // the small-modulus multiply-add forms are safe for this (q, n)
    std::vector<int32_t> w, r;    // reference-format tables (host)
    uint32_t *d_tab[2];           // per mode: [zf | zfp | zi | zip], n entries each
    uint32_t *d_tab_s;            // MODE_SMAD (q = 7681 at n = 256; q < 14654 at n = 512, 1024): centred twiddles, signed companions; same layout
    sc::ModK k[2];
    uint32_t c_inv[2][4];         // scale = n^-1          : c0,c0p,c1,c1p
    uint32_t c_mont[2][4];        // scale = n^-1 * 2^32
    uint32_t c_inv_s[4];          // MODE_SMAD / MODE_S32: the n^-1 scale constants centred, signed companions
    uint32_t c_mont_s[4];         // MODE_S32: the same for n^-1 2^32
    bool smad_ok, s32_ok;
    uint32_t r2, r2p;             // 2^32 mod q (as a U32-mode Shoup pair)
    uint32_t *mv_ws;              // sc_hip_matvec scratch: NTT(y) and the matrix in internal order; guarded by mv_mu (host) and mv_event (device)
    size_t mv_cap;
    std::mutex mv_mu;
    void *mv_event;
    void *g_exec;                 // sc_hip_polymul_ntt_steps: cached executable graph and what it was captured for
    const int32_t *g_a, *g_b; int32_t *g_out; size_t g_batch; int g_steps;
};

namespace sc {
int polymul_ntt_bcast(sc_hip_plan *p, size_t batch, const int32_t *d_a, const int32_t *d_bhat, int32_t *d_out, void *stream);
}

struct sc_hip_cdt {
    sc_hip_ctx *ctx;
    float tail, sigma;
    int size;
    std::vector<uint64_t> cdf;
    uint64_t *d_cdf;              // size entries + CDT_PAD of 2^64 - 1 (g.S != 0: 2^64 - 1 also behind the first entry of the closing run, cdt_guide.h)
    uint16_t *d_guide;            // g.total bins (cdt_guide.h); g.S = 0: no usable guide, the kernels fall back to the halving search
    CdtGuide g;
};
