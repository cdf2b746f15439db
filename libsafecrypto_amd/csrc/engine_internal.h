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
// set by the multi-device driver on its worker threads: the caller's buffers are pinned ONCE for all workers by the driver (multi.cpp) and the per-device
// host entry uses the driver's registrations (t_driver_pins) instead of registering its slice - neighbouring slices share a page, and the workers
// would register and release overlapping pages concurrently
struct HostPins;
extern thread_local bool t_skip_host_pin;
extern thread_local const HostPins *t_driver_pins;
void cdt_build_guide(const std::vector<uint64_t> &cdf, CdtGuide *g, std::vector<uint16_t> *guide, std::vector<uint64_t> *probe_table);
size_t cdt_lds_bytes(const sc_hip_cdt *c);      // table + padding + guide: the dynamic LDS of a kernel that samples from c
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return sc::fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace sc {
#ifdef __HIPCC__
// x mod q for 0 <= x < 2^32 without the divider (the epilogue kernels reduced every coefficient with a runtime `%`: ~30 instructions for 32 bits, ~150 for
// the 64-bit forms): one reciprocal per thread, then a multiply-high, a multiply-subtract and two conditional subtractions per value.
// m = floor((2^32 - 1) / q) underestimates 2^32 / q by less than 1 + 1/q, so the quotient estimate is short by at most 2.
struct FastMod {
    uint32_t q, m;
    __device__ explicit FastMod(uint32_t q_) : q(q_), m(0xFFFFFFFFu / q_) {}
    __device__ __forceinline__ uint32_t operator()(uint32_t x) const
    {
        uint32_t r = x - __umulhi(x, m) * q;
        r = r < r - q ? r : r - q;
        return r < r - q ? r : r - q;
    }
};
// index = item * per + part for a power of two `per`: the shift instead of a 64-bit division per thread
__device__ __forceinline__ int pow2_shift(int per) { return 31 - __builtin_clz((unsigned)per); }
#endif
}

namespace sc {
// Copies between the caller's (pageable) memory and the device.  The HIP runtime moves a pageable buffer of more than 1 MiB by pinning the caller's
// pages in place for the duration of the copy ("HSA Copy Using Pinned resource" in its log) and anything smaller through its own pinned staging
// buffer.  The in-place path is the one a long-lived process could not rely on in round 2: with it, 20 of 101 sessions of the GPU test suite (heap buffers
// of 4-16 MiB allocated, copied and freed in a loop) died with "Memory access fault by GPU ... on address <a host heap address>" inside a later copy; with
// every copy staged, none of 76 (profiles/r02_pageable_copy_fault.txt; round 3 did not see it in 60 sessions either way: profiles/r03_host_pin_bisect.txt).  So the library never hands the runtime a pageable
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
// Explicit pinning of caller memory for the length of one call.  Only WHOLE pages that lie entirely inside the caller's buffers are registered: a heap
// buffer's first and last page are shared with whatever the allocator put beside it - possibly memory the runtime (or another thread of this library)
// has pinned or will pin on its own account - and registering / unregistering a range that overlaps such a page is the one thing the explicit path
// must never do (the root cause of the intermittent "Memory access fault by GPU" was never isolated; the candidates were the two
// places that registered unaligned caller ranges).  Ranges that overlap or touch (in / out buffers of an in-place or partially overlapping call) are
// merged and registered once.  The edge fragments (< one page at each end) and anything too small to be worth a registration move through the
// runtime's staged path (copy_in_async / copy_out_async below with pinned = false).
struct HostPins {
    struct Range { uintptr_t lo, hi; };
    std::vector<Range> want, held;
    unsigned flags = hipHostRegisterDefault;
    static constexpr uintptr_t PAGE = 4096;
    static constexpr size_t MIN_BYTES = (size_t)1 << 20;
    static int &mode() { static int m = 1; return m; }                    // measurement hook (sc_hip_bench_set_host_pin): 0 off, 1 whole interior pages, 2 round 2's unaligned ranges
    static bool enabled() { return mode() != 0; }
    HostPins() = default;
    HostPins(const HostPins &) = delete;
    HostPins &operator=(const HostPins &) = delete;
    void add(const void *p, size_t bytes) { if (p && bytes) want.push_back({ (uintptr_t)p, (uintptr_t)p + bytes }); }
    void commit(unsigned register_flags = hipHostRegisterDefault)
    {
        flags = register_flags;
        if (!enabled() || t_skip_host_pin) return;
        // merge what overlaps or abuts; two DISJOINT ranges that merely share a page stay separate - the gap between them is somebody else's memory
        // (allocator metadata, a foreign heap block) and must not become an interior page of a union - and the later one gives the shared
        // page up: its start moves past it, so that page is staged, not registered
        for (size_t i = 0; i < want.size(); i++)
            for (size_t j = i + 1; j < want.size();) {
                if (want[j].lo <= want[i].hi && want[j].hi >= want[i].lo) {
                    want[i].lo = want[i].lo < want[j].lo ? want[i].lo : want[j].lo; want[i].hi = want[i].hi > want[j].hi ? want[i].hi : want[j].hi;
                    want.erase(want.begin() + (long)j); j = i + 1;
                } else j++;
            }
        for (size_t i = 0; i < want.size(); i++)
            for (size_t j = 0; j < want.size(); j++) {
                if (i == j || want[j].lo < want[i].hi) continue;                    // j lies behind i
                const uintptr_t page_end = (want[i].hi + PAGE - 1) & ~(PAGE - 1);   // end of the page i's last byte is on
                if (want[j].lo < page_end) want[j].lo = page_end < want[j].hi ? page_end : want[j].hi;
            }
        for (const Range &r : want) {
            const bool raw = mode() == 2;                                  // bisecting only: the caller's range as it is, edge pages included
            const uintptr_t lo = raw ? r.lo : (r.lo + PAGE - 1) & ~(PAGE - 1), hi = raw ? r.hi : r.hi & ~(PAGE - 1);
            if (hi <= lo || hi - lo < MIN_BYTES) continue;
            if (hipHostRegister((void *)lo, hi - lo, flags) == hipSuccess) held.push_back({ lo, hi });
            else (void)hipGetLastError();                                  // already pinned by the caller, or not registrable: staged / direct by attribute
        }
    }
    // the part of [p, p + bytes) that lies inside a registered range (at most one: the ranges are disjoint); false if none
    bool inside(const void *p, size_t bytes, uintptr_t *lo, uintptr_t *hi) const
    {
        const uintptr_t a = (uintptr_t)p, b = a + bytes;
        for (const Range &r : held)
            if (a < r.hi && b > r.lo) { *lo = a > r.lo ? a : r.lo; *hi = b < r.hi ? b : r.hi; return true; }
        return false;
    }
    ~HostPins() { for (const Range &r : held) (void)hipHostUnregister((void *)r.lo); }
};
// a transfer between a caller buffer and the device: the part inside `pins` (or all of it when the caller allocated the buffer pinned) goes direct,
// the rest in staged pieces
inline hipError_t copy_in_pinned(void *dev, const void *host, size_t bytes, hipStream_t s, const HostPins *pins, bool caller_pinned)
{
    if (caller_pinned) return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s);
    uintptr_t lo, hi;
    if (!pins || !pins->inside(host, bytes, &lo, &hi)) return copy_in_async(dev, host, bytes, s, false);
    const uintptr_t a = (uintptr_t)host, b = a + bytes;
    hipError_t e = hipSuccess;
    if (lo > a) e = copy_in_async(dev, host, lo - a, s, false);
    if (e == hipSuccess) e = hipMemcpyAsync((char *)dev + (lo - a), (const void *)lo, hi - lo, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && b > hi) e = copy_in_async((char *)dev + (hi - a), (const void *)hi, b - hi, s, false);
    return e;
}
inline hipError_t copy_out_pinned(void *host, const void *dev, size_t bytes, hipStream_t s, const HostPins *pins, bool caller_pinned)
{
    if (caller_pinned) return hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
    uintptr_t lo, hi;
    if (!pins || !pins->inside(host, bytes, &lo, &hi)) return copy_out_async(host, dev, bytes, s, false);
    const uintptr_t a = (uintptr_t)host, b = a + bytes;
    hipError_t e = hipSuccess;
    if (lo > a) e = copy_out_async(host, dev, lo - a, s, false);
    if (e == hipSuccess) e = hipMemcpyAsync((void *)lo, (const char *)dev + (lo - a), hi - lo, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && b > hi) e = copy_out_async((void *)hi, (const char *)dev + (hi - a), b - hi, s, false);
    return e;
}
inline bool caller_allocated_pinned(const void *p)
{
    hipPointerAttribute_t at;
    if (p && hipPointerGetAttributes(&at, p) == hipSuccess) return at.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    return false;
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

namespace sc {
// Host-buffer forms of the batched scheme calls (the shape safecrypto_sign / safecrypto_verify / safecrypto_encapsulation have, for a batch): the
// caller's arrays are staged into device memory that lives for the call, the device entry runs on the null stream, the results come back.  These are
// conveniences for callers without device buffers of their own (the libsafecrypto-side glue, the multi-device driver) and are PCIe-bound: a BLISS
// signature comes with 33 KB of CSPRNG words.  Device temporaries are zeroed before they are freed (DevBuf).
template <class Call>
int sign_host(int device, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens,
              const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
              uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials, uint32_t *h_status, Call call)
{
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!h_sigs || !h_siglens || !h_status) return fail("sign_host: null output buffer");
    if (!h_tapes) return fail("sign_host: null tape buffer");
    if (!h_msgs && msg_stride) return fail("sign_host: null message buffer");
    HIPCHK(hipSetDevice(device));
    DevBuf m, ml, t, tl, sg, meta;
    if (m.get(batch * msg_stride + 8) || ml.get(batch * 4) || t.get(batch * tape_stride * 4 + 8) || tl.get(batch * 4) || sg.get(batch * sig_stride + 8) || meta.get(batch * 12))
        return SC_HIP_FAILURE;
    if (h_msgs && msg_stride) HIPCHK(copy_in(m.p, h_msgs, batch * msg_stride));
    if (h_mlens) HIPCHK(copy_in(ml.p, h_mlens, batch * 4));
    HIPCHK(copy_in(t.p, h_tapes, batch * tape_stride * 4));
    if (h_tape_lens) HIPCHK(copy_in(tl.p, h_tape_lens, batch * 4));
    uint32_t *d_siglens = meta.as<uint32_t>(), *d_trials = d_siglens + batch, *d_status = d_trials + batch;
    HIPCHK(hipMemset(sg.p, 0, batch * sig_stride));
    HIPCHK(hipMemset(meta.p, 0, batch * 12));
    const int rc = call(m.as<uint8_t>(), h_mlens ? ml.as<uint32_t>() : nullptr, t.as<uint32_t>(), h_tape_lens ? tl.as<uint32_t>() : nullptr,
                        sg.as<uint8_t>(), d_siglens, d_trials, d_status);
    if (rc != SC_HIP_SUCCESS) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    HIPCHK(copy_out(h_sigs, sg.p, batch * sig_stride));
    HIPCHK(copy_out(h_siglens, d_siglens, batch * 4));
    if (h_trials) HIPCHK(copy_out(h_trials, d_trials, batch * 4));
    HIPCHK(copy_out(h_status, d_status, batch * 4));
    return SC_HIP_SUCCESS;
}
template <class Call>
int verify_host(int device, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens,
                const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts, Call call)
{
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!h_sigs || !h_siglens || !h_verdicts) return fail("verify_host: null buffer");
    if (!h_msgs && msg_stride) return fail("verify_host: null message buffer");
    HIPCHK(hipSetDevice(device));
    DevBuf m, ml, sg, meta;
    if (m.get(batch * msg_stride + 8) || ml.get(batch * 4) || sg.get(batch * sig_stride + 8) || meta.get(batch * 8)) return SC_HIP_FAILURE;
    if (h_msgs && msg_stride) HIPCHK(copy_in(m.p, h_msgs, batch * msg_stride));
    if (h_mlens) HIPCHK(copy_in(ml.p, h_mlens, batch * 4));
    HIPCHK(copy_in(sg.p, h_sigs, batch * sig_stride));
    uint32_t *d_siglens = meta.as<uint32_t>(), *d_verdicts = d_siglens + batch;
    HIPCHK(copy_in(d_siglens, h_siglens, batch * 4));
    const int rc = call(m.as<uint8_t>(), h_mlens ? ml.as<uint32_t>() : nullptr, sg.as<uint8_t>(), d_siglens, d_verdicts);
    if (rc != SC_HIP_SUCCESS) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    HIPCHK(copy_out(h_verdicts, d_verdicts, batch * 4));
    return SC_HIP_SUCCESS;
}
}

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
    int32_t *d_sink;              // 4 KB nobody reads: store target of the lanes past the end of a ragged last group (ntt_kernel)
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
// The reference's CSPRNG for a SEEDED call (sc_hip_*_sign_seeded: message i signs with the generator create_csprng() would make of seed i - AES-256 CTR-DRBG,
// csrc/drbg_dev.h - instead of a tape the caller filled): 16-byte blocks [blk0, blk1) of the streams of rows d_list[0..count) (null: rows 0..count-1; d_count:
// the count on the device, `count` its bound) into d_out + row * out_stride_bytes + 16 * block; words != 0: prng_32 word order (the tape format).
// zero: the same rows and blocks zeroed (the per-call secrets do not outlive the call).
int drbg_fill(sc_hip_ctx *ctx, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len, const uint32_t *d_list, uint32_t count, const uint32_t *d_count,
              void *d_out, size_t out_stride_bytes, size_t blk0, size_t blk1, int words, bool zero, hipStream_t s);
}

namespace sc {
// batch_dev != null: the number of rows where only the device knows it (batch is then the host's bound; rows past the count are not touched)
int polymul_ntt_bcast(sc_hip_plan *p, size_t batch, const int32_t *d_a, const int32_t *d_bhat, int32_t *d_out, void *stream, const uint32_t *batch_dev = nullptr);
int polymul_ntt_bcast2(sc_hip_plan *p, size_t batch, const int32_t *d_a, const int32_t *d_b1hat, const int32_t *d_b2hat, int32_t *d_out1, int32_t *d_out2, void *stream,
                       const uint32_t *batch_dev = nullptr);      // two broadcast operands, two outputs, one forward transform
}

struct sc_hip_cdt {
    sc_hip_ctx *ctx;
    float tail, sigma;
    int size;
    std::vector<uint64_t> cdf;
    uint64_t *d_cdf;              // size entries + CDT_PAD of 2^64 - 1 (g.S != 0: 2^64 - 1 also behind the first entry of the closing run, cdt_guide.h)
    uint16_t *d_guide;            // g.total bins (cdt_guide.h); g.S = 0: no usable guide, the kernels fall back to the halving search
    CdtGuide g;
    bool w32 = false;             // a 32-bit precision table widened to the 64-bit form (sc_hip_cdt32_*): searched with widened 32-bit words
    int hp_words = 0;             // 2 / 3: a 128- / 192-bit precision sampler (sc_hip_cdt_hp_*): cdf holds the table's most significant limbs, hp_low the limbs below them (kept for sc_hip_cdt_hp_table only)
    std::vector<uint64_t> hp_low;
};
