// hip_host_emu.h - the HOST side of <hip/hip_runtime.h> for the lane-by-lane emulation: enough of the HIP API for the product's own host code
// (libsafecrypto_amd/csrc/*.hip, *.cpp) to be compiled for the CPU and run - every allocation host memory (zeroed, as the memory the library sees on the
// device is - fresh from the driver, or its own after wipe_free; EMU_MALLOC_FILL=<byte> in the environment fills it with that byte instead, to show
// that nothing but the residue hooks relies on the zeroes), every copy a memcpy, every stream the one in-order stream, every launch run to completion before hipLaunchKernelGGL returns
// (emu::launch: a host thread per lane).  One "device" of EMU_CUS compute units.  Stream capture is refused (hipErrorNotSupported).
// TEST INFRASTRUCTURE ONLY (tests/test_emu_library.py builds tests/emu/_build/libsc_hip_emu_*.so from it); the product never sees this file.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <vector>
#include <tuple>
#include <utility>
#include <type_traits>

#ifndef EMU_CUS
#define EMU_CUS 2
#endif

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotSupported = 801, hipErrorInvalidDevice = 101 };
typedef struct EmuStream *hipStream_t;
typedef struct EmuEvent *hipEvent_t;
typedef struct EmuGraph *hipGraph_t;
typedef struct EmuGraphExec *hipGraphExec_t;
typedef void *hipDeviceptr_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostRegisterDefault = 0, hipHostRegisterPortable = 1, hipStreamCaptureModeThreadLocal = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipDeviceAttribute_t { hipDeviceAttributeMaxSharedMemoryPerBlock = 74, hipDeviceAttributeMultiprocessorCount = 63 };
enum hipMemoryType { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeUnregistered = 0 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void *devicePointer, *hostPointer; int isManaged; unsigned allocationFlags; };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; size_t sharedMemPerBlock; int warpSize; char gcnArchName[256]; };

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorNotSupported ? "operation not supported (emulation)" : "error (emulation)"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize();
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof *p); strcpy(p->name, "MI355X (CPU emulation)"); strcpy(p->gcnArchName, "gfx950"); p->multiProcessorCount = EMU_CUS; p->totalGlobalMem = (size_t)1 << 34; p->sharedMemPerBlock = 160 * 1024; p->warpSize = 64; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeMaxSharedMemoryPerBlock ? 160 * 1024 : a == hipDeviceAttributeMultiprocessorCount ? EMU_CUS : 0; return hipSuccess; }
// ---- fault injection (tests/test_emu_faults.py): the (N+1)-th call from now of a class fails, once; live allocations are counted so that a failed call can be held to
// "nothing of it stays allocated".  Classes: 0 allocations (hipMalloc / hipHostMalloc), 1 copies and memsets, 2 stream / event creation, 3 hipHostRegister.
namespace emu {
inline std::atomic<long> fault_in[4] = { {-1}, {-1}, {-1}, {-1} };      // calls of the class still to succeed before one fails; < 0: none will
inline std::atomic<long> calls[4] = { {0}, {0}, {0}, {0} };             // calls of the class seen so far
inline std::atomic<long> live_allocs{0};
static inline bool trip(int cls)
{
    calls[cls].fetch_add(1, std::memory_order_relaxed);
    long v = fault_in[cls].load(std::memory_order_relaxed);
    while (v >= 0) if (fault_in[cls].compare_exchange_weak(v, v - 1, std::memory_order_relaxed)) return v == 0;
    return false;
}
}  // namespace emu
extern "C" __attribute__((weak, used, visibility("default"))) void sc_emu_fault(int cls, long countdown) { if (cls >= 0 && cls < 4) emu::fault_in[cls].store(countdown); }
extern "C" __attribute__((weak, used, visibility("default"))) long sc_emu_calls(int cls) { return cls >= 0 && cls < 4 ? emu::calls[cls].load() : -1; }
extern "C" __attribute__((weak, used, visibility("default"))) long sc_emu_live_allocs(void) { return emu::live_allocs.load(); }

template <typename T> static inline hipError_t hipMalloc(T **p, size_t n)
{
    void *q = nullptr;
    if (emu::trip(0) || posix_memalign(&q, 256, n ? n : 1)) { *p = nullptr; return hipErrorOutOfMemory; }
    static const int fill = getenv("EMU_MALLOC_FILL") ? (int)strtol(getenv("EMU_MALLOC_FILL"), nullptr, 0) : 0;
    memset(q, fill, n);
    *p = (T *)q;
    emu::live_allocs.fetch_add(1, std::memory_order_relaxed);
    return hipSuccess;
}
static inline hipError_t hipFree(void *p);
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void *p);
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return emu::trip(3) ? hipErrorInvalidValue : hipSuccess; }
static inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { memset(a, 0, sizeof *a); return hipErrorInvalidValue; }      // "not a registered pointer"
static inline hipError_t hipMemPtrGetInfo(void *, size_t *) { return hipErrorInvalidValue; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind);
// ---- streams and events ---------------------------------------------------------------------------------------------------------------------------------
// Default: every stream is the one in-order stream - work runs where it is issued.
// EMU_SIDE_STREAMS (tests/test_emu_stream_order.py): two adversaries for the library's fork / join logic.
//   late      - work issued to the NULL stream (the caller's: the tests read results right behind a call, as a caller that synchronises its stream would) runs at once;
//               work issued to any OTHER stream - the streams the library creates for itself - is QUEUED and runs only when something the HIP model says must wait for it
//               asks: hipStreamWaitEvent on an event recorded behind it (when the waiting stream's own work reaches that point), hipEventSynchronize,
//               hipStreamSynchronize, hipDeviceSynchronize, hipFree (a device-wide synchronisation), hipStreamDestroy.  A consumer on the caller's stream that was not made
//               to wait for its producer on a side stream reads what was there BEFORE (a missing JOIN) - and so does a synchronous hipMemcpy behind unjoined side work
//               (a non-blocking stream does not synchronise with the null stream).
//   null-late - the other way round: the caller's stream is the one held back (until a synchronisation of it or of the device, a synchronous copy, hipFree, or an event
//               wait issued to a side stream that names a point on it) and the library's own streams run at once - side work that was not made to wait for the caller's
//               earlier work reads its inputs before they are written (a missing FORK).  The caller has to synchronise before it reads (sc_emu_sync).
struct EmuStream { std::deque<std::function<void()>> q; unsigned long long enq = 0, done = 0; bool blocking = true; };
struct EmuEvent { EmuStream *s = nullptr; unsigned long long seq = 0; };
namespace emu {
inline const int stream_mode = !getenv("EMU_SIDE_STREAMS") ? 0 : !strcmp(getenv("EMU_SIDE_STREAMS"), "late") ? 1 : !strcmp(getenv("EMU_SIDE_STREAMS"), "null-late") ? 2 : 0;
inline const bool side_late = stream_mode != 0;                          // (either adversary: streams are real queues)
inline std::recursive_mutex order_mu;                                   // (the multi-device driver issues from several host threads)
inline EmuStream null_stream;
inline std::vector<EmuStream *> all_streams;
inline std::atomic<long> deferred_total{0};                              // items that ran later than they were issued (a run that reports 0 has exercised nothing)
static inline void run_upto(EmuStream *s, unsigned long long seq)
{
    std::lock_guard<std::recursive_mutex> g(order_mu);
    while (s && s->done < seq && !s->q.empty()) { std::function<void()> f = std::move(s->q.front()); s->q.pop_front(); s->done++; f(); }
}
static inline void flush_all()
{
    std::lock_guard<std::recursive_mutex> g(order_mu);
    run_upto(&null_stream, null_stream.enq);
    for (size_t i = 0; i < all_streams.size(); i++) run_upto(all_streams[i], all_streams[i]->enq);
    run_upto(&null_stream, null_stream.enq);
}
static inline bool held(hipStream_t st) { return stream_mode == 1 ? st != nullptr : stream_mode == 2 ? st == nullptr : false; }
// issue `f` to stream `st`: at once, or behind what the stream already holds
template <typename F> static inline void issue(hipStream_t st, F &&f)
{
    if (!held(st)) { f(); return; }
    std::lock_guard<std::recursive_mutex> g(order_mu);
    EmuStream *s = st ? st : &null_stream;
    s->q.emplace_back(std::forward<F>(f)); s->enq++;
    deferred_total.fetch_add(1, std::memory_order_relaxed);
}
}  // namespace emu
extern "C" __attribute__((weak, used, visibility("default"))) long sc_emu_deferred(void) { return emu::deferred_total.load(); }
extern "C" __attribute__((weak, used, visibility("default"))) void sc_emu_sync(void) { if (emu::side_late) emu::flush_all(); }     // what a caller's hipDeviceSynchronize would do

static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind kind, hipStream_t st = nullptr)
{
    if (emu::trip(1)) return hipErrorInvalidValue;
    if (!n) return hipSuccess;
    if (emu::held(st) && kind == hipMemcpyHostToDevice) {         // a pageable source is staged before the call returns: what the copy delivers is what the host held NOW
        auto buf = std::make_shared<std::vector<unsigned char>>((const unsigned char *)s, (const unsigned char *)s + n);
        emu::issue(st, [d, buf]() { memcpy(d, buf->data(), buf->size()); });
    } else emu::issue(st, [d, s, n]() { memmove(d, s, n); });
    return hipSuccess;
}
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (emu::trip(1)) return hipErrorInvalidValue; if (emu::stream_mode == 2) emu::run_upto(&emu::null_stream, emu::null_stream.enq); if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (emu::trip(1)) return hipErrorInvalidValue; if (emu::stream_mode == 2) emu::run_upto(&emu::null_stream, emu::null_stream.enq); if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st = nullptr) { if (emu::trip(1)) return hipErrorInvalidValue; if (n) emu::issue(st, [d, v, n]() { memset(d, v, n); }); return hipSuccess; }
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t st = nullptr) { emu::issue(st, [d, v, count]() { for (size_t i = 0; i < count; i++) ((int *)d)[i] = v; }); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags)
{
    if (emu::trip(2)) { *s = nullptr; return hipErrorOutOfMemory; }
    EmuStream *st = new EmuStream; st->blocking = !(flags & hipStreamNonBlocking);
    { std::lock_guard<std::recursive_mutex> g(emu::order_mu); emu::all_streams.push_back(st); }
    *s = st; emu::live_allocs.fetch_add(1);
    return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t s) { EmuStream *q = s ? s : &emu::null_stream; emu::run_upto(q, q->enq); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s)
{
    if (!s) return hipSuccess;
    emu::run_upto(s, s->enq);                                            // (the runtime lets a destroyed stream drain)
    { std::lock_guard<std::recursive_mutex> g(emu::order_mu); for (size_t i = 0; i < emu::all_streams.size(); i++) if (emu::all_streams[i] == s) { emu::all_streams.erase(emu::all_streams.begin() + (long)i); break; } }
    emu::live_allocs.fetch_sub(1); delete s;
    return hipSuccess;
}
static inline hipError_t hipEventCreate(hipEvent_t *e) { if (emu::trip(2)) { *e = nullptr; return hipErrorOutOfMemory; } *e = new EmuEvent; emu::live_allocs.fetch_add(1); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { if (e) { emu::live_allocs.fetch_sub(1); delete e; } return hipSuccess; }
// an event stands for "everything issued to that stream so far"; a wait captures that point when it is issued (a later re-record does not move it)
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr) { if (e) { std::lock_guard<std::recursive_mutex> g(emu::order_mu); e->s = s ? s : &emu::null_stream; e->seq = e->s->enq; } return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    if (!e || !emu::side_late) return hipSuccess;
    EmuStream *es; unsigned long long seq;
    { std::lock_guard<std::recursive_mutex> g(emu::order_mu); es = e->s; seq = e->seq; }
    if (es) emu::issue(s, [es, seq]() { emu::run_upto(es, seq); });
    return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t e) { if (e && e->s) emu::run_upto(e->s, e->seq); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { hipEventSynchronize(a); hipEventSynchronize(b); *ms = 1.0f; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { if (emu::side_late) emu::flush_all(); return hipSuccess; }
static inline hipError_t hipFree(void *p) { if (emu::side_late) emu::flush_all(); if (p) emu::live_allocs.fetch_sub(1, std::memory_order_relaxed); free(p); return hipSuccess; }   // (hipFree synchronises the device)
static inline hipError_t hipHostFree(void *p) { return hipFree(p); }
template <typename F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
template <typename F> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *nb, F, int, size_t) { *nb = 2; return hipSuccess; }
static inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorNotSupported; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = nullptr; return hipErrorNotSupported; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t, void *, void *, size_t) { *e = nullptr; return hipErrorNotSupported; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#include <sanitizer/asan_interface.h>
#define EMU_POISON(p, n) __asan_poison_memory_region((p), (n))
#define EMU_UNPOISON(p, n) __asan_unpoison_memory_region((p), (n))
#endif
#endif
#ifndef EMU_POISON
#define EMU_POISON(p, n) ((void)0)
#define EMU_UNPOISON(p, n) ((void)0)
#endif

namespace emu {
// a launch as the host code makes it: one at a time (the product's multi-device driver launches from several host threads), the dynamic LDS beyond what the
// launch asked for poisoned under AddressSanitizer, the launch complete on return
inline std::mutex launch_mu;
template <typename F> static inline void launch_host(dim3 grid, dim3 block, size_t lds_bytes, F &&body)
{
    if (block.y != 1 || block.z != 1 || !block.x || lds_bytes > sizeof dyn_lds) { fprintf(stderr, "emu: launch shape (%u,%u,%u) x (%u,%u,%u), %zu bytes of LDS not emulated\n", grid.x, grid.y, grid.z, block.x, block.y, block.z, lds_bytes); abort(); }
    std::lock_guard<std::mutex> g(launch_mu);
    if (!grid.x || !grid.y || !grid.z) return;
    const size_t keep = (lds_bytes + 7) & ~(size_t)7;
    if (keep < sizeof dyn_lds) EMU_POISON(dyn_lds + keep, sizeof dyn_lds - keep);
    const std::function<void()> fn(body);
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++) { gridDim.y = grid.y; gridDim.z = grid.z; blockIdx.y = y; blockIdx.z = z; launch(grid.x, block.x, fn); }
    gridDim.y = gridDim.z = 1; blockIdx.y = blockIdx.z = 0;
    EMU_UNPOISON(dyn_lds, sizeof dyn_lds);
}
// the argument expressions of a launch are evaluated ONCE, on the launching thread, and kept by value - as a real launch marshals them; every lane then calls the
// kernel with the same values (a lambda around `kernel(expressions...)` would re-evaluate the expressions in every lane thread, concurrently).  The kernel is still
// called BY NAME inside the macro so that its default arguments and template-ids work as they do in a <<<>>> launch.
template <typename... A> static inline std::tuple<std::decay_t<A>...> launch_args(A &&...a) { return std::tuple<std::decay_t<A>...>(std::forward<A>(a)...); }
}  // namespace emu
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) do { const auto emu_launch_args = std::make_shared<decltype(emu::launch_args(__VA_ARGS__))>(emu::launch_args(__VA_ARGS__)); \
    const dim3 emu_g = dim3(grid), emu_b = dim3(block); const size_t emu_l = (size_t)(lds); \
    emu::issue((hipStream_t)(stream), [=]() { emu::launch_host(emu_g, emu_b, emu_l, [&]() { std::apply([&](const auto &...emu_a) { kernel(emu_a...); }, *emu_launch_args); }); }); } while (0)
