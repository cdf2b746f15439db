// experiment: how many shader cycles does one wave64 vector-ALU instruction occupy a SIMD for on gfx950?
// Settles the denominator of the scheme legs' VALU rooflines (VERDICT r02 "What's weak" 1b: 2 or 4 cycles).
//
// Method: every instruction is written in inline assembly (the compiler can neither fold nor re-order it), 8 independent
// destination registers per wave so that no instruction waits for the one before it; a workgroup is 256 threads = one wave
// on each SIMD of a CU, and W x 256 workgroups are launched so that every SIMD holds exactly W waves for the whole kernel.
// Cycles come from s_memtime (the shader clock itself, not wall time x an assumed frequency); s_memrealtime (100 MHz)
// gives the clock the chip actually held.  Reported per (instruction, W):
//    cyc/instr/SIMD = median wave's s_memtime span / (instructions the W waves of its SIMD issued in that span)
// The same binary run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE`
// calibrates what those counters read on a kernel whose issue rate is known (tools/prof_valu_peak.sh).
//
// build: hipcc -O2 --offload-arch=gfx950 -o tools/exp_valu_cycles tools/exp_valu_cycles.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define R8(a) a(0) a(1) a(2) a(3) a(4) a(5) a(6) a(7)
#define OUT32 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define OUT64 "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])

// instruction text generators: 8 independent destinations %0..%7, sources %8 (and %9)
#define I2(op, d) op " %" #d ", %" #d ", %8\n"
#define I3(op, d) op " %" #d ", %" #d ", %8, %9\n"
#define IK(op, d, k) op " %" #d ", %" #d ", %8, " k "\n"
#define IS(op, d, k) op " %" #d ", " k ", %" #d "\n"
#define A2(op) I2(op, 0) I2(op, 1) I2(op, 2) I2(op, 3) I2(op, 4) I2(op, 5) I2(op, 6) I2(op, 7)
#define A3(op) I3(op, 0) I3(op, 1) I3(op, 2) I3(op, 3) I3(op, 4) I3(op, 5) I3(op, 6) I3(op, 7)
#define AK(op, k) IK(op, 0, k) IK(op, 1, k) IK(op, 2, k) IK(op, 3, k) IK(op, 4, k) IK(op, 5, k) IK(op, 6, k) IK(op, 7, k)
#define AS(op, k) IS(op, 0, k) IS(op, 1, k) IS(op, 2, k) IS(op, 3, k) IS(op, 4, k) IS(op, 5, k) IS(op, 6, k) IS(op, 7, k)
#define X4(t) t t t t

#define OPS(F) \
    F(ADD_U32, "v_add_u32", 0, A2("v_add_u32")) F(SUB_U32, "v_sub_u32", 0, A2("v_sub_u32")) F(XOR_B32, "v_xor_b32", 0, A2("v_xor_b32")) F(AND_B32, "v_and_b32", 0, A2("v_and_b32")) \
    F(LSHLREV, "v_lshlrev_b32", 0, AS("v_lshlrev_b32", "3")) F(ASHRREV, "v_ashrrev_i32", 0, AS("v_ashrrev_i32", "3")) \
    F(MIN_U32, "v_min_u32", 0, A2("v_min_u32")) F(MAX_I32, "v_max_i32", 0, A2("v_max_i32")) \
    F(MUL_LO_U32, "v_mul_lo_u32", 0, A2("v_mul_lo_u32")) F(MUL_HI_U32, "v_mul_hi_u32", 0, A2("v_mul_hi_u32")) F(MUL_U32_U24, "v_mul_u32_u24", 0, A2("v_mul_u32_u24")) \
    F(MAD_I32_I24, "v_mad_i32_i24", 0, A3("v_mad_i32_i24")) F(MAD_U64_U32, "v_mad_u64_u32", 2, "") \
    F(BITOP3, "v_bitop3_b32", 0, AK("v_bitop3_b32", "%9 bitop3:0x96")) F(ALIGNBIT, "v_alignbit_b32", 0, AK("v_alignbit_b32", "13")) \
    F(AND_OR, "v_and_or_b32", 0, A3("v_and_or_b32")) F(ADD3, "v_add3_u32", 0, A3("v_add3_u32")) F(LSHL_ADD, "v_lshl_add_u32", 0, AK("v_lshl_add_u32", "3")) \
    F(XAD, "v_xad_u32", 0, A3("v_xad_u32")) F(PERM, "v_perm_b32", 0, A3("v_perm_b32")) F(CNDMASK, "v_cndmask_b32", 3, "") \
    F(ADD_F32, "v_add_f32", 0, A2("v_add_f32")) F(MUL_F32, "v_mul_f32", 0, A2("v_mul_f32")) F(FMA_F32, "v_fma_f32", 0, A3("v_fma_f32")) F(PK_FMA_F32, "v_pk_fma_f32", 1, A3("v_pk_fma_f32")) \
    F(ADD_F64, "v_add_f64", 1, A2("v_add_f64")) F(MUL_F64, "v_mul_f64", 1, A2("v_mul_f64")) F(FMA_F64, "v_fma_f64", 1, A3("v_fma_f64"))

#define ENUM_(id, name, kind, txt) id,
enum { OPS(ENUM_) NOPS };
#define NAME_(id, name, kind, txt) name,
static const char *names[NOPS] = { OPS(NAME_) };

template <int OP> __global__ __launch_bounds__(256) void k(uint64_t *stamps, uint32_t *sink, uint32_t seed, int iters)
{
    uint32_t a[8];
    double d[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + 1) + i * 977u; d[i] = 1.0 + 1e-9 * (double)(threadIdx.x + i); }
    uint32_t c = seed | 1u, e = seed * 77u + 5u;
    double fc = 1.0000001, fe = 1e-12;
    const uint64_t mask = 0x5555aaaa3333ccccull ^ seed;                 // v_cndmask_b32: the condition comes from an SGPR pair
    uint64_t t0, t1, r0, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int it = 0; it < iters; it++) {
        // 4 x 8 = 32 instructions per trip keep the loop overhead (s_add, s_cmp, s_cbranch: scalar, issued beside the vector stream) negligible
#define CASE_(id, name, kind, txt) \
        if (OP == id && kind == 0) asm volatile(X4(txt) : OUT32 : "v"(c), "v"(e)); \
        if (OP == id && kind == 1) asm volatile(X4(txt) : OUT64 : "v"(fc), "v"(fe));
        OPS(CASE_)
        if (OP == MAD_U64_U32) asm volatile(X4("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                                               "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n")
                                            : OUT64 : "v"(c), "v"(e) : "vcc");
        if (OP == CNDMASK) asm volatile(X4("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                                           "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n")
                                        : OUT32 : "v"(c), "s"(mask));
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i] ^ (uint32_t)__double_as_longlong(d[i]) ^ (uint32_t)(__double_as_longlong(d[i]) >> 32);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int OP> void run(uint64_t *d_st, uint32_t *d_sink, int only_w)
{
    for (int w = 1; w <= 8; w *= 2) {
        if (only_w && w != only_w) continue;
        const int blocks = 256 * w, iters = 4096, per_trip = 32;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, 12345u, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, 12345u, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> st(2 * blocks * 4);
        hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, mhz;
        for (int i = 0; i < blocks * 4; i++) { cyc.push_back((double)st[2 * i]); if (st[2 * i + 1]) mhz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 100.0); }
        std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
        const double med = cyc[cyc.size() / 2], clk = mhz.empty() ? 0 : mhz[mhz.size() / 2];
        const double instr_wave = (double)iters * per_trip;
        const double winstr = instr_wave * blocks * 4;
        printf("%-15s W=%d  wave span %9.0f cyc (min %9.0f max %9.0f)  %5.2f cyc/instr/SIMD by s_memtime   clock held %6.0f MHz   wall %7.3f ms -> %6.1f G wave-instr/s = %5.2f cyc/instr/SIMD at the held clock\n",
               names[OP], w, med, cyc.front(), cyc.back(), med / (instr_wave * w), clk, ms, winstr / ms / 1e6, 1024.0 * clk * 1e6 / (winstr / (ms * 1e-3)));
        fflush(stdout);
    }
}

int main(int argc, char **argv)
{
    const int only_w = argc > 1 ? atoi(argv[1]) : 0;
    uint64_t *d_st; uint32_t *d_sink;
    hipMalloc(&d_st, 2 * 8 * 256 * 4 * 8); hipMalloc(&d_sink, 8 * 256 * 256 * 4);
#define RUN_(id, name, kind, txt) run<id>(d_st, d_sink, only_w);
    OPS(RUN_)
    return 0;
}
