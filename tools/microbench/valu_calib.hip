// VALU issue-cost calibration for gfx950 (MI355X): what one wave-instruction of each class costs a SIMD, read two ways -
// wall time (HIP events) with 1 / 2 / 4 / 8 waves per SIMD issuing independent streams, and the SQ counters rocprofv3
// reads for the same launches (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES ...).  The table it
// prints is what DESIGN.md's "VALU pipe busy" is calibrated against (profiles/r02/valu_calibration.json).
//
//   hipcc -O3 --offload-arch=gfx950 -o valu_calib valu_calib.hip && ./valu_calib [waves_per_simd ...]
//
// Every kernel runs ITERS loop iterations of 32 instructions of one class on 8 independent register chains (so the
// instruction's own latency does not serialise one wave's stream); one-wave workgroups, grid = CUs x 4 SIMDs x W.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP4(X) X X X X
// one class = 8 chains x 4 repeats = 32 instructions per loop iteration
#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

#define KERNEL_F32(NAME, ASM)                                                                                         \
    __global__ __launch_bounds__(64, 8) void k_##NAME(float* out, int iters, unsigned long long* stamp)               \
    {                                                                                                                 \
        float a0 = 1.0f + threadIdx.x * 1e-3f, a1 = a0 + 0.1f, a2 = a0 + 0.2f, a3 = a0 + 0.3f, a4 = a0 + 0.4f,       \
              a5 = a0 + 0.5f, a6 = a0 + 0.6f, a7 = a0 + 0.7f;                                                         \
        const float b = 0.99999f + out[0], c = 1e-6f;                                                                 \
        unsigned long long t0 = 0, r0 = 0;                                                                            \
        if (stamp) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }                      \
        for (int i = 0; i < iters; i++)                                                                               \
        {                                                                                                             \
            asm volatile(REP4(ASM)                                                                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                         : "v"(b), "v"(c) : "vcc", "s20", "s21");                                                     \
        }                                                                                                             \
        if (stamp && blockIdx.x == 0 && threadIdx.x == 0)                                                             \
        { stamp[0] = __builtin_amdgcn_s_memtime() - t0; stamp[1] = __builtin_amdgcn_s_memrealtime() - r0; }           \
        out[1 + blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                               \
    }

#define KERNEL_PK(NAME, ASM)                                                                                          \
    __global__ __launch_bounds__(64, 8) void k_##NAME(float* out, int iters, unsigned long long* stamp)               \
    {                                                                                                                 \
        f2 a0 = { 1.0f + threadIdx.x * 1e-3f, 1.5f }, a1 = a0 + 0.1f, a2 = a0 + 0.2f, a3 = a0 + 0.3f, a4 = a0 + 0.4f, \
           a5 = a0 + 0.5f, a6 = a0 + 0.6f, a7 = a0 + 0.7f;                                                            \
        const f2 b = { 0.99999f + out[0], 0.99998f }, c = { 1e-6f, 2e-6f };                                           \
        unsigned long long t0 = 0, r0 = 0;                                                                            \
        if (stamp) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }                      \
        for (int i = 0; i < iters; i++)                                                                               \
        {                                                                                                             \
            asm volatile(REP4(ASM)                                                                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                         : "v"(b), "v"(c) : "vcc");                                                                   \
        }                                                                                                             \
        if (stamp && blockIdx.x == 0 && threadIdx.x == 0)                                                             \
        { stamp[0] = __builtin_amdgcn_s_memtime() - t0; stamp[1] = __builtin_amdgcn_s_memrealtime() - r0; }           \
        f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                 \
        out[1 + blockIdx.x * 64 + threadIdx.x] = s.x + s.y;                                                           \
    }

#define KERNEL_F64(NAME, ASM)                                                                                         \
    __global__ __launch_bounds__(64, 8) void k_##NAME(float* out, int iters, unsigned long long* stamp)               \
    {                                                                                                                 \
        double a0 = 1.0 + threadIdx.x * 1e-3, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3, a4 = a0 + 0.4,             \
               a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7;                                                           \
        const double b = 0.99999 + out[0], c = 1e-6;                                                                  \
        unsigned long long t0 = 0, r0 = 0;                                                                            \
        if (stamp) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }                      \
        for (int i = 0; i < iters; i++)                                                                               \
        {                                                                                                             \
            asm volatile(REP4(ASM)                                                                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                         : "v"(b), "v"(c) : "vcc");                                                                   \
        }                                                                                                             \
        if (stamp && blockIdx.x == 0 && threadIdx.x == 0)                                                             \
        { stamp[0] = __builtin_amdgcn_s_memtime() - t0; stamp[1] = __builtin_amdgcn_s_memrealtime() - r0; }           \
        out[1 + blockIdx.x * 64 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                      \
    }

#define S8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
#define A_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define A_MUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define A_ADD(k) "v_add_f32 %" #k ", %" #k ", %9\n"
#define A_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
#define A_MINMAX(k) "v_min_f32 %" #k ", %" #k ", %8\n"
#define A_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define A_SQRT(k) "v_sqrt_f32 %" #k ", %" #k "\n"
#define A_RSQ(k) "v_rsq_f32 %" #k ", %" #k "\n"
#define A_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define A_MULHI(k) "v_mul_hi_u32 %" #k ", %" #k ", %8\n"
#define A_MUL24(k) "v_mul_u32_u24 %" #k ", %" #k ", %8\n"
#define A_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define A_LSHR(k) "v_lshrrev_b32 %" #k ", 3, %" #k "\n"
#define A_ADDU(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define A_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define A_CMP(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n"
#define A_CVTUB(k) "v_cvt_f32_ubyte0 %" #k ", %" #k "\n"
#define A_CVTI(k) "v_cvt_i32_f32 %" #k ", %" #k "\n"
#define A_DIVSCALE(k) "v_div_scale_f32 %" #k ", vcc, %" #k ", %8, %" #k "\n"
#define A_DIVFMAS(k) "v_div_fmas_f32 %" #k ", %" #k ", %8, %9\n"
#define A_DIVFIXUP(k) "v_div_fixup_f32 %" #k ", %" #k ", %8, %9\n"
#define A_CNDMASK_S(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[20:21]\n"
#define A_CMPCND(k) "v_cmp_lt_f32 vcc, %" #k ", %8\nv_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
#define A_MAX(k) "v_max_f32 %" #k ", %" #k ", %8\n"
#define A_MIN3(k) "v_min3_f32 %" #k ", %" #k ", %8, %9\n"
#define A_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n"
#define A_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
#define A_MINU(k) "v_min_u32 %" #k ", %" #k ", %8\n"
#define A_SUB(k) "v_sub_f32 %" #k ", %" #k ", %8\n"
#define A_MAD24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n"
#define A_CVTF32U32(k) "v_cvt_f32_u32 %" #k ", %" #k "\n"
#define A_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %8\n"
#define A_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 8, 8\n"
#define A_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define A_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define A_PKMUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n"
#define A_PKADD(k) "v_pk_add_f32 %" #k ", %" #k ", %9\n"
#define A_PKFMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define A_FMAMIX(k) "v_fma_mix_f32 %" #k ", %" #k ", %8, %9 op_sel_hi:[1,0,0]\n"
#define A_FMAMIXHI(k) "v_fma_mix_f32 %" #k ", %" #k ", %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define A_BFI(k) "v_bfi_b32 %" #k ", %8, %" #k ", %9\n"
#define A_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n"
#define A_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 8, %8\n"
#define A_CVTF16(k) "v_cvt_f32_f16 %" #k ", %" #k "\n"
#define A_CMPE64(k) "v_cmp_lt_f32_e64 s[20:21], %" #k ", %8\n"
#define A_FMA64(k) "v_fma_f64 %" #k ", %" #k ", %8, %9\n"
#define A_MUL64(k) "v_mul_f64 %" #k ", %" #k ", %8\n"
#define A_ADD64(k) "v_add_f64 %" #k ", %" #k ", %9\n"

KERNEL_F32(fma, S8(A_FMA))
KERNEL_F32(mul, S8(A_MUL))
KERNEL_F32(add, S8(A_ADD))
KERNEL_F32(max3, S8(A_MAX3))
KERNEL_F32(min, S8(A_MINMAX))
KERNEL_F32(rcp, S8(A_RCP))
KERNEL_F32(sqrt, S8(A_SQRT))
KERNEL_F32(rsq, S8(A_RSQ))
KERNEL_F32(mul_lo_u32, S8(A_MULLO))
KERNEL_F32(mul_hi_u32, S8(A_MULHI))
KERNEL_F32(mul_u32_u24, S8(A_MUL24))
KERNEL_F32(xor, S8(A_XOR))
KERNEL_F32(lshr, S8(A_LSHR))
KERNEL_F32(add_u32, S8(A_ADDU))
KERNEL_F32(cndmask, S8(A_CNDMASK))
KERNEL_F32(cmp, S8(A_CMP))
KERNEL_F32(cvt_f32_ubyte, S8(A_CVTUB))
KERNEL_F32(cvt_i32_f32, S8(A_CVTI))
KERNEL_F32(div_scale, S8(A_DIVSCALE))
KERNEL_F32(div_fmas, S8(A_DIVFMAS))
KERNEL_F32(div_fixup, S8(A_DIVFIXUP))
KERNEL_F32(bfe, S8(A_BFE))
KERNEL_F32(cndmask_sgpr, S8(A_CNDMASK_S))
KERNEL_F32(cmp_cndmask, S8(A_CMPCND))
KERNEL_F32(max, S8(A_MAX))
KERNEL_F32(min3, S8(A_MIN3))
KERNEL_F32(and, S8(A_AND))
KERNEL_F32(and_or, S8(A_ANDOR))
KERNEL_F32(min_u32, S8(A_MINU))
KERNEL_F32(sub, S8(A_SUB))
KERNEL_F32(mad_u32_u24, S8(A_MAD24))
KERNEL_F32(cvt_f32_u32, S8(A_CVTF32U32))
KERNEL_F32(lshl_add_u32, S8(A_LSHLADD))
KERNEL_F32(perm, S8(A_PERM))
KERNEL_F32(mov, S8(A_MOV))
KERNEL_F32(fma_mix, S8(A_FMAMIX))
KERNEL_F32(fma_mix_hi, S8(A_FMAMIXHI))
KERNEL_F32(bfi, S8(A_BFI))
KERNEL_F32(med3, S8(A_MED3))
KERNEL_F32(lshl_or, S8(A_LSHLOR))
KERNEL_F32(cvt_f32_f16, S8(A_CVTF16))
KERNEL_F32(cmp_e64, S8(A_CMPE64))
KERNEL_PK(pk_mul, S8(A_PKMUL))
KERNEL_PK(pk_add, S8(A_PKADD))
KERNEL_PK(pk_fma, S8(A_PKFMA))
KERNEL_F64(fma_f64, S8(A_FMA64))
KERNEL_F64(mul_f64, S8(A_MUL64))
KERNEL_F64(add_f64, S8(A_ADD64))

// compiler-expanded sequences, as the render kernel uses them (-ffp-contract=off): 8 independent chains x 4
__global__ __launch_bounds__(64, 8) void k_ieee_div(float* out, int iters, unsigned long long* stamp)
{
    unsigned long long t0 = 0, r0 = 0;
    if (stamp) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float a[8];
    for (int k = 0; k < 8; k++) a[k] = 1.0f + threadIdx.x * 1e-3f + 0.1f * k;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = 1.0f / a[k];
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) { stamp[0] = __builtin_amdgcn_s_memtime() - t0; stamp[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0; for (int k = 0; k < 8; k++) s += a[k];
    out[1 + blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ __launch_bounds__(64, 8) void k_ieee_sqrt(float* out, int iters, unsigned long long* stamp)
{
    unsigned long long t0 = 0, r0 = 0;
    if (stamp) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float a[8];
    for (int k = 0; k < 8; k++) a[k] = 1.0f + threadIdx.x * 1e-3f + 0.1f * k;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = sqrtf(a[k]) + 1.0f;
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) { stamp[0] = __builtin_amdgcn_s_memtime() - t0; stamp[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0; for (int k = 0; k < 8; k++) s += a[k];
    out[1 + blockIdx.x * 64 + threadIdx.x] = s;
}
// one PCG draw as the render kernel makes it (state update + output permutation + conversion)
__global__ __launch_bounds__(64, 8) void k_pcg_draw(float* out, int iters, unsigned long long* stamp)
{
    unsigned long long t0 = 0, r0 = 0;
    if (stamp) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    unsigned st[8]; float acc[8];
    for (int k = 0; k < 8; k++) { st[k] = threadIdx.x * 977u + k; acc[k] = 0.0f; }
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                unsigned old = st[k];
                st[k] = old * 747796405u + 2891336453u;
                unsigned w = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
                w = (w >> 22u) ^ w;
                acc[k] += (float)(w >> 8) * 5.9604644775390625e-8f;
            }
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) { stamp[0] = __builtin_amdgcn_s_memtime() - t0; stamp[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0; for (int k = 0; k < 8; k++) s += acc[k];
    out[1 + blockIdx.x * 64 + threadIdx.x] = s;
}
// ballot + popcount as the wave state machine does (v_cmp -> SGPR pair, s_bcnt1)
__global__ __launch_bounds__(64, 8) void k_ballot(float* out, int iters, unsigned long long* stamp)
{
    unsigned long long t0 = 0, r0 = 0;
    if (stamp) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    int st = threadIdx.x & 3; int n = 0;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int r = 0; r < 32; r++)
        {
            n += __popcll(__ballot(st == (r & 3)));
            st = (st + (n & 1)) & 3;
        }
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) { stamp[0] = __builtin_amdgcn_s_memtime() - t0; stamp[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    out[1 + blockIdx.x * 64 + threadIdx.x] = (float)n;
}
// LDS stack traffic: ds_write_b32 + ds_read_b32 at [level][lane]
__global__ __launch_bounds__(64, 8) void k_lds_stack(float* out, int iters, unsigned long long* stamp)
{
    unsigned long long t0 = 0, r0 = 0;
    if (stamp) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    __shared__ int stack[16 * 64];
    int* s = stack + threadIdx.x; int v = threadIdx.x, sp = 0;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            s[sp * 64] = v; sp = (sp + 1) & 15;
            v += s[((sp + 7) & 15) * 64];
        }
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) { stamp[0] = __builtin_amdgcn_s_memtime() - t0; stamp[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    out[1 + blockIdx.x * 64 + threadIdx.x] = (float)v;
}

// The node arm's byte -> float conversion through f16 DENORMALS: a half whose bits are 0x00NN is NN * 2^-24 exactly, and
// v_fma_mix_f32 converts it on the fly, so fma(NN * 2^-24, A * 2^24, B) must equal fma((float)NN, A, B) bit for bit
// (power-of-two scalings are exact).  Checked for every byte against a spread of A and B.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k_mix_check(unsigned* mismatches)
{
    const unsigned q = threadIdx.x & 255u;
    const unsigned packed = q | (q << 16);
    const h2 h = __builtin_bit_cast(h2, packed);
    unsigned bad = 0;
    for (int i = 0; i < 4096; i++)
    {
        const float A = __uint_as_float(0x2f800000u + (unsigned)(blockIdx.x * 4096 + i) * 2654435761u % 0x20000000u) * ((i & 1) ? -1.0f : 1.0f);
        const float B = __uint_as_float(0x30000000u + (unsigned)(blockIdx.x * 4096 + i) * 40503u % 0x1f000000u) * ((i & 2) ? -1.0f : 1.0f);
        const float ref = __builtin_fmaf((float)q, A, B);
        const float lo = __builtin_fmaf((float)h.x, A * 16777216.0f, B), hi = __builtin_fmaf((float)h.y, A * 16777216.0f, B);
        bad += (__float_as_uint(ref) != __float_as_uint(lo)) + (__float_as_uint(ref) != __float_as_uint(hi));
    }
    if (bad) atomicAdd(mismatches, bad);
}

struct Entry { const char* name; void (*fn)(float*, int, unsigned long long*); int per_iter; const char* note; };
#define E(n, per, note) { #n, k_##n, per, note }
static const Entry entries[] = {
    E(fma, 32, "v_fma_f32"), E(mul, 32, "v_mul_f32"), E(add, 32, "v_add_f32"), E(max3, 32, "v_max3_f32"), E(min, 32, "v_min_f32"),
    E(mov, 32, "v_mov_b32"), E(cndmask, 32, "v_cndmask_b32 (vcc)"), E(cmp, 32, "v_cmp_lt_f32 -> vcc"),
    E(xor, 32, "v_xor_b32"), E(lshr, 32, "v_lshrrev_b32"), E(add_u32, 32, "v_add_u32"), E(bfe, 32, "v_bfe_u32"), E(cndmask_sgpr, 32, "v_cndmask_b32_e64 (mask in an SGPR pair)"), E(cmp_cndmask, 32, "v_cmp_lt_f32 + v_cndmask_b32 pair (per pair)"),
    E(max, 32, "v_max_f32"), E(min3, 32, "v_min3_f32"), E(and, 32, "v_and_b32"), E(and_or, 32, "v_and_or_b32"), E(min_u32, 32, "v_min_u32"), E(sub, 32, "v_sub_f32"),
    E(mad_u32_u24, 32, "v_mad_u32_u24"), E(cvt_f32_u32, 32, "v_cvt_f32_u32"), E(lshl_add_u32, 32, "v_lshl_add_u32"), E(perm, 32, "v_perm_b32"),
    E(mul_u32_u24, 32, "v_mul_u32_u24"), E(mul_lo_u32, 32, "v_mul_lo_u32"), E(mul_hi_u32, 32, "v_mul_hi_u32"),
    E(cvt_f32_ubyte, 32, "v_cvt_f32_ubyte0"), E(cvt_i32_f32, 32, "v_cvt_i32_f32"),
    E(rcp, 32, "v_rcp_f32"), E(sqrt, 32, "v_sqrt_f32"), E(rsq, 32, "v_rsq_f32"),
    E(div_scale, 32, "v_div_scale_f32"), E(div_fmas, 32, "v_div_fmas_f32"), E(div_fixup, 32, "v_div_fixup_f32"),
    E(fma_mix, 32, "v_fma_mix_f32, src0 = low f16 half"), E(fma_mix_hi, 32, "v_fma_mix_f32, src0 = high f16 half"), E(bfi, 32, "v_bfi_b32"), E(med3, 32, "v_med3_f32"),
    E(lshl_or, 32, "v_lshl_or_b32"), E(cvt_f32_f16, 32, "v_cvt_f32_f16"), E(cmp_e64, 32, "v_cmp_lt_f32_e64 -> SGPR pair"),
    E(pk_mul, 32, "v_pk_mul_f32 (2 results per lane)"), E(pk_add, 32, "v_pk_add_f32"), E(pk_fma, 32, "v_pk_fma_f32"),
    E(fma_f64, 32, "v_fma_f64"), E(mul_f64, 32, "v_mul_f64"), E(add_f64, 32, "v_add_f64"),
    E(ieee_div, 32, "1.0f / x as hipcc expands it (per division)"), E(ieee_sqrt, 32, "sqrtf(x) + 1 as hipcc expands it (per sqrt)"),
    E(pcg_draw, 32, "one PCG-RXS-M-XS draw + u01 conversion (per draw)"),
    E(ballot, 32, "v_cmp + s_bcnt1 + dependent update (per ballot)"), E(lds_stack, 16, "ds_write_b32 + ds_read_b32 pair, [level][lane] (per pair)"),
};

int main(int argc, char** argv)
{
    std::vector<int> waves;
    const char* only = nullptr;
    for (int i = 1; i < argc; i++) { if (!strncmp(argv[i], "--only=", 7)) only = argv[i] + 7; else waves.push_back(atoi(argv[i])); }
    if (waves.empty()) waves = { 1, 2, 4, 8 };
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; unsigned long long* stamp;
    CHECK(hipMalloc(&out, sizeof(float) * (1 + (size_t)cus * 4 * 8 * 64)));
    CHECK(hipMemset(out, 0, sizeof(float) * (1 + (size_t)cus * 4 * 8 * 64)));
    CHECK(hipMalloc(&stamp, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    // in-kernel clock under load (s_memtime / s_memrealtime), from the fma kernel at 4 waves per SIMD
    double ghz = 0;
    {
        hipLaunchKernelGGL(k_fma, dim3(cus * 16), dim3(64), 0, 0, out, iters * 4, stamp);
        CHECK(hipDeviceSynchronize());
        unsigned long long h[2]; CHECK(hipMemcpy(h, stamp, 16, hipMemcpyDeviceToHost));
        ghz = (double)h[0] / (double)h[1] * 0.1;
    }
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz_in_kernel\": %.3f, \"iters\": %d, \"rows\": [\n", prop.name, cus, ghz, iters);
    bool first = true;
    for (const Entry& en : entries)
    {
        if (only && strcmp(only, en.name)) continue;
        printf("%s {\"op\": \"%s\", \"what\": \"%s\"", first ? "" : ",\n", en.name, en.note); first = false;
        for (int w : waves)
        {
            const int blocks = cus * 4 * w;
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(64), 0, 0, out, 100, (unsigned long long*)nullptr);   // warm
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(64), 0, 0, out, iters, stamp);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2]; CHECK(hipMemcpy(h, stamp, 16, hipMemcpyDeviceToHost));
            const double kghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : ghz;     // this kernel's own clock (block 0)
            // every SIMD ran w waves, each iters x per_iter units: time / cycles a SIMD spent per unit
            const double ns = (double)ms * 1e6 / ((double)iters * en.per_iter * w);
            printf(", \"w%d\": {\"ns\": %.3f, \"ghz\": %.3f, \"cyc\": %.2f}", w, ns, kghz, ns * kghz);
        }
        printf("}");
        fflush(stdout);
    }
    unsigned* d_bad; unsigned h_bad = 0;
    CHECK(hipMalloc(&d_bad, 4)); CHECK(hipMemset(d_bad, 0, 4));
    hipLaunchKernelGGL(k_mix_check, dim3(1024), dim3(256), 0, 0, d_bad);
    CHECK(hipMemcpy(&h_bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("\n],\n \"fma_mix_f16_denormal_check\": {\"what\": \"fma(byte as f16 denormal, A * 2^24, B) vs fma((float)byte, A, B), 256 bytes x 2 halves x 4 M (A, B) pairs\", \"mismatches\": %u}}\n", h_bad);
    return 0;
}
