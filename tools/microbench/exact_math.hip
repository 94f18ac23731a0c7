// Exhaustive proof (all 2^32 float bit patterns, on the GPU the kernels run on) that the short reciprocal / square-root
// sequences of ptk_kernels.hip return the very bits of the IEEE-754 operations the CPU oracle and the reference compute
// (`1.0f / a`, `sqrtf(x)`, round-to-nearest-even), so that swapping them in cannot change a single accumulator bit.
//   rcp:  y = v_rcp_f32(a) (1 ulp);  two Newton steps with exact residuals (fma)            [Markstein's scheme]
//   sqrt: y = v_rsq_f32(x); g = x*y, h = y/2; one coupled Newton step, then a residual correction of g
// Anything the short form gets wrong shows up here with its input; the kernel's helpers take the IEEE path there.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o exact_math exact_math.hip && ./exact_math
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float rcp_short(float a)
{
    float y = __builtin_amdgcn_rcpf(a);
    float e = __builtin_fmaf(-a, y, 1.0f);
    y = __builtin_fmaf(y, e, y);
    e = __builtin_fmaf(-a, y, 1.0f);
    y = __builtin_fmaf(y, e, y);
    return y;
}
__device__ __forceinline__ float rcp_short1(float a)     // one step only: is it enough?
{
    float y = __builtin_amdgcn_rcpf(a);
    float e = __builtin_fmaf(-a, y, 1.0f);
    return __builtin_fmaf(y, e, y);
}
__device__ __forceinline__ float sqrt_short(float x)
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    float r = __builtin_fmaf(-g, h, 0.5f);
    g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h);
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

// v_sqrt_f32 (1 ulp) and the two neighbour tests of the compiler's own IEEE expansion - exact residuals of the candidates
// one ulp below and above - WITHOUT that expansion's range scaling (x < 2^-96 is multiplied by 2^32 first) and special-case
// selects: which inputs need those?
__device__ __forceinline__ float sqrt_nb(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    s = rm <= 0.0f ? sm : s;
    return rp > 0.0f ? sp : s;
}

struct Report { unsigned long long bad[4]; unsigned first[4][8]; unsigned long long lo_bad[4], hi_bad[4]; };

__device__ void note(Report* r, int which, unsigned bits, bool in_domain)
{
    const unsigned long long k = atomicAdd(&r->bad[which], 1ull);
    if (k < 8) r->first[which][k] = bits;
    if (in_domain) { atomicAdd(&r->lo_bad[which], 1ull); atomicMax(&r->hi_bad[which], (unsigned long long)bits); }
}

__global__ void sweep(Report* rep)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride)
    {
        const unsigned bits = (unsigned)b;
        const float a = __uint_as_float(bits);
        const unsigned ex = (bits >> 23) & 255u;
        const bool is_nan = ex == 255u && (bits & 0x7fffffu);
        // reciprocal: domain = normal inputs whose reciprocal is normal too (2^-126 <= |a| <= 2^126)
        {
            const float ref = 1.0f / a;
            const bool dom = ex >= 1u && ex <= 252u;
            const float f2 = rcp_short(a), f1 = rcp_short1(a);
            const bool refnan = ref != ref;
            if (!(refnan ? (f2 != f2) : (__float_as_uint(f2) == __float_as_uint(ref)))) note(rep, 0, bits, dom && !is_nan);
            if (!(refnan ? (f1 != f1) : (__float_as_uint(f1) == __float_as_uint(ref)))) note(rep, 1, bits, dom && !is_nan);
        }
        {
            const float ref = sqrtf(a);
            const bool dom = ex >= 1u && ex <= 254u && !(bits >> 31);
            const float f = sqrt_short(a);
            const bool refnan = ref != ref;
            if (!(refnan ? (f != f) : (__float_as_uint(f) == __float_as_uint(ref)))) note(rep, 2, bits, dom);
            // neighbour-test form: domain = +0, positive normals, +inf (everything a squared length or a unit-interval draw can be)
            const bool dom_nb = bits == 0u || (!(bits >> 31) && ex >= 1u && !is_nan);
            const float g = sqrt_nb(a);
            if (!(refnan ? (g != g) : (__float_as_uint(g) == __float_as_uint(ref)))) note(rep, 3, bits, dom_nb);
        }
    }
}

int main()
{
    Report* d; Report h = {};
    CHECK(hipMalloc(&d, sizeof(Report)));
    CHECK(hipMemset(d, 0, sizeof(Report)));
    hipLaunchKernelGGL(sweep, dim3(256 * 16), dim3(256), 0, 0, d);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
    const char* names[4] = { "rcp, two Newton steps", "rcp, one Newton step", "sqrt, rsq + coupled step + residual",
                             "sqrt, v_sqrt_f32 + residual tests of both neighbours, no range scaling" };
    printf("{\"inputs\": 4294967296, \"rows\": [\n");
    for (int k = 0; k < 4; k++)
    {
        printf(" {\"sequence\": \"%s\", \"mismatches_all_inputs\": %llu, \"mismatches_in_domain\": %llu, \"largest_mismatching_input_in_domain_hex\": \"%08llx\", \"first_inputs_hex\": [", names[k], h.bad[k], h.lo_bad[k], h.hi_bad[k]);
        for (int j = 0; j < 8 && (unsigned long long)j < h.bad[k]; j++) printf("%s\"%08x\"", j ? ", " : "", h.first[k][j]);
        printf("]}%s\n", k < 3 ? "," : "");
    }
    printf("], \"domain\": \"rcp: normal a with 2^-126 <= |a| <= 2^126 (biased exponent 1..252); sqrt: positive normal x; neighbour-test sqrt: +0, positive normal x, +inf\"}\n");
    return 0;
}
