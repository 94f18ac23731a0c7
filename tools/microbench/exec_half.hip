// Does a wave64 VALU instruction cost less when one 32-lane half of the wave is switched off?  (If it did, keeping a
// wave's live lanes in one half would pay.)  Streams of v_fma_f32 / v_cndmask under different exec masks, 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o exec_half exec_half.hip && ./exec_half
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(64, 8) void k(float* out, int iters)
{
    const int lane = threadIdx.x;
    const bool on = MODE == 0 ? true : MODE == 1 ? lane < 32 : MODE == 2 ? (lane & 1) == 0 : MODE == 3 ? lane < 16 : MODE == 4 ? lane >= 32 : (lane & 32) == ((lane & 1) << 5);
    float a0 = 1.0f + lane * 1e-3f, a1 = a0 + 0.1f, a2 = a0 + 0.2f, a3 = a0 + 0.3f, a4 = a0 + 0.4f, a5 = a0 + 0.5f, a6 = a0 + 0.6f, a7 = a0 + 0.7f;
    const float b = 0.99999f + out[0], c = 1e-6f;
    if (on)
        for (int i = 0; i < iters; i++)
            asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                         "v_max_f32 %0, %0, %8\nv_max_f32 %1, %1, %8\nv_max_f32 %2, %2, %8\nv_max_f32 %3, %3, %8\n"
                         "v_max_f32 %4, %4, %8\nv_max_f32 %5, %5, %8\nv_max_f32 %6, %6, %8\nv_max_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    out[1 + blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * 16, iters = 200000;
    float* out; CHECK(hipMalloc(&out, sizeof(float) * (1 + (size_t)blocks * 64))); CHECK(hipMemset(out, 0, sizeof(float) * (1 + (size_t)blocks * 64)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[6] = { "all 64 lanes", "lanes 0-31", "even lanes (32)", "lanes 0-15", "lanes 32-63", "16 lanes in each half" };
    void (*fns[6])(float*, int) = { k<0>, k<1>, k<2>, k<3>, k<4>, k<5> };
    printf("{\"what\": \"8 v_fma_f32 + 8 v_max_f32 per iteration, 4 waves per SIMD, ms for %d iterations\", \"rows\": [\n", iters);
    for (int m = 0; m < 6; m++)
    {
        hipLaunchKernelGGL(fns[m], dim3(blocks), dim3(64), 0, 0, out, 1000);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(fns[m], dim3(blocks), dim3(64), 0, 0, out, iters);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf(" {\"exec\": \"%s\", \"ms\": %.3f}%s\n", names[m], ms, m < 5 ? "," : "");
    }
    printf("]}\n");
    return 0;
}
