// Cache-hierarchy gather ceiling for the BVH walk on gfx950 (MI355X): every lane chases its own chain of fixed-size
// records (the next index comes out of the record just loaded - a dependent load per step, as a BVH descent), 4 waves per
// SIMD (what trace_kernel runs with), for
//   * record sizes 16 .. 128 B (one BVH node is 64 B), tables from L2-resident to HBM-resident, uniformly random records;
//   * a varying share of active lanes (the walk runs with 50-80 % of its lanes);
//   * "tree" chains: step s reads a random node of level (s mod D) of a complete binary tree - the top levels are hot in
//     L1/L2 as in a real walk;
//   * "quad" loads: four neighbouring lanes fetch the four 16-B quarters of ONE 64-B record (does the address coalescer
//     merge them?).
// Prints records/s and bytes/s per CU and the time of one dependent step; profiles/r02/gather_ceiling.json is this
// program's output.
//
//   hipcc -O3 --offload-arch=gfx950 -o gather_bench gather_bench.hip && ./gather_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x = x * 747796405u + 2891336453u;
    unsigned w = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
    return (w >> 22u) ^ w;
}

enum { MODE_UNIFORM = 0, MODE_TREE = 1, MODE_QUAD = 2 };

// table[r * F4 + 0].x holds random bits; the chain r -> f(bits)
template <int F4, int MODE>
__global__ __launch_bounds__(64, 4) void chase(const float4* __restrict__ table, unsigned nrec, int steps, int active, int depth, float* out)
{
    __shared__ int pad[2048];           // 8 KiB per wave, as the walk's LDS stack: keeps residency at 16 waves per CU
    const int lane = threadIdx.x;
    pad[lane] = lane;
    unsigned r = hash32(blockIdx.x * 64u + lane) % nrec;
    if (MODE == MODE_TREE) r = 0;
    float acc = 0.0f;
    if (lane < active)
        for (int s = 0; s < steps; s++)
        {
            float4 q[F4];
            if (MODE == MODE_QUAD)
            {
                // instruction k serves the records of lanes 4q + k: every quad reads 64 contiguous bytes
                static_assert(MODE != MODE_QUAD || F4 == 4, "quad mode is for 64-byte records");
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const unsigned rk = (unsigned)__shfl((int)r, (lane & ~3) + k);
                    q[k] = table[(size_t)rk * 4 + (lane & 3)];
                }
                // (no redistribution: this measures the load side only; lane l uses what it got)
            }
            else
            {
                const float4* p = table + (size_t)r * F4;
#pragma unroll
                for (int k = 0; k < F4; k++) q[k] = p[k];
            }
#pragma unroll
            for (int k = 1; k < F4; k++) acc += q[k].x + q[k].w;
            const unsigned bits = __float_as_uint(q[0].x) + (unsigned)s;
            if (MODE == MODE_TREE)
            {
                const int level = (s + 1) % depth;                          // next node: a random one of that level
                r = ((1u << level) - 1u) + (bits & ((1u << level) - 1u));
                if (r >= nrec) r = bits % nrec;
            }
            else r = bits % nrec;
            acc += q[0].y;
        }
    out[blockIdx.x * 64 + lane] = acc + (float)pad[(lane * 7) & 63];
}

__global__ void fill(float4* table, size_t n_f4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_f4) { unsigned h = hash32((unsigned)i); table[i] = make_float4(__uint_as_float(h >> 1), 0.0f, 0.0f, 0.0f); }
}

template <int F4, int MODE>
static float run(const float4* table, unsigned nrec, int steps, int active, int depth, float* out, int blocks, hipEvent_t e0, hipEvent_t e1)
{
    float ms = 0;
    for (int rep = 0; rep < 2; rep++)
    {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((chase<F4, MODE>), dim3(blocks), dim3(64), 0, 0, table, nrec, steps, active, depth, out);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    return ms;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t max_bytes = (size_t)1 << 30;
    float4* table; float* out;
    CHECK(hipMalloc(&table, max_bytes));
    CHECK(hipMalloc(&out, sizeof(float) * (size_t)cus * 16 * 64));
    hipLaunchKernelGGL(fill, dim3((unsigned)((max_bytes / 16 + 255) / 256)), dim3(256), 0, 0, table, max_bytes / 16);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int steps = 2000, blocks = cus * 16;
    printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_cu\": 16, \"steps\": %d, \"rows\": [\n", prop.name, cus, steps);
    bool first = true;
    auto row = [&](const char* mode, int rec, double mib, int active, int depth, float ms) {
        const double recs = (double)blocks * active * steps;
        printf("%s {\"mode\": \"%s\", \"record_bytes\": %d, \"table_MiB\": %.2f, \"active_lanes\": %d, \"tree_depth\": %d, \"ms\": %.3f, "
               "\"Grecords_per_s_per_cu\": %.4f, \"GBps_per_cu\": %.2f, \"TBps_chip\": %.2f, \"ns_per_dependent_step\": %.1f}",
               first ? "" : ",\n", mode, rec, mib, active, depth, ms, recs / (ms * 1e-3) / cus / 1e9,
               recs * rec / (ms * 1e-3) / cus / 1e9, recs * rec / (ms * 1e-3) / 1e12, ms * 1e6 / steps);
        first = false;
        fflush(stdout);
    };
    const size_t sizes[] = { (size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)192 << 20, (size_t)1 << 30 };
    for (size_t bytes : sizes)
    {
        const double mib = bytes / 1048576.0;
        row("uniform", 16, mib, 64, 0, run<1, MODE_UNIFORM>(table, (unsigned)(bytes / 16), steps, 64, 0, out, blocks, e0, e1));
        row("uniform", 32, mib, 64, 0, run<2, MODE_UNIFORM>(table, (unsigned)(bytes / 32), steps, 64, 0, out, blocks, e0, e1));
        row("uniform", 48, mib, 64, 0, run<3, MODE_UNIFORM>(table, (unsigned)(bytes / 48), steps, 64, 0, out, blocks, e0, e1));
        row("uniform", 64, mib, 64, 0, run<4, MODE_UNIFORM>(table, (unsigned)(bytes / 64), steps, 64, 0, out, blocks, e0, e1));
        row("uniform", 128, mib, 64, 0, run<8, MODE_UNIFORM>(table, (unsigned)(bytes / 128), steps, 64, 0, out, blocks, e0, e1));
        row("quad", 64, mib, 64, 0, run<4, MODE_QUAD>(table, (unsigned)(bytes / 64), steps, 64, 0, out, blocks, e0, e1));
    }
    for (int active : { 48, 32, 16 })
        for (size_t bytes : { (size_t)4 << 20, (size_t)64 << 20 })
            row("uniform", 64, bytes / 1048576.0, active, 0, run<4, MODE_UNIFORM>(table, (unsigned)(bytes / 64), steps, active, 0, out, blocks, e0, e1));
    // tree chains: C4's tree has ~38 k interior nodes (depth 23, 2.4 MB of nodes), C5's ~518 k (depth 24, 33 MB)
    for (int depth : { 12, 15, 17, 19, 21 })
        for (int active : { 64, 48, 32 })
        {
            const unsigned nrec = (1u << depth) - 1u;
            row("tree", 64, nrec * 64.0 / 1048576.0, active, depth, run<4, MODE_TREE>(table, nrec, steps, active, depth, out, blocks, e0, e1));
        }
    printf("\n]}\n");
    return 0;
}
