// profiles/calib_coop.hip -- is a per-lane gather of 96-byte records bound by the L1 tag rate
// (64 different lines per wave-instruction)? Compares, on the same index streams,
//   naive: lane l loads the six 16-byte pieces of ITS record (6 wave-instructions, 64 lines each)
//   coop : the 64 records of a wave are loaded piece-major, six consecutive lanes per record
//          (6 wave-instructions, ~11 records = ~16 lines each), staged in LDS, then every lane
//          reads its own record back from LDS
// for (a) uniformly random records of a 192 MB table and (b) records random within a sliding
// window of 32768 records (3 MB: L2 hits, the regime of the BVH dual traversal).
//   hipcc -O3 --offload-arch=gfx950 profiles/calib_coop.hip -o build/calib_coop
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x)                                                                                                       \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                               \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

struct rec96 {
    float4 p[6];
};

__device__ inline uint32_t hash32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__device__ inline uint32_t pick(uint32_t i, uint32_t nrec, uint32_t window)
{
    const uint32_t h = hash32(i * 2654435761u + 12345u);
    if (window == 0u) {
        return h % nrec;
    }
    const uint32_t base = static_cast<uint32_t>((static_cast<uint64_t>(i) * nrec) >> 26) ; // i < 2^26: slides over the table
    return (base + h % window) % nrec;
}

__global__ __launch_bounds__(128) void k_naive(const rec96 *tab, uint32_t nrec, uint32_t n, uint32_t window, float *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const rec96 r = tab[pick(i, nrec, window)];
    const float s = r.p[0].x + r.p[1].y + r.p[2].z + r.p[3].w + r.p[4].x + r.p[5].y;
    if (s == 12345.678f) {
        out[i] = s;
    }
}

constexpr int REC_LDS = 7; // float4 slots per record in LDS (96 B + 16 B pad: conflict-free b128 reads)

__global__ __launch_bounds__(128) void k_coop(const rec96 *tab, uint32_t nrec, uint32_t n, uint32_t window, float *out)
{
    __shared__ float4 stage[2][64 * REC_LDS];
    __shared__ uint32_t sidx[2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    sidx[wave][lane] = i < n ? pick(i, nrec, window) : 0u;
    // (one wave: LDS accesses of a wave are ordered; the compiler inserts the waitcnt)
    __builtin_amdgcn_wave_barrier();
    const float4 *t4 = reinterpret_cast<const float4 *>(tab);
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const int q = p * 64 + lane, r = q / 6, k = q - r * 6;
        stage[wave][r * REC_LDS + k] = t4[static_cast<size_t>(sidx[wave][r]) * 6u + k];
    }
    __builtin_amdgcn_wave_barrier();
    const float4 *m = &stage[wave][lane * REC_LDS];
    const float s = m[0].x + m[1].y + m[2].z + m[3].w + m[4].x + m[5].y;
    if (i < n && s == 12345.678f) {
        out[i] = s;
    }
}

int main()
{
    const uint32_t nrec = 2000000u, n = 60000000u;
    rec96 *d_tab = nullptr;
    float *d_out = nullptr;
    CHECK(hipMalloc(&d_tab, static_cast<size_t>(nrec) * sizeof(rec96)));
    CHECK(hipMemset(d_tab, 0, static_cast<size_t>(nrec) * sizeof(rec96)));
    CHECK(hipMalloc(&d_out, static_cast<size_t>(n) * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (uint32_t window : {0u, 32768u}) {
        for (int variant = 0; variant < 2; ++variant) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                if (variant == 0) {
                    hipLaunchKernelGGL(k_naive, dim3((n + 127) / 128), dim3(128), 0, 0, d_tab, nrec, n, window, d_out);
                } else {
                    hipLaunchKernelGGL(k_coop, dim3((n + 127) / 128), dim3(128), 0, 0, d_tab, nrec, n, window, d_out);
                }
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            std::printf("{\"pattern\": \"%s\", \"variant\": \"%s\", \"gathers\": %u, \"ms\": %.4f, \"Grec_per_s\": %.2f}\n",
                        window ? "window 32768 records (L2)" : "uniform over 192 MB", variant ? "coop (6 lanes per record, LDS)" : "naive (record per lane)", n,
                        best, n / best / 1e6);
        }
    }
    return 0;
}
