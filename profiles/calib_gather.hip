// profiles/calib_gather.hip -- calibration of rocprofv3's FETCH_SIZE on gfx950 for the access
// pattern of the dual BVH traversal (k_bp_dual: one 96-byte record per lane at an unrelated
// address), next to the wide streaming read the microarchitecture guide documents (FETCH_SIZE =
// 1/2 of the bytes). Known byte counts per launch are printed; profiles/calib_summarise.py sets
// them against the counter.
//
//   hipcc -O3 --offload-arch=gfx950 profiles/calib_gather.hip -o build/calib_gather
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o c -- build/calib_gather
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                                                       \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                               \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

struct rec96 {
    float4 a, b, c, d, e, f;
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

// One random 96-byte record per lane (six 16-byte loads), nrec records in the table.
__global__ __launch_bounds__(128) void k_calib_gather96(const rec96 *tab, uint32_t nrec, uint32_t n, uint32_t seed, float *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const rec96 r = tab[hash32(i * 2654435761u + seed) % nrec];
    const float s = r.a.x + r.b.y + r.c.z + r.d.w + r.e.x + r.f.y;
    if (s == 12345.678f) {
        out[i] = s;
    }
}

// Wide coalesced streaming read of the whole table, 16 bytes per lane.
__global__ __launch_bounds__(256) void k_calib_stream16(const float4 *tab, uint64_t n16, float *out)
{
    float s = 0.f;
    for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < n16; i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const float4 v = tab[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) {
        out[threadIdx.x] = s;
    }
}

int main()
{
    const uint32_t sizes[2] = {2000000u, 12000000u}; // 192 MB (fits the 256 MiB Infinity Cache), 1.15 GB (does not)
    const uint32_t n = 60000000u;                     // gathers per launch (~ node records read by one traversal)
    float *d_out = nullptr;
    CHECK(hipMalloc(&d_out, static_cast<size_t>(n) * sizeof(float)));
    for (int t = 0; t < 2; ++t) {
        const uint32_t nrec = sizes[t];
        rec96 *d_tab = nullptr;
        CHECK(hipMalloc(&d_tab, static_cast<size_t>(nrec) * sizeof(rec96)));
        CHECK(hipMemset(d_tab, 0, static_cast<size_t>(nrec) * sizeof(rec96)));
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_calib_gather96, dim3((n + 127) / 128), dim3(128), 0, 0, d_tab, nrec, n, 17u * rep + t, d_out);
            const uint64_t n16 = static_cast<uint64_t>(nrec) * 6u;
            hipLaunchKernelGGL(k_calib_stream16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const float4 *>(d_tab), n16, d_out);
        }
        CHECK(hipDeviceSynchronize());
        std::printf("{\"table_records\": %u, \"table_bytes\": %llu, \"gathers\": %u, \"gather_bytes_96\": %llu, \"gather_bytes_2x64B_sectors\": %llu, "
                    "\"gather_bytes_128B_lines_avg\": %llu, \"stream_bytes\": %llu}\n",
                    nrec, static_cast<unsigned long long>(nrec) * 96ull, n, static_cast<unsigned long long>(n) * 96ull,
                    static_cast<unsigned long long>(n) * 128ull, static_cast<unsigned long long>(n) * 192ull,
                    static_cast<unsigned long long>(nrec) * 96ull);
        CHECK(hipFree(d_tab));
    }
    CHECK(hipFree(d_out));
    return 0;
}
