// ubench_gather.hip — how fast can gfx950 do dependent random 64-byte block reads, by access shape?
// (design input for the FM-index kernels; not part of the product library)
//   mode 0: lane-per-chain, 4 x dwordx4 per 64 B block            (current SMEM kernel shape)
//   mode 1: quad-cooperative: 4 lanes read the 4 x 16 B pieces of one block, 4 blocks per quad per step
//   mode 2: lane-per-chain, 2 x dwordx4 = 32 B (compact-block layout candidate)
//   mode 3: lane-per-chain, 1 x dwordx4 = 16 B
//   mode 4: lane-per-chain, 8 x dwordx4 = 128 B line
// Each chain: pos = hash(pos ^ data) so that loads are dependent, `steps` steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void chase(const uint4 *__restrict__ tab, uint64_t nblk, int steps, uint64_t *out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t pos = mix(tid * 0x9e3779b97f4a7c15ULL + 1);
    uint64_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        const uint64_t b = pos % nblk;
        uint32_t v = 0;
        if (MODE == 0) {
            const uint4 *p = tab + b * 4;
            uint4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
            v = a0.x ^ a1.y ^ a2.z ^ a3.w ^ a0.w ^ a1.x ^ a2.y ^ a3.z;
        } else if (MODE == 1) {
            const int lane = threadIdx.x & 63, q = lane & 3;
            uint32_t r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // block index of quad member j, broadcast inside the quad
                const uint32_t blo = __shfl((uint32_t)b, (lane & ~3) | j), bhi = __shfl((uint32_t)(b >> 32), (lane & ~3) | j);
                const uint64_t bj = ((uint64_t)bhi << 32) | blo;
                const uint4 a = tab[bj * 4 + q];
                r[j] = a.x ^ a.y ^ a.z ^ a.w;
            }
            // each member collects the 4 partial words of its own block
            uint32_t mine = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const uint32_t t = __shfl(r[j], (lane & ~3) | qq);
                    if (q == j) mine ^= t;
                }
            }
            v = mine;
        } else if (MODE == 2) {
            const uint4 *p = tab + b * 4;
            uint4 a0 = p[0], a1 = p[1];
            v = a0.x ^ a1.y ^ a0.w ^ a1.x;
        } else if (MODE == 3) {
            uint4 a0 = tab[b * 4];
            v = a0.x ^ a0.w;
        } else {
            const uint4 *p = tab + (b & ~1ull) * 4;
            uint32_t t = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { uint4 a = p[j]; t ^= a.x ^ a.w; }
            v = t;
        }
        acc += v;
        pos = mix(pos ^ v);
    }
    out[tid] = acc;
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const int steps = argc > 2 ? atoi(argv[2]) : 200;
    const uint64_t nblk = (uint64_t)(gib * 1073741824.0 / 64);
    uint4 *tab; CK(hipMalloc(&tab, nblk * 64));
    {   // fill with pseudo-random words
        std::vector<uint32_t> h(1 << 24);
        uint64_t x = 88172645463325252ULL;
        for (auto &w : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (uint32_t)x; }
        for (uint64_t o = 0; o < nblk * 64; o += h.size() * 4) {
            uint64_t n = std::min<uint64_t>(h.size() * 4, nblk * 64 - o);
            CK(hipMemcpy((char *)tab + o, h.data(), n, hipMemcpyHostToDevice));
        }
    }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("table %.1f GiB, %d steps, %d CUs\n", gib, steps, prop.multiProcessorCount);
    for (int wpc : {8, 16, 20, 32}) {           // waves per CU
        const int blocks = prop.multiProcessorCount * wpc / 4;
        uint64_t *out; CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
        for (int mode = 0; mode < 5; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                switch (mode) {
                    case 0: chase<0><<<blocks, 256>>>(tab, nblk, steps, out); break;
                    case 1: chase<1><<<blocks, 256>>>(tab, nblk, steps, out); break;
                    case 2: chase<2><<<blocks, 256>>>(tab, nblk, steps, out); break;
                    case 3: chase<3><<<blocks, 256>>>(tab, nblk, steps, out); break;
                    case 4: chase<4><<<blocks, 256>>>(tab, nblk, steps, out); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double nacc = (double)blocks * 256 * steps;
            const int bytes[5] = {64, 64, 32, 16, 128};
            printf("waves/CU %2d mode %d: %8.3f ms  %7.2f G blocks/s  %7.1f GB/s useful  latency/step %.0f ns\n", wpc, mode, best,
                   nacc / best / 1e6, nacc * bytes[mode] / best / 1e6, best * 1e6 / steps);
        }
        CK(hipFree(out));
    }
    return 0;
}
