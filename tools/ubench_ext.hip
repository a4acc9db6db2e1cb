// ubench_ext.hip — throughput of the production backward_ext_coop() in isolation:
// every lane runs a chain of extensions on pseudo-random intervals over a random "index".
//   variant 0: s = 1 (one block per extension)      variant 1: s ~ U[0, 4096) (mostly two blocks)
//   variant 2: like 0 but with ~300 extra VALU ops per step (models the state machine)
#include "../bwa-mem-scale_amd/csrc/fmi_seed.hip"
#include <vector>
#include <cstdlib>
using namespace bwams;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mixu(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

template <int V>
__global__ __launch_bounds__(256) void ext_chain(DevFmi f, int64_t nrows, int steps, uint64_t *out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t h = mixu(tid + 12345);
    uint64_t acc = 0;
    for (int st = 0; st < steps; ++st) {
        const int64_t k = (int64_t)(h % (uint64_t)(nrows - 8192));
        const int64_t s = V == 1 ? (int64_t)((h >> 40) & 4095) : 1;
        int64_t nk, nl, ns;
        backward_ext_coop(f, true, k, 7, s, (int)(h & 3), nk, nl, ns);
        uint64_t r = (uint64_t)nk ^ (uint64_t)ns ^ (uint64_t)nl;
        if (V == 2) {
#pragma unroll 1
            for (int i = 0; i < 40; ++i) r = mixu(r + i);
        }
        acc += r;
        h = mixu(h ^ r);
    }
    out[tid] = acc;
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 1.0;
    const int steps = argc > 2 ? atoi(argv[2]) : 200;
    const int64_t nblk = (int64_t)(gib * 1073741824.0 / 64);
    uint4 *tab; CK(hipMalloc(&tab, (size_t)nblk * 64));
    {
        std::vector<uint32_t> hbuf(1 << 24);
        uint64_t x = 88172645463325252ULL;
        for (auto &w : hbuf) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (uint32_t)x; }
        for (size_t o = 0; o < (size_t)nblk * 64; o += hbuf.size() * 4)
            CK(hipMemcpy((char *)tab + o, hbuf.data(), std::min(hbuf.size() * 4, (size_t)nblk * 64 - o), hipMemcpyHostToDevice));
    }
    DevFmi f{}; f.cp = tab; f.sentinel = 5; for (int i = 0; i < 5; ++i) f.count[i] = i * 1000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("index %.1f GiB, %d steps\n", gib, steps);
    for (int wpc : {8, 16, 20}) {
        const int blocks = prop.multiProcessorCount * wpc / 4;
        uint64_t *out; CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
        for (int v = 0; v < 3; ++v) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (v == 0) ext_chain<0><<<blocks, 256>>>(f, nblk * 64, steps, out);
                if (v == 1) ext_chain<1><<<blocks, 256>>>(f, nblk * 64, steps, out);
                if (v == 2) ext_chain<2><<<blocks, 256>>>(f, nblk * 64, steps, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double n = (double)blocks * 256 * steps;
            printf("waves/CU %2d variant %d: %8.3f ms  %6.2f G ext/s  step %.0f ns\n", wpc, v, best, n / best / 1e6, best * 1e6 / steps);
        }
        CK(hipFree(out));
    }
    return 0;
}
