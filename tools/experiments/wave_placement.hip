// wave_placement.hip -- where do the waves of a 256-thread, 256-VGPR workgroup land?  (Round 4: the two edge waves of the
// 2160p sweep do 4 % more work than the two inner ones -- profiles/r4_ab_experiments.md 10. -- and whether that can be
// balanced by swapping strips depends on which SIMD a wave index gets and which workgroups share a CU.)
// Every wave records HW_ID, XCC_ID and the shader clock when it starts; the host prints, per workgroup, the SIMD of each
// wave index and, per CU, the workgroups that ran on it in order of their start.
//   hipcc --offload-arch=gfx950 -O2 -o wave_placement wave_placement.hip && ./wave_placement [workgroups = 1024] [spin = 200000]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <map>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

struct Rec { unsigned hw_id, xcc_id; unsigned long long t0, t1; };

__global__ void __launch_bounds__(256, 2) k_place(Rec* out, int spin, unsigned* sink)
{
    extern __shared__ unsigned lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("v_mov_b32 v255, 0" ::: "v255");  // 256 VGPRs: two waves per SIMD, like the sweep
    unsigned x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;
    lds[threadIdx.x] = x;
    __syncthreads();
    if (x == 0x12345678u) *sink = lds[(threadIdx.x + 1) & 255];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = Rec{hw, xcc, t0, __builtin_amdgcn_s_memtime()};
}

int main(int argc, char** argv)
{
    const int nwg = argc > 1 ? atoi(argv[1]) : 1024, spin = argc > 2 ? atoi(argv[2]) : 200000;
    Rec* d;
    unsigned* sink;
    if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(Rec) * 4 * nwg) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&sink), 4) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_place, dim3(nwg), dim3(256), 70 * 1024, 0, d, spin, sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<Rec> r(4 * nwg);
    if (hipMemcpy(r.data(), d, sizeof(Rec) * 4 * nwg, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    // gfx9 HW_ID: wave_id 3:0, simd_id 5:4, pipe 7:6, cu_id 11:8, sh_id 12, se_id 15:13
    auto simd = [](unsigned h) { return (h >> 4) & 3; };
    auto cu_key = [](const Rec& x) { return (x.xcc_id & 15) << 16 | ((x.hw_id >> 13) & 7) << 8 | ((x.hw_id >> 12) & 1) << 4 | ((x.hw_id >> 8) & 15); };
    std::map<unsigned, int> orders;  // SIMDs of waves 0..3 as a 4-digit number -> count
    std::map<unsigned, std::vector<int>> on_cu;
    int split = 0;
    for (int g = 0; g < nwg; ++g) {
        unsigned o = 0;
        for (int w = 0; w < 4; ++w) o = o * 10 + simd(r[g * 4 + w].hw_id);
        ++orders[o];
        for (int w = 1; w < 4; ++w) split += cu_key(r[g * 4 + w]) != cu_key(r[g * 4]);
        on_cu[cu_key(r[g * 4])].push_back(g);
    }
    printf("%d workgroups on %zu CUs; waves of a workgroup on different CUs: %d\n", nwg, on_cu.size(), split);
    printf("SIMD of wave 0,1,2,3 -> workgroups:");
    for (auto& kv : orders) printf("  %04u: %d", kv.first, kv.second);
    printf("\n");
    int shown = 0;
    for (auto& kv : on_cu) {
        std::vector<int>& v = kv.second;
        std::sort(v.begin(), v.end(), [&](int x, int y) { return r[x * 4].t0 < r[y * 4].t0; });
        if (shown++ < 6) {
            printf("CU %05x:", kv.first);
            for (int g : v) printf(" %d[%u%u%u%u]", g, simd(r[g * 4].hw_id), simd(r[g * 4 + 1].hw_id), simd(r[g * 4 + 2].hw_id), simd(r[g * 4 + 3].hw_id));
            printf("\n");
        }
    }
    // pairs that overlap in time on a CU: parity of blockIdx, of blockIdx / 8, of blockIdx / 256
    long same[3] = {0, 0, 0}, pairs = 0;
    for (auto& kv : on_cu) {
        const std::vector<int>& v = kv.second;
        for (size_t i = 0; i < v.size(); ++i)
            for (size_t j = i + 1; j < v.size(); ++j) {
                const Rec &x = r[v[i] * 4], &y = r[v[j] * 4];
                const unsigned long long lo = std::max(x.t0, y.t0), hi = std::min(x.t1, y.t1);
                if (hi > lo && (hi - lo) * 2 > (x.t1 - x.t0)) {  // together for more than half of a lifetime
                    ++pairs;
                    same[0] += ((v[i] ^ v[j]) & 1) == 0;
                    same[1] += (((v[i] >> 3) ^ (v[j] >> 3)) & 1) == 0;
                    same[2] += (((v[i] >> 8) ^ (v[j] >> 8)) & 1) == 0;
                }
            }
    }
    printf("pairs sharing a CU for most of their life: %ld; same parity of blockIdx %ld, of blockIdx/8 %ld, of blockIdx/256 %ld\n", pairs, same[0], same[1], same[2]);
    return 0;
}
