// ubench_xwg_sync.hip -- what a hand-off between workgroups costs per round on gfx950 (for the chain of a history-carrying
// stream spread over several CUs): M workgroups in a ring, every round each one stores a row, releases it
// (agent scope), publishes its round counter, and waits (bounded) for the counter of the workgroup before it to be at least
// `round - slack` before it reads that workgroup's row.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/ubench_xwg_sync.hip -o tools/bin/ubench_xwg_sync && tools/bin/ubench_xwg_sync
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE>  // 0: no sync at all (timing floor); 1: agent-scope fences + flags; 2: no fences -- the rows themselves go through
                     // agent-scope relaxed atomics (sc1 stores / loads), ordered by s_waitcnt vmcnt(0) and the barrier
__global__ void __launch_bounds__(512) k(unsigned* rows, unsigned* flags, unsigned* bad, int rounds, int slack, int work, int M)
{
    const int ring = blockIdx.x / M, m = blockIdx.x % M, pred = (m + M - 1) % M;
    unsigned* my = rows + (size_t)(ring * M + m) * 64 * 4096;
    const unsigned* his = rows + (size_t)(ring * M + pred) * 64 * 4096;
    unsigned* myflag = flags + (ring * M + m) * 32;
    unsigned* hisflag = flags + (ring * M + pred) * 32;
    const int tid = threadIdx.x;
    unsigned acc = tid, errs = 0;
    unsigned seen = 0;
    __shared__ int dead;  // a wait timed out: no more waiting in this workgroup
    if (tid == 0) dead = 0;
    __syncthreads();
    for (int r = 1; r <= rounds; ++r) {
        // "work": dependent VALU chain
        for (int i = 0; i < work; ++i) acc = acc * 1664525u + 1013904223u;
        if (MODE == 2) __hip_atomic_store(my + (size_t)(r & 63) * 4096 + tid, (unsigned)r * 4096u + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else my[(size_t)(r & 63) * 4096 + tid] = (unsigned)r * 4096u + tid;   // the row of this round (ring of 64 rows)
        if (MODE == 0) { __syncthreads(); continue; }
        if (MODE == 1) __threadfence();       // release, agent scope
        else { __atomic_signal_fence(__ATOMIC_SEQ_CST); __builtin_amdgcn_s_waitcnt(0x0f70); }
        __syncthreads();
        if (tid == 0) __hip_atomic_store(myflag, (unsigned)r, MODE == 1 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int need = r - slack;           // the predecessor's row `need` is read next
        if (need >= 1) {
            if (tid == 0) {
                int spins = 0;
                while (!dead && (seen = __hip_atomic_load(hisflag, MODE == 1 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (unsigned)need) {
                    if (++spins > (1 << 20)) { atomicAdd(bad + 1, 1u); dead = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            unsigned v;
            if (MODE == 1) {
                __threadfence();              // acquire
                v = __builtin_nontemporal_load(his + (size_t)(need & 63) * 4096 + tid);
            } else {
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                v = __hip_atomic_load(his + (size_t)(need & 63) * 4096 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (v != (unsigned)need * 4096u + tid) ++errs;
            acc += v;
        }
    }
    if (errs) atomicAdd(bad, errs);
    if (acc == 0x12345678u) bad[2] = acc;
}

template <int MODE>
static double run(int rings, int M, int rounds, int slack, int work, unsigned* rows, unsigned* flags, unsigned* bad)
{
    CHECK(hipMemset(flags, 0, 4096 * sizeof(unsigned)));
    CHECK(hipMemset(bad, 0, 16));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(rings * M), dim3(512), 0, 0, rows, flags, bad, rounds, slack, work, M);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    unsigned h[4];
    CHECK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    if (h[0] || h[1]) printf("   !! %u stale reads, %u timeouts\n", h[0], h[1]);
    return ms * 1e3 / rounds;
}

int main()
{
    unsigned *rows, *flags, *bad;
    CHECK(hipMalloc(&rows, (size_t)128 * 64 * 4096 * 4));
    CHECK(hipMalloc(&flags, 4096 * sizeof(unsigned)));
    CHECK(hipMalloc(&bad, 16));
    CHECK(hipMemset(rows, 0, (size_t)128 * 64 * 4096 * 4));
    const int rounds = 4000;
    printf("us per round; 9 rings of M workgroups of 512 threads\n");
    printf("%4s %6s %6s | %9s %9s %9s\n", "M", "work", "slack", "no sync", "fences", "sc1 rows");
    for (int M : {2, 4, 8}) {
        for (int work : {0, 200, 1000}) {
            for (int slack : {0, 1, 3}) {
                run<1>(9, M, 200, slack, work, rows, flags, bad);
                const double t0 = run<0>(9, M, rounds, slack, work, rows, flags, bad);
                const double t1 = run<1>(9, M, rounds, slack, work, rows, flags, bad);
                const double t2 = run<2>(9, M, rounds, slack, work, rows, flags, bad);
                printf("%4d %6d %6d | %9.3f %9.3f %9.3f\n", M, work, slack, t0, t1, t2);
            }
        }
    }
    return 0;
}
