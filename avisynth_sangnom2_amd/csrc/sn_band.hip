// sn_band.hip -- the check behind the row-band sweeps (mode kBand, sn_fused_v3_common.h).
//
// A band sweep starts a few rows early from a guessed (zero) state of the stage-2 recurrence (SangNom2.cpp:126-159) and
// leaves two snapshots of its state: on entering its first own row and after its last one.  Band 0 starts at the top of
// the plane, where zero IS the state; if band b's end snapshot equals band b+1's start snapshot for every b, every band
// has produced exactly the rows the top-to-bottom sweep would have.  This kernel compares the snapshots (one workgroup
// per boundary and frame) and raises the frame's flag otherwise; the pool-path launches that follow are guarded by the
// same flag and redo such a frame from scratch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sn_internal.h"

namespace sn {

__global__ void __launch_bounds__(256) k_band_verify(const uint32_t* state, int words, int nbands, int32_t* flags, int64_t* fallbacks,
                                                     int64_t* host_mirror)
{
    const int b = blockIdx.x, f = blockIdx.y;
    const uint4* end = reinterpret_cast<const uint4*>(state + ((int64_t)(f * nbands + b) * 2 + 1) * words);
    const uint4* start = reinterpret_cast<const uint4*>(state + ((int64_t)(f * nbands + b + 1) * 2 + 0) * words);
    unsigned diff = 0;
    for (int i = threadIdx.x; i < words / 4; i += blockDim.x) {
        const uint4 x = end[i], y = start[i];
        diff |= (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w);
    }
    if (__syncthreads_or(diff != 0) && threadIdx.x == 0) {
        if (atomicExch(&flags[f], 1) == 0 && fallbacks) {
            const unsigned long long count = atomicAdd(reinterpret_cast<unsigned long long*>(fallbacks), 1ull) + 1ull;
            if (host_mirror) __hip_atomic_store(host_mirror, (int64_t)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

hipError_t launch_band_verify(hipStream_t s, const uint32_t* state, int threads, int nbands, int nframes, int32_t* flags, int64_t* fallbacks,
                              int64_t* host_mirror)
{
    if (nbands < 2) return hipSuccess;
    hipLaunchKernelGGL(k_band_verify, dim3(nbands - 1, nframes), dim3(256), 0, s, state, kBuffers * 8 * threads, nbands, flags, fallbacks, host_mirror);
    return hipGetLastError();
}

}  // namespace sn
