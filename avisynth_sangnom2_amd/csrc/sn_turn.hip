// sn_turn.hip -- TurnRight / TurnLeft of a plane on the device.
//
// Anti-aliasing scripts run SangNom2 twice around a quarter turn (TurnLeft().SangNom2().TurnRight().SangNom2(),
// README.md:3 of the reference: "mainly used in anti-aliasing scripts").  With the turn on the device the
// frame stays in HBM between the two passes instead of crossing PCIe four times.
//   right (clockwise):      dst[y'][x'] = src[H-1-x'][y'],  dst is H wide and W high
//   left (anti-clockwise):  dst[y'][x'] = src[x'][W-1-y']
// 64 x 64 sample tiles go through LDS so that both the reads and the writes are row-contiguous; HBM-bound:
// 2 x W x H x B bytes per plane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sn_internal.h"

namespace sn {

constexpr int kTile = 64;

template <class T>
__global__ void __launch_bounds__(256) k_turn(const uint8_t* src, int64_t sfs, int spitch, int w, int h, uint8_t* dst, int64_t dfs, int dpitch,
                                             int right, int dword_ok)
{
    constexpr int P = 4 / (int)sizeof(T);  // samples per dword
    constexpr int RD = kTile / P;          // dwords per tile row
    __shared__ uint32_t tile32[kTile][RD + 1];
    T(*tile)[(RD + 1) * P] = reinterpret_cast<T(*)[(RD + 1) * P]>(tile32);
    const int f = blockIdx.z;
    const int x0 = blockIdx.x * kTile, y0 = blockIdx.y * kTile;  // tile origin in the source
    const uint8_t* s = src + (int64_t)f * sfs;
    uint8_t* d = dst + (int64_t)f * dfs;

    // Whole tiles with dword-aligned rows move as dwords on both sides (a byte-wide global access wastes most of
    // the 64 bytes a wave instruction moves); ragged edges and odd alignments take the sample-wide path.
    if (dword_ok && x0 + kTile <= w && y0 + kTile <= h) {
        for (int i = threadIdx.x; i < kTile * RD; i += 256) {
            const int r = i / RD, q = i % RD;
            tile32[r][q] = reinterpret_cast<const uint32_t*>(s + (int64_t)(y0 + r) * spitch)[x0 / P + q];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < kTile * RD; i += 256) {
            const int r = i / RD, q = i % RD;  // r: source column inside the tile = destination row; q: dword of that row
            uint32_t v = 0;
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const int dxj = P * q + j;  // position along the destination row, inside the tile
                const int sy = right ? kTile - 1 - dxj : dxj;
                v |= (uint32_t)tile[sy][r] << (8 * (int)sizeof(T) * j);
            }
            const int dy = right ? x0 + r : w - 1 - (x0 + r);
            const int dx0 = right ? h - y0 - kTile : y0;  // first destination column of the tile
            reinterpret_cast<uint32_t*>(d + (int64_t)dy * dpitch)[dx0 / P + q] = v;
        }
        return;
    }

    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4 threads
    for (int r = ty; r < kTile; r += 4) {
        const int x = x0 + tx, y = y0 + r;
        if (x < w && y < h) tile[r][tx] = reinterpret_cast<const T*>(s + (int64_t)y * spitch)[x];
    }
    __syncthreads();
    // destination rows correspond to source columns; a destination row segment is a source column segment
    for (int r = ty; r < kTile; r += 4) {
        const int sx = x0 + r;
        if (sx >= w) continue;
        // right: dst[sx][h-1-y] = src[y][sx] (x' runs backwards in y);  left: dst[w-1-sx][y] = src[y][sx]
        const int sy = right ? y0 + (kTile - 1 - tx) : y0 + tx;
        if (sy >= h) continue;
        const int dx = right ? h - 1 - sy : sy;
        const int dy = right ? sx : w - 1 - sx;
        reinterpret_cast<T*>(d + (int64_t)dy * dpitch)[dx] = tile[sy - y0][r];
    }
}

hipError_t launch_turn(hipStream_t st, int bytes, int right, int nframes, const uint8_t* src, int64_t sfs, int spitch, int w, int h, uint8_t* dst,
                       int64_t dfs, int dpitch)
{
    if (nframes <= 0 || w <= 0 || h <= 0) return hipSuccess;
    dim3 grid((w + kTile - 1) / kTile, (h + kTile - 1) / kTile, nframes), block(256);
    const int per = 4 / bytes;  // samples per dword: the turned tile must start on a dword of the destination row
    const int dword_ok = ((uintptr_t)src % 4 == 0 && (uintptr_t)dst % 4 == 0 && spitch % 4 == 0 && dpitch % 4 == 0 && sfs % 4 == 0 && dfs % 4 == 0 &&
                          (!right || h % per == 0))
                             ? 1
                             : 0;
    switch (bytes) {
    case 1: hipLaunchKernelGGL(k_turn<uint8_t>, grid, block, 0, st, src, sfs, spitch, w, h, dst, dfs, dpitch, right, dword_ok); break;
    case 2: hipLaunchKernelGGL(k_turn<uint16_t>, grid, block, 0, st, src, sfs, spitch, w, h, dst, dfs, dpitch, right, dword_ok); break;
    default: hipLaunchKernelGGL(k_turn<uint32_t>, grid, block, 0, st, src, sfs, spitch, w, h, dst, dfs, dpitch, right, dword_ok); break;
    }
    return hipGetLastError();
}

}  // namespace sn
