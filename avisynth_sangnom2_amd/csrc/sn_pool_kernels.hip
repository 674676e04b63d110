// sn_pool_kernels.hip -- the pool path: frame assembly + the three stages as three kernels over
// an HBM-resident scratch pool laid out exactly like the reference's
// (/root/reference/src/SangNom2.cpp:287-310: 9 buffers x (bufferHeight + 1) rows x stride_e).
//
// This path serves every format and reproduces the reference's shared-pool behaviour by
// construction (chroma passes sweep the luma-sized pool and see stale luma results, widths that
// are not a multiple of 32 carry pool state from frame to frame).  It is also the exact fallback
// of the fused kernel.  Stage 2 is a strictly sequential recurrence over rows
// (SangNom2.cpp:126-159), so k_smooth walks the rows with one workgroup per buffer.
#include "sn_fused_v3_common.h"
#include "sn_internal.h"
#include "sn_pixel.h"

namespace sn {

// ------------------------------------------------------------------------------------------------
// Frame assembly: GetFrame's copies, /root/reference/src/SangNom2.cpp:361-391.
// One thread block row per destination line; the source line of each destination line is decided
// arithmetically so that the whole assembly is one dependency-free pass.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int assemble_source_row(const PlaneArgs& p, int y)
{
    if (!p.enabled) return y;  // plane copied unchanged (SangNom2.cpp:369-374)
    int yy = y;
    if (p.offset == 0 && y == p.h_out - 1) yy = p.h_out - 2;  // SangNom2.cpp:380-385
    if (p.offset == 1 && y == 0) yy = 1;                      // SangNom2.cpp:386-391
    if (((yy - p.offset) & 1) != 0) return -1;                // interpolated line: written by stage 3
    return p.dh ? (yy - p.offset) >> 1 : yy;                  // SangNom2.cpp:365 / :376
}

template <int VEC>
__global__ void __launch_bounds__(256) k_assemble(PlaneArgs p, int row_bytes)
{
    const int y = blockIdx.y;
    const int f = blockIdx.z;
    if (p.guard && p.guard[p.guard_single ? 0 : f] == 0) return;
    const int sy = assemble_source_row(p, y);
    if (sy < 0) return;
    const uint8_t* s = p.src + (int64_t)f * p.src_frame_stride + (int64_t)sy * p.src_pitch;
    uint8_t* d = p.dst + (int64_t)f * p.dst_frame_stride + (int64_t)y * p.dst_pitch;
    const int nvec = row_bytes / VEC;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gridDim.x * blockDim.x) {
        if constexpr (VEC == 16) {
            reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
        } else if constexpr (VEC == 4) {
            reinterpret_cast<uint32_t*>(d)[i] = reinterpret_cast<const uint32_t*>(s)[i];
        } else {
            d[i] = s[i];
        }
    }
    if (blockIdx.x == 0) {  // tail bytes
        const int t = nvec * VEC + threadIdx.x;
        if (t < row_bytes) d[t] = s[t];
    }
}

hipError_t launch_assemble(hipStream_t st, const PlaneArgs& p, int bytes, int nframes)
{
    const int row_bytes = p.w * bytes;
    auto aligned = [&](int a) {
        return ((uintptr_t)p.src % a == 0) && ((uintptr_t)p.dst % a == 0) && (p.src_pitch % a == 0) &&
               (p.dst_pitch % a == 0) && (p.src_frame_stride % a == 0) && (p.dst_frame_stride % a == 0);
    };
    const int vec = aligned(16) ? 16 : aligned(4) ? 4 : 1;
    const int nvec = row_bytes / vec;
    int gx = (nvec + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    dim3 grid(gx, p.h_out, nframes), block(256);
    if (vec == 16)
        hipLaunchKernelGGL(k_assemble<16>, grid, block, 0, st, p, row_bytes);
    else if (vec == 4)
        hipLaunchKernelGGL(k_assemble<4>, grid, block, 0, st, p, row_bytes);
    else
        hipLaunchKernelGGL(k_assemble<1>, grid, block, 0, st, p, row_bytes);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Shared tap loader: the 14 edge-clamped taps and the four SangNom values of one pixel pair
// (loadPixel + calculateSangNom, SangNom2.cpp:25-34,60-72,84-103).
// ------------------------------------------------------------------------------------------------
template <class T>
struct Taps {
    using W = typename Px<T>::W;
    W c[7], n[7];  // index k+3 for tap k = -3..3
    W f1, f2, b1, b2;
    __device__ __forceinline__ void load(const T* cl, const T* nl, int x, int w)
    {
#pragma unroll
        for (int k = -3; k <= 3; ++k) {
            int q = x + k;
            q = q < 0 ? 0 : (q > w - 1 ? w - 1 : q);
            c[k + 3] = (W)cl[q];
            n[k + 3] = (W)nl[q];
        }
        f1 = Px<T>::sg(c[2], c[3], c[4]);
        f2 = Px<T>::sg(n[4], n[3], n[2]);
        b1 = Px<T>::sg(c[4], c[3], c[2]);
        b2 = Px<T>::sg(n[2], n[3], n[4]);
    }
};

// ------------------------------------------------------------------------------------------------
// Stage 1: prepareBuffers_c, SangNom2.cpp:74-124.  One thread per pixel of one line pair.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t slot_of(const PoolArgs& pool, int slot0, int f)
{
    const int s = slot0 + f * (pool.slot_step ? pool.slot_step : 1);
    return pool.slot_mod ? s % pool.slot_mod : s;
}

template <class T>
__global__ void __launch_bounds__(256) k_prepare(PlaneArgs p, PoolArgs pool, int slot0)
{
    using P = Px<T>;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int f = blockIdx.z;
    if (p.guard && p.guard[p.guard_single ? 0 : f] == 0) return;
    if (x >= p.w) return;
    const uint8_t* plane = p.dst + (int64_t)f * p.dst_frame_stride;
    const T* cl = reinterpret_cast<const T*>(plane + (int64_t)(p.offset + 2 * y) * p.dst_pitch);
    const T* nl = reinterpret_cast<const T*>(plane + (int64_t)(p.offset + 2 * y + 2) * p.dst_pitch);
    Taps<T> t;
    t.load(cl, nl, x, p.w);
    T* pb = reinterpret_cast<T*>(pool.base + slot_of(pool, slot0, f) * pool.slot_bytes);
    const size_t bufsz = (size_t)pool.stride_e * (pool.bh + 1);
    const size_t at = (size_t)(y + 1) * pool.stride_e + x;
    pb[0 * bufsz + at] = (T)P::adiff(t.c[0], t.n[6]);  // ADIFF_M3_P3
    pb[1 * bufsz + at] = (T)P::adiff(t.c[1], t.n[5]);  // ADIFF_M2_P2
    pb[2 * bufsz + at] = (T)P::adiff(t.c[2], t.n[4]);  // ADIFF_M1_P1
    pb[3 * bufsz + at] = (T)P::adiff(t.f1, t.f2);      // SG_FORWARD
    pb[4 * bufsz + at] = (T)P::adiff(t.c[3], t.n[3]);  // ADIFF_P0_M0
    pb[5 * bufsz + at] = (T)P::adiff(t.b1, t.b2);      // SG_REVERSE
    pb[6 * bufsz + at] = (T)P::adiff(t.c[4], t.n[2]);  // ADIFF_P1_M1
    pb[7 * bufsz + at] = (T)P::adiff(t.c[5], t.n[1]);  // ADIFF_P2_M2
    pb[8 * bufsz + at] = (T)P::adiff(t.c[6], t.n[0]);  // ADIFF_P3_M3
}

// ------------------------------------------------------------------------------------------------
// Stage 2: processBuffers_c, SangNom2.cpp:126-159.  In place, rows 1..bh-1, all stride_e columns.
// Row r needs the already-filtered row r-1, so one workgroup per (buffer, frame) walks the rows.  A thread owns
// NC consecutive columns: it keeps rows r-1, r+1, r+2 of them in registers (each pool row is read once, with one
// vector load per thread), publishes its 3-row sums to a double-buffered LDS line -- one barrier per row -- and
// needs only the three sums on either side of its columns back for the 7-tap box.
// ------------------------------------------------------------------------------------------------
constexpr int kSmoothThreads = 1024;

template <class T, int NC>
struct alignas(NC * sizeof(T)) SampleVec {
    T v[NC];
};

template <class T, int NC>
__global__ void __launch_bounds__(kSmoothThreads) k_smooth(PoolArgs pool, int slot0)
{
    using P = Px<T>;
    using W = typename P::W;
    using Vec = SampleVec<T, NC>;
    extern __shared__ __align__(16) unsigned char smem[];
    const int se = pool.stride_e;
    W* line0 = reinterpret_cast<W*>(smem);
    W* line1 = line0 + se;
    const int b = blockIdx.x;
    const int f = blockIdx.y;
    if (pool.guard && pool.guard[pool.guard_single ? 0 : f] == 0) return;
    const size_t bufsz = (size_t)se * (pool.bh + 1);
    T* buf = reinterpret_cast<T*>(pool.base + slot_of(pool, slot0, f) * pool.slot_bytes) + (size_t)b * bufsz;
    // the workgroup has ceil(se / NC) threads rounded up to whole waves; the idle lanes of the last wave compute on
    // column 0's data and store nothing, so every thread reaches every barrier
    const bool active = (int)threadIdx.x * NC < se;
    const int x0 = active ? threadIdx.x * NC : 0;

    auto load = [&](int row, W (&out)[NC]) {
        const Vec t = *reinterpret_cast<const Vec*>(buf + (size_t)row * se + x0);
#pragma unroll
        for (int k = 0; k < NC; ++k) out[k] = (W)t.v[k];
    };
    auto row_or_last = [&](int row) { return row <= pool.bh ? row : pool.bh; };  // past the end: loaded, never used
    W prev[NC], cur[NC], nxt[NC];
    load(0, prev);
    load(1, cur);
    load(row_or_last(2), nxt);
    const int rows = pool.rows > 0 && pool.rows < pool.bh ? pool.rows : pool.bh;  // rows 1 .. rows - 1 are smoothed
    for (int r = 1; r < rows; ++r) {
        W pre[NC];  // row r+2, fetched one row step ahead of its first use (unconditional: a branch around the load
        load(row_or_last(r + 2), pre);  // makes the compiler wait for it at once)
        W* line = (r & 1) ? line1 : line0;
        W X[NC + 6];  // the sums of columns x0-3 .. x0+NC+2, clamped to the pool row (SangNom2.cpp:144-150)
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            X[3 + k] = P::sum3(prev[k], cur[k], nxt[k]);
            if (active) line[x0 + k] = X[3 + k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            X[3 - k] = line[x0 - k < 0 ? 0 : x0 - k];
            X[NC + 2 + k] = line[x0 + NC - 1 + k > se - 1 ? se - 1 : x0 + NC - 1 + k];
        }
        Vec o;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            // left-to-right, SangNom2.cpp:152
            const W s = (((((X[k] + X[k + 1]) + X[k + 2]) + X[k + 3]) + X[k + 4]) + X[k + 5]) + X[k + 6];
            prev[k] = P::div16(s);
            o.v[k] = (T)prev[k];
            cur[k] = nxt[k];
            nxt[k] = pre[k];
        }
        if (active) *reinterpret_cast<Vec*>(buf + (size_t)r * se + x0) = o;
    }
}

// 8-bit pools: the same walk with two columns per register.  A thread owns 8 columns as four registers of two 16-bit sums
// each (a 7-tap sum of 3-row sums stays below 5 355, so plain 32-bit adds never carry from one half into the other);
// the odd-aligned pairs the box needs come from v_alignbit, the box slides (two instructions per further pair), the
// neighbours' sums arrive as two 8-byte LDS reads instead of six 4-byte ones.  45 vector instructions per 8 columns and
// row against 79 per 4: stage 2 of one 2160p pool 830 -> 454 us.
__global__ void __launch_bounds__(kSmoothThreads) k_smooth_u8x2(PoolArgs pool, int slot0)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int se = pool.stride_e;       // a multiple of 32
    const int nt = se >> 3;             // threads that own columns
    uint4* line0 = reinterpret_cast<uint4*>(smem);
    uint4* line1 = line0 + nt;
    const int b = blockIdx.x;
    const int f = blockIdx.y;
    if (pool.guard && pool.guard[pool.guard_single ? 0 : f] == 0) return;
    const size_t bufsz = (size_t)se * (pool.bh + 1);
    uint8_t* buf = pool.base + slot_of(pool, slot0, f) * pool.slot_bytes + (size_t)b * bufsz;
    const int tid = threadIdx.x;
    const bool active = tid < nt;
    const int t = active ? tid : 0;  // idle lanes of the last wave shadow thread 0 and store nothing
    const bool first = t == 0, last = t == nt - 1;
    const int tl = first ? 0 : t - 1, tr = last ? t : t + 1;

    struct Row {
        unsigned v[4];  // columns 8t + 2i | 8t + 2i + 1 << 16
    };
    auto unpack = [](uint2 q) {
        Row r;
        r.v[0] = __builtin_amdgcn_perm(0u, q.x, 0x0c010c00u);
        r.v[1] = __builtin_amdgcn_perm(0u, q.x, 0x0c030c02u);
        r.v[2] = __builtin_amdgcn_perm(0u, q.y, 0x0c010c00u);
        r.v[3] = __builtin_amdgcn_perm(0u, q.y, 0x0c030c02u);
        return r;
    };
    auto load = [&](int row) { return *reinterpret_cast<const uint2*>(buf + (size_t)row * se + 8 * t); };
    auto row_or_last = [&](int row) { return row <= pool.bh ? row : pool.bh; };  // past the end: loaded, never used
    Row prev = unpack(load(0)), cur = unpack(load(1)), nxt = unpack(load(row_or_last(2)));
    const int rows = pool.rows > 0 && pool.rows < pool.bh ? pool.rows : pool.bh;  // rows 1 .. rows - 1 are smoothed
    // rows r + 2 .. r + 1 + kAhead are in flight: with two waves per SIMD a row step is too short to cover the latency of
    // a pool row that comes from HBM or a far L2; the ring is indexed by the unrolled position only
    constexpr int kAhead = 4;
    uint2 ring[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) ring[u] = load(row_or_last(3 + u));
    auto row_step = [&](int r, uint2 pre) {  // pre = row r + 2
        uint4* line = (r & 1) ? line1 : line0;
        // E[2 + j] = sums of columns 8t + 2j | 8t + 2j + 1, j = -2 .. 5
        unsigned E[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) E[2 + i] = prev.v[i] + cur.v[i] + nxt.v[i];
        if (active) line[t] = make_uint4(E[2], E[3], E[4], E[5]);
        __syncthreads();
        const uint2 lf = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned*>(line + tl) + 2);
        const uint2 rt = *reinterpret_cast<const uint2*>(line + tr);
        // the pool row is clamped at both ends (SangNom2.cpp:144-150)
        const unsigned left_edge = (E[2] & 0xffffu) * 0x10001u, right_edge = (E[5] >> 16) * 0x10001u;
        E[0] = first ? left_edge : lf.x;
        E[1] = first ? left_edge : lf.y;
        E[6] = last ? right_edge : rt.x;
        E[7] = last ? right_edge : rt.y;
        // O[j] = sums of columns 8t + 2j - 3 | 8t + 2j - 2 (the pairs one column to the left), j = 0 .. 6
        unsigned O[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) O[j] = __builtin_amdgcn_alignbit(E[j + 1], E[j], 16);
        // output pair m (columns 8t + 2m | 8t + 2m + 1) = O[m] + E[1+m] + O[m+1] + E[2+m] + O[m+2] + E[3+m] + O[m+3]
        unsigned T = ((O[0] + E[1]) + (O[1] + E[2])) + ((O[2] + E[3]) + O[3]);
        Row o;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            o.v[m] = (T >> 4) & 0x00ff00ffu;  // (sum / 16) wraps to uint8_t, SangNom2.cpp:152; integer sums: any order
            if (m < 3) T = (T - O[m] - E[1 + m]) + (E[4 + m] + O[m + 4]);
        }
        uint2 q;
        q.x = __builtin_amdgcn_perm(o.v[1], o.v[0], 0x06040200u);
        q.y = __builtin_amdgcn_perm(o.v[3], o.v[2], 0x06040200u);
        if (active) *reinterpret_cast<uint2*>(buf + (size_t)r * se + 8 * t) = q;
        prev = o;
        cur = nxt;
        nxt = unpack(pre);
    };
    for (int r = 1; r < rows; r += kAhead) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (r + u < rows) {  // uniform
                const uint2 pre = ring[u];
                ring[u] = load(row_or_last(r + u + 2 + kAhead));
                row_step(r + u, pre);
            }
        }
    }
}

// 8-bit pools of 512 columns and more, second cut: no barrier per row.  The pool row is cut into strips of 60 lanes x 8
// columns, one wave each, exactly like the fused sweeps (sn_fused_v3_common.h): inside a wave the neighbours' sums come
// over DPP, and two ghost lanes on either inner side of a strip repeat the neighbouring wave's seam columns, good for
// K = 5 rows (the box spreads 3 columns a row over their 16), after which they take the seam lanes' state from an LDS
// mailbox -- one barrier every five rows instead of one per row, no LDS on the row's critical path.  Waves drift by
// up to four rows between barriers, and the rows are smoothed IN PLACE: a lane fetches pool row r + 2 + kAhead while it
// works on row r, i.e. before the wave that owns those columns (at most four rows ahead) can have overwritten it.
__global__ void __launch_bounds__(kSmoothThreads) k_smooth_u8_strips(PoolArgs pool, int slot0)
{
    using namespace v3c;
    extern __shared__ __align__(16) unsigned char smem[];
    const int se = pool.stride_e;
    const int nl = se >> 3;             // lanes that own columns
    const int nw = (int)blockDim.x >> 6;
    const int b = blockIdx.x;
    const int f = blockIdx.y;
    if (pool.guard && pool.guard[pool.guard_single ? 0 : f] == 0) return;
    const size_t bufsz = (size_t)se * (pool.bh + 1);
    uint8_t* buf = pool.base + slot_of(pool, slot0, f) * pool.slot_bytes + (size_t)b * bufsz;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int gl;
    bool ghost;
    if (wave == 0) {
        gl = lane;
        ghost = nw > 1 && lane >= 64 - GH;
    } else {
        gl = kFirst + kInner * (wave - 1) + (lane - GH);
        ghost = lane < GH || (lane >= 64 - GH && wave < nw - 1);
    }
    const bool live = gl < nl, real = live && !ghost;
    const int x0 = live ? gl * 8 : 0;  // dead lanes shadow column 0 and store nothing
    const unsigned first_mask = live && gl == 0 ? 0xffffffffu : 0u, last_mask = live && gl == nl - 1 ? 0xffffffffu : 0u;
    // mailbox: [copy][wave][side][slot][4 registers]
    unsigned* mb = reinterpret_cast<unsigned*>(smem);
    auto mb_at = [&](int copy, int w, int side, int slot) { return mb + ((((copy * nw + w) * 2 + side) * GH + slot) * 4); };
    const bool pub_right = lane >= 64 - 2 * GH && lane < 64 - GH && wave < nw - 1;  // feeds the next wave's left ghosts
    const bool pub_left = lane >= GH && lane < 2 * GH && wave > 0;                   // ... the previous wave's right ghosts
    const bool recv = ghost && live;
    const int slot = lane < GH ? lane : lane >= 64 - GH ? lane - (64 - GH) : pub_right ? lane - (64 - 2 * GH) : lane - GH;

    struct Row {
        unsigned v[4];  // columns x0 + 2i | x0 + 2i + 1 << 16
    };
    auto unpack = [](uint2 q) {
        Row r;
        r.v[0] = __builtin_amdgcn_perm(0u, q.x, 0x0c010c00u);
        r.v[1] = __builtin_amdgcn_perm(0u, q.x, 0x0c030c02u);
        r.v[2] = __builtin_amdgcn_perm(0u, q.y, 0x0c010c00u);
        r.v[3] = __builtin_amdgcn_perm(0u, q.y, 0x0c030c02u);
        return r;
    };
    auto load = [&](int row) { return *reinterpret_cast<const uint2*>(buf + (size_t)row * se + x0); };
    auto row_or_last = [&](int row) { return row <= pool.bh ? row : pool.bh; };  // past the end: loaded, never used
    constexpr int kAhead = K;  // >= the drift between two waves + 1
    Row prev = unpack(load(0)), cur = unpack(load(1)), nxt = unpack(load(row_or_last(2)));
    uint2 ring[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) ring[u] = load(row_or_last(3 + u));
    __syncthreads();  // nobody stores before everybody has fetched its first rows
    const int rows = pool.rows > 0 && pool.rows < pool.bh ? pool.rows : pool.bh;  // rows 1 .. rows - 1 are smoothed
    auto row_step = [&](int r, uint2 pre) {  // pre = row r + 2
        if (r > 1 && (r - 1) % K == 0) {  // the ghosts take over what the seam lanes held after row r - 1
            __syncthreads();
            if (recv) {
                const unsigned* from = mb_at((r / K) & 1, wave, lane < GH ? 0 : 1, slot);
#pragma unroll
                for (int i = 0; i < 4; ++i) prev.v[i] = from[i];
            }
        }
        unsigned E[8];  // E[2 + j] = sums of columns x0 + 2j | x0 + 2j + 1, j = -2 .. 5
#pragma unroll
        for (int i = 0; i < 4; ++i) E[2 + i] = prev.v[i] + cur.v[i] + nxt.v[i];
        // the neighbouring lanes' sums; the pool row is clamped at both ends (SangNom2.cpp:144-150).  Bitwise selects:
        // a DPP read under a lane mask would see the masked-off lanes as zero.
        const unsigned left_edge = (E[2] & 0xffffu) * 0x10001u, right_edge = (E[5] >> 16) * 0x10001u;
        E[0] = (first_mask & left_edge) | (~first_mask & dpp_from_left(E[4]));
        E[1] = (first_mask & left_edge) | (~first_mask & dpp_from_left(E[5]));
        E[6] = (last_mask & right_edge) | (~last_mask & dpp_from_right(E[2]));
        E[7] = (last_mask & right_edge) | (~last_mask & dpp_from_right(E[3]));
        unsigned O[7];  // O[j] = sums of columns x0 + 2j - 3 | x0 + 2j - 2
#pragma unroll
        for (int j = 0; j < 7; ++j) O[j] = __builtin_amdgcn_alignbit(E[j + 1], E[j], 16);
        unsigned T = ((O[0] + E[1]) + (O[1] + E[2])) + ((O[2] + E[3]) + O[3]);
        Row o;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            o.v[m] = (T >> 4) & 0x00ff00ffu;  // (sum / 16) wraps to uint8_t, SangNom2.cpp:152; integer sums: any order
            if (m < 3) T = (T - O[m] - E[1 + m]) + (E[4 + m] + O[m + 4]);
        }
        if (real) {
            uint2 q;
            q.x = __builtin_amdgcn_perm(o.v[1], o.v[0], 0x06040200u);
            q.y = __builtin_amdgcn_perm(o.v[3], o.v[2], 0x06040200u);
            *reinterpret_cast<uint2*>(buf + (size_t)r * se + x0) = q;
        }
        prev = o;
        cur = nxt;
        nxt = unpack(pre);
        if (r % K == 0 && r < rows - 1 && (pub_right || pub_left)) {
            unsigned* to = pub_right ? mb_at(((r + 1) / K) & 1, wave + 1, 0, slot) : mb_at(((r + 1) / K) & 1, wave - 1, 1, slot);
#pragma unroll
            for (int i = 0; i < 4; ++i) to[i] = prev.v[i];
        }
    };
    for (int r = 1; r < rows; r += kAhead) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (r + u < rows) {  // uniform
                const uint2 pre = ring[u];
                ring[u] = load(row_or_last(r + u + 2 + kAhead));
                row_step(r + u, pre);
            }
        }
    }
}

// ---- chains over several workgroups per buffer (see k_smooth_u8_chain): the pieces the three sample types share ----
constexpr int kChainSpinLimit = 1 << 20;  // x (s_sleep 8 + a load from memory): two seconds and more

// Release side of the hand-off between workgroups: EVERY wave drains its own sc1 stores (s_waitcnt vmcnt(0): a store
// counts down when memory has acknowledged it at the scope of its sc bits) before the barrier that precedes the
// publish of the round counter.  The barrier alone does not do it: __syncthreads() is a workgroup-scope fence, which on
// gfx950 outside tgsplit mode waits for LDS traffic only, and the compiler's own s_waitcnt before s_barrier covered the
// stores in the 8-bit kernel by accident and left vmcnt(1..2) in the 16-bit / float ones (round-3 advisor, disassembly).
// Data and flag travel to different L2 channels, so without the drain a consumer could see the counter before the rows.
// tests/test_capi_cpu.py checks the disassembly for this instruction in front of every s_barrier of the *_chain<true> loops.
__device__ __forceinline__ void chain_release_barrier()
{
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0), expcnt and lgkmcnt left alone (gfx9 encoding: vmcnt = bits 3:0 and 15:14)
    __syncthreads();
}

// The round counters of a buffer's workgroups.  enter(), right after the barrier that opens round `round`: the workgroup
// publishes "rounds < round are complete" (chain_release_barrier() drained every wave's stores), and the waves of its first slot make sure
// the workgroup before it has completed round `round - 1 - slack` -- with the value they fetched while the round before
// ran if that is enough (in step it is: the schedule starts a workgroup's slots `slack` rounds late), polling otherwise,
// for two seconds at most: then the workgroup gives up waiting for good and raises the launch's fault word, which the guarded
// launches that follow every chain over several workgroups read: stage 1 again and the chain on ONE workgroup per buffer
// (nothing to wait for there), so the frames are right whatever happened (sn_api.hip, run_chain).
struct ChainSync {
    unsigned* mine;
    const unsigned* his;
    unsigned* status;
    int slack;
    unsigned seen;
    bool gave_up;
    __device__ __forceinline__ void init(const ChainArgs& ch, int b, int grp)
    {
        mine = ch.flags + (b * ch.groups + grp) * 32;  // a cache line of 128 bytes each
        his = ch.flags + (b * ch.groups + (grp + ch.groups - 1) % ch.groups) * 32;
        status = ch.status;
        slack = ch.slack;
        seen = 0;
        // (a fault word that is already up -- sn_debug_raise_chain_fault -- makes every wave skip its waits from the start: the
        // launch then produces what a launch that timed out produces, rows taken before they were written)
        gave_up = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    }
    __device__ __forceinline__ void enter(int round, bool first_slot, int tid)
    {
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (tid == 0) __hip_atomic_store(mine, (unsigned)round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int need = round - slack;
        if (first_slot && !gave_up) {
            if (need > 0 && (int)seen < need) {
                int spins = 0;
                while ((int)(seen = __hip_atomic_load(his, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
                    if (++spins > kChainSpinLimit) {
                        gave_up = true;
                        __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            seen = __hip_atomic_load(his, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next round
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
    }
};

// History-carrying 8-bit clips (pool cells a pass does not write keep what the pass before it left): the
// passes of a stream form a chain, but pass j + 1 only needs rows r + 1, r + 2 of what pass j smoothed, so one
// persistent workgroup per buffer keeps `lanes` passes in flight, each on the nw waves of a strip set
// (k_smooth_u8_strips), each kLag blocks of K rows behind the one before it.  Every pass has a pool slot of its own
// (ChainArgs); a lane reads a raw row from its own slot where its pass's k_prepare wrote it and from the slot of the
// pass before it everywhere else.  All waves meet at one barrier per block -- the one the strips need anyway for their
// ghost lanes -- which is also what orders pass j's stores before pass j + 1's loads (same workgroup, same CU), so
// nothing ever waits on another workgroup.  A pass: one round that fetches its first rows, then ceil((bh - 1) / K)
// blocks; block t of a pass fetches up to row K (t + 2) + 2 of the pass before it, which finished that block one
// round earlier when the lag is three rounds.
constexpr int kChainLag = 3;

// Round 3: TWO passes per wave.  A 32-bit register holds column x of one pass in its low half and column x of the pass
// that follows it in the chain (three blocks behind) in its high half -- the packing of the fused 8-bit sweeps, with two
// passes where those have two strips -- so one instruction stream smooths a row of each: eight packed registers per lane,
// three-row sums, DPP for the neighbours, a sliding box, `(sum >> 4) & 0x00ff00ff`.  76 instructions per lane and round
// row for sixteen columns where the one-pass body took 75 for eight; the workgroup is bound by what its CU issues, so the
// chain runs twice as fast.  The halves are at different rows of different pool slots: each has its own pointers, fetch
// ring, fresh-row count, and its loads, stores, ghost refresh and first-rows round are masked per half.  The mailbox is
// indexed by the parity of the ROUND (both halves publish and receive at the same barrier), not of a pass's block.
#ifndef SN_CHAIN8_THREADS
#define SN_CHAIN8_THREADS 1024  // sixteen waves, four per SIMD: the two-pass body fits 128 registers (512 threads measured slower)
#endif
constexpr int kChain8Threads = SN_CHAIN8_THREADS;

// Round 3: SEVERAL workgroups per buffer (GROUPED).  One workgroup is bound by what its CU issues (profiles/r3_chain.md:
// one VALU instruction per 4.7 cycles and SIMD where the instruction class's floor is 4.06), and a pass can only start
// kChainLag rounds after the one before it -- so the stream's rate is 1 / (kChainLag x the time of a round), and a round
// is as long as the waves of a SIMD are many.  With G workgroups a buffer's passes in flight are spread over G CUs (slots
// g * lanes .. g * lanes + lanes - 1 in workgroup g, each with 1 / G of the sixteen waves): the rounds get shorter, the
// passes in flight stay sixteen.  A slot's predecessor is in the same workgroup (the barrier orders the two, as before)
// except for a workgroup's first slot, whose predecessor is the last slot of the workgroup before it (around the ring:
// the first workgroup follows the last one's previous cycle).  That one hand-off goes through memory:
//   * every pool access of the kernel is an agent-scope relaxed atomic (global_load / global_store ... sc1: coherent
//     across the XCDs without cache maintenance.  Agent-scope FENCES cost 7 - 20 us per round here and the more the more
//     workgroups issue them, the sc1 accesses 1.3 us when the reader sits waiting for them:
//     tools/experiments/ubench_xwg_sync.hip, profiles/r3_chain.md);
//   * a workgroup publishes "rounds completed" after chain_release_barrier() (every wave drains its stores, then the barrier), and the waves of its
//     first slot start round R only when the workgroup before them has completed round R - 1 - slack.  The schedule
//     starts workgroup g's slots g * slack rounds late, so in step that counter is `slack` rounds old news: the waves look
//     at the value they fetched while the round before ran (a fresh poll is a trip to memory on the round's critical path,
//     which cost 7 - 10 % of the rate with four and eight workgroups) and poll only if that is not enough;
//   * a wait is bounded: after two seconds and more a workgroup gives up waiting for good and raises ChainArgs::status
//     (a workgroup that was never scheduled -- another context's sweep holding every CU slot -- must not hang the GPU);
//     the launch is then redone on one workgroup per buffer by the guarded launches queued behind it (ChainArgs::only_if);
//   * a pass's first rows come from memory as well: they are fetched at the start of the pass's first round and taken
//     into the registers at its end (LATE_PRIME), not waited for on the spot.
// 720x480 YUV420P8, 512 frames per launch: 23.1 k frames/s with one workgroup per buffer, 28.4 k with two, 31.5 k with
// four (36 CUs), 32.9 k with eight (the default: kChainDefaultGroups, sn_api.hip).  Tried and dropped: fetching a round's rows all at its start into a
// second ring (slower), waiting only for the stores of a round's first two rows at the barrier (no change).
#ifdef SN_CHAIN_TIMING
// -DSN_CHAIN_TIMING (tools/chain_timing.py): shader-clock cycles the waves that had rows spent, summed over them, in
// [0] the barrier, [1] the hand-off wait, [2] the schedule and ghost refresh, [3] the rows, [4] the publish; [5] = wave-rounds counted
__device__ unsigned long long sn_chain_cycles[6];
#define SN_TICK(k)                                                     \
    do {                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        acc_[k] += now_ - last_;                                       \
        last_ = now_;                                                  \
    } while (0)
#else
#define SN_TICK(k) do { } while (0)
#endif
// THREADS: the launch bound.  The workgroups of a spread chain and short chains (a frame's two or three passes) have eight
// waves at most and 256 registers each; a long chain on one workgroup per buffer has sixteen waves and 128.
template <bool GROUPED, int THREADS>
__global__ void __launch_bounds__(THREADS) k_smooth_u8_chain(PoolArgs pool, ChainArgs ch, int nw, int lanes, int pass_rounds,
                                                                    int cycle)
{
    using namespace v3c;
    extern __shared__ __align__(16) unsigned char smem[];
    if (ch.only_if && *ch.only_if == 0) return;  // the guarded redo of a launch that did not time out: nothing to do
    constexpr unsigned kLo = 0x0000ffffu, kHi = 0xffff0000u;
    const int se = pool.stride_e;
    const int nl = se >> 3;  // lanes that own columns
    constexpr bool LATE_PRIME = GROUPED;  // a pass's first rows are taken in at the END of its first round
    const int groups = GROUPED ? ch.groups : 1, slack = GROUPED ? ch.slack : 0;
    const int b = GROUPED ? (int)blockIdx.x / groups : (int)blockIdx.x, grp = GROUPED ? (int)blockIdx.x % groups : 0;
    const size_t bufsz = (size_t)se * (pool.bh + 1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, and the compiler must know it: everything
    const int pair = wid / nw, wave = wid % nw;                // the schedule derives from it then lives in scalar registers
    // (pair: which pair of the passes in flight; wave: which strip of it)
    int gl;
    bool ghost;
    if (wave == 0) {
        gl = lane;
        ghost = nw > 1 && lane >= 64 - GH;
    } else {
        gl = kFirst + kInner * (wave - 1) + (lane - GH);
        ghost = lane < GH || (lane >= 64 - GH && wave < nw - 1);
    }
    const bool live = gl < nl, real = live && !ghost;
    const int x0 = live ? gl * 8 : 0;  // dead lanes shadow column 0 and store nothing
    const unsigned last_mask = live && gl == nl - 1 ? 0xffffffffu : 0u;
    // mailbox: [pair in flight][copy][wave][side][slot][8 packed registers]
    unsigned* mb = reinterpret_cast<unsigned*>(smem) + (size_t)pair * (2 * nw * 2 * GH * 8);
    auto mb_at = [&](int copy, int w, int side, int slot) { return mb + ((((copy * nw + w) * 2 + side) * GH + slot) * 8); };
    const bool pub_right = lane >= 64 - 2 * GH && lane < 64 - GH && wave < nw - 1;
    const bool pub_left = lane >= GH && lane < 2 * GH && wave > 0;
    const bool recv = ghost && live;
    const int slot = lane < GH ? lane : lane >= 64 - GH ? lane - (64 - GH) : pub_right ? lane - (64 - 2 * GH) : lane - GH;

    struct Row {
        unsigned v[8];  // column x0 + c of the low-half pass | of the high-half pass << 16
    };
    auto unpack = [](uint2 lo, uint2 hi) {  // byte k of the lo words -> bits 0..7, of the hi words -> bits 16..23
        Row r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r.v[k] = __builtin_amdgcn_perm(hi.x, lo.x, 0x0c040c00u + (unsigned)k * 0x00010001u);
            r.v[4 + k] = __builtin_amdgcn_perm(hi.y, lo.y, 0x0c040c00u + (unsigned)k * 0x00010001u);
        }
        return r;
    };
    const int rows = ch.rows > 0 && ch.rows < pool.bh ? ch.rows : pool.bh;  // rows 1 .. rows - 1 are smoothed
    const int origin = ch.origin + (int)blockIdx.y * ch.chain_step;           // blockIdx.y: one of several independent chains
    const int slots = lanes * groups;  // passes in flight per buffer; slot g starts (g * kChainLag + its workgroup * slack) rounds in
    const int last_slot = (ch.npass - 1) % slots;  // the chain is over when its last pass is: no rounds for slots nothing fills
    const int total = ((ch.npass - 1) / slots) * cycle + last_slot * kChainLag + (last_slot / lanes) * slack + pass_rounds;

    // state of the two passes this wave is working on (lo: low halves, pass slot 2 * pair; hi: high halves)
    // Addresses: ONE buffer resource over the whole ring; a pass's slot and cost buffer and the pool row are wave-uniform
    // and travel in the scalar offset, a lane adds 32 bits of its own -- its columns, plus which of the two slots it reads
    // (its own where the pass's k_prepare wrote, the one of the pass before it elsewhere).  Per load one select, per store
    // nothing, where 64-bit pointers per lane took seven and four instructions; lanes that store nothing carry an offset
    // the range check rejects.  (The ring stays below 4 GB for that: ensure_chain, sn_api.hip.)
    constexpr unsigned kNoAccess = 0xffffffffu;
    constexpr int kAux = GROUPED ? 16 : 0;  // sc1: coherent at agent scope, as the atomics of the 16-bit and float chains
    const unsigned long long ring_bytes = (unsigned long long)pool.slot_bytes * (unsigned long long)pool.slot_mod;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pool.base, 0, ring_bytes < 0xfffffffeull ? (int)(unsigned)ring_bytes : (int)0xfffffffeu, 0x00020000);
    struct Half {
        unsigned vo_fresh;      // this lane's offset for rows 1 .. nr: in the pass's own slot where its k_prepare wrote ...
        unsigned vo_stale;      // ... and for every other row: in the slot of the pass before it
        unsigned vo_out;        // where it stores (kNoAccess: ghost and dead lanes)
        int nr;                 // rows 1 .. nr of columns < w were written by the pass's k_prepare
        int t;                  // block of K rows the pass is at in this round
        bool run, fetch;        // the pass has rows in this round / is in its first round
        int phase, turn;        // the slot's clock: rounds since it last took a pass (< 0: not started yet), passes taken so far
        uint2 ring[K];          // rows fetched ahead
        uint2 first[3];         // GROUPED: rows 0 .. 2, from the pass's first round's start to its end
    };
    constexpr int kAhead = K;
    Half lo{kNoAccess, kNoAccess, kNoAccess, 0, 0, false, false, 0, 0, {}, {}}, hi{kNoAccess, kNoAccess, kNoAccess, 0, 0, false, false, 0, 0, {}, {}};
    Row prev{}, cur{}, nxt{};
    auto load = [&](const Half& H, int row) {
        row = row < 0 ? 0 : row <= pool.bh ? row : pool.bh;  // past the end: loaded, never used
        const bool fresh = row >= 1 && row <= H.nr;  // (uniform)
        const unsigned of = H.vo_fresh, os = H.vo_stale;
        const auto q = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(fresh ? of : os), row * se, kAux);
        return make_uint2((unsigned)q[0], (unsigned)q[1]);
    };
    auto store = [&](const Half& H, int row, uint2 q) {
        u32x2 v;
        v[0] = q.x;
        v[1] = q.y;
        __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)H.vo_out, row * se, kAux);
    };
    // a pass's rows 0 .. 2 into its half of the registers
    auto prime_from = [&](uint2 q0, uint2 q1, uint2 q2, bool high) {
        const uint2 z = make_uint2(0u, 0u);
        const Row p0 = high ? unpack(z, q0) : unpack(q0, z), p1 = high ? unpack(z, q1) : unpack(q1, z), p2 = high ? unpack(z, q2) : unpack(q2, z);
        const unsigned keep = high ? kLo : kHi;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            prev.v[c] = (prev.v[c] & keep) | p0.v[c];
            cur.v[c] = (cur.v[c] & keep) | p1.v[c];
            nxt.v[c] = (nxt.v[c] & keep) | p2.v[c];
        }
    };
    auto prime = [&](const Half& H, bool high) { prime_from(H.first[0], H.first[1], H.first[2], high); };
    // where a pass slot stands in this round; fetches the pass's first rows (one workgroup per buffer: into its half of
    // the registers at once.  GROUPED: the loads come from memory, not from the L2, and a wave that sat through them would
    // hold up its workgroup at the next barrier and the workgroups behind it: they are taken in when the round is over)
    // (The slot's clock is kept by counting: phase = rel % cycle and turn = rel / cycle for rel = round - the slot's start.
    // Dividing every round cost a quarter of a round's cycles in scalar instructions, profiles/r3_chain.md.)
    auto start_clock = [&](Half& H, int ps) {
        H.phase = -((grp * lanes + ps) * kChainLag + grp * slack) - 1;
        H.turn = 0;
    };
    start_clock(lo, 2 * pair);
    start_clock(hi, 2 * pair + 1);
    auto schedule = [&](Half& H, int ps, bool high, int round) {
        const int g = grp * lanes + ps;  // the slot among all of the buffer's
        if (++H.phase == cycle) {
            H.phase = 0;
            ++H.turn;
        }
        const int j = H.phase >= 0 ? H.turn * slots + g : ch.npass;
        H.t = H.phase >= 0 ? H.phase - 1 : 0;
        const bool active = j < ch.npass && H.t < pass_rounds - 1;
        H.run = active && H.t >= 0;
        H.fetch = active && H.t < 0;
        if (H.fetch) {
            const int k = j % ch.pn;
            const int64_t s_own = (origin + 1 + j) % pool.slot_mod, s_before = (origin + j) % pool.slot_mod;
            const unsigned own_off = (unsigned)(s_own * pool.slot_bytes + (int64_t)b * (int64_t)bufsz);        // (uniform; the ring is
            const unsigned before_off = (unsigned)(s_before * pool.slot_bytes + (int64_t)b * (int64_t)bufsz);  // smaller than 4 GB)
            H.vo_stale = before_off + x0;
            H.vo_fresh = (x0 < ch.w[k] ? own_off : before_off) + x0;
            H.vo_out = real ? own_off + x0 : kNoAccess;
            H.nr = ch.nr[k];
            if constexpr (LATE_PRIME) {  // only issued here: prime() takes them in when the round is over
#pragma unroll
                for (int u = 0; u < 3; ++u) H.first[u] = load(H, u);
#pragma unroll
                for (int u = 0; u < kAhead; ++u) H.ring[u] = load(H, 3 + u);
            } else {
                const uint2 q0 = load(H, 0), q1 = load(H, 1), q2 = load(H, 2);
                prime_from(q0, q1, q2, high);
#pragma unroll
                for (int u = 0; u < kAhead; ++u) H.ring[u] = load(H, 3 + u);
            }
        }
    };

    ChainSync sync{};
    if constexpr (GROUPED) sync.init(ch, b, grp);
#ifdef SN_CHAIN_TIMING
    unsigned long long acc_[6] = {0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime(), keep_[2] = {0, 0};
#endif
    for (int round = 0; round < total; ++round) {
#ifdef SN_CHAIN_TIMING
        last_ = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (GROUPED) chain_release_barrier();  // this wave's stores have landed, then everybody's
        else __syncthreads();  // (one workgroup: same CU, the barrier orders LDS and the L1-coherent rows)
        SN_TICK(0);
        if constexpr (GROUPED) sync.enter(round, pair == 0, tid);
        SN_TICK(1);
        schedule(lo, 2 * pair, false, round);
        schedule(hi, 2 * pair + 1, true, round);
        if (!lo.run && !hi.run) {
            if constexpr (LATE_PRIME) {
                if (lo.fetch) prime(lo, false);
                if (hi.fetch) prime(hi, true);
            }
#ifdef SN_CHAIN_TIMING
            acc_[0] = keep_[0];  // rounds without rows are not counted
            acc_[1] = keep_[1];
#endif
            continue;
        }
#ifdef SN_CHAIN_TIMING
        keep_[0] = acc_[0];
        keep_[1] = acc_[1];
        acc_[5] += 1;
#endif
        const int copy = round & 1;
        const unsigned refresh = (lo.run && lo.t > 0 ? kLo : 0u) | (hi.run && hi.t > 0 ? kHi : 0u);
        if (refresh && recv) {  // the ghosts take over what the seam lanes held after the block before
            const unsigned* from = mb_at(copy, wave, lane < GH ? 0 : 1, slot);
#pragma unroll
            for (int c = 0; c < 8; ++c) prev.v[c] = (from[c] & refresh) | (prev.v[c] & ~refresh);
        }
        // one row of each pass; BOTH: both halves have a row here (known at compile time in the steady state)
        auto row_body = [&](int u, bool v0, bool v1, auto both_tag) __attribute__((always_inline)) {
            constexpr bool BOTH = decltype(both_tag)::value;
            const int r0 = K * lo.t + 1 + u, r1 = K * hi.t + 1 + u;
            const uint2 pre0 = lo.ring[u], pre1 = hi.ring[u];
            if (BOTH || v0) lo.ring[u] = load(lo, r0 + 2 + kAhead);
            if (BOTH || v1) hi.ring[u] = load(hi, r1 + 2 + kAhead);
            unsigned E[8];  // three-row sums of the lane's eight columns, both passes
#pragma unroll
            for (int c = 0; c < 8; ++c) E[c] = prev.v[c] + cur.v[c] + nxt.v[c];
            // the neighbouring lanes' sums; the pool row is clamped at both ends (SangNom2.cpp:144-150): column 0 is
            // lane 0 of the first strip, whose DPP move keeps its `old` operand (lane 0 of the other strips is the
            // outermost ghost lane); the last column takes a bitwise select (a DPP read must not run under a lane mask)
            unsigned X[14];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                X[k] = dpp_from_left_or(E[0], E[5 + k]);
                X[11 + k] = (last_mask & E[7]) | (~last_mask & dpp_from_right(E[k]));
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) X[3 + c] = E[c];
            unsigned T = ((X[0] + X[1]) + (X[2] + X[3])) + ((X[4] + X[5]) + X[6]);
            Row o;
            if constexpr (THREADS <= 512) {
                // two windows slide side by side (columns 0 .. 3 and 4 .. 7): with one or two waves on a SIMD a row is as long
                // as its longest chain of dependent instructions, and one window sliding over eight columns is fourteen deep
                unsigned T4 = ((X[4] + X[5]) + (X[6] + X[7])) + ((X[8] + X[9]) + X[10]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    o.v[c] = (T >> 4) & 0x00ff00ffu;  // (sum / 16) wraps to uint8_t, SangNom2.cpp:152; integer sums: any order
                    o.v[4 + c] = (T4 >> 4) & 0x00ff00ffu;
                    if (c < 3) {
                        T = (T - X[c]) + X[c + 7];
                        T4 = (T4 - X[4 + c]) + X[c + 11];
                    }
                }
            } else {  // sixteen waves: bound by what the CU issues, three instructions fewer
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    o.v[c] = (T >> 4) & 0x00ff00ffu;
                    if (c < 7) T = (T - X[c]) + X[c + 7];
                }
            }
            {   // o.v[c] = low-pass byte | high-pass byte << 16  ->  eight bytes per pass
                const unsigned t01 = __builtin_amdgcn_perm(o.v[1], o.v[0], 0x06020400u), t23 = __builtin_amdgcn_perm(o.v[3], o.v[2], 0x06020400u);
                const unsigned t45 = __builtin_amdgcn_perm(o.v[5], o.v[4], 0x06020400u), t67 = __builtin_amdgcn_perm(o.v[7], o.v[6], 0x06020400u);
                uint2 qa, qb;
                qa.x = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
                qa.y = __builtin_amdgcn_perm(t67, t45, 0x05040100u);
                qb.x = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
                qb.y = __builtin_amdgcn_perm(t67, t45, 0x07060302u);
                if (BOTH || v0) store(lo, r0, qa);  // (lanes that keep nothing: dropped by the range check, no branch)
                if (BOTH || v1) store(hi, r1, qb);
            }
            if (BOTH || (v0 && v1)) {
                prev = o;
                cur = nxt;
                nxt = unpack(pre0, pre1);
            } else {
                // a half that has no row here -- its pass has just fetched its first rows, or is over -- keeps its
                // registers (carries never cross the halves: every quantity stays below 2^16 whatever a half holds)
                const unsigned m = v0 ? kLo : kHi;
                const uint2 z = make_uint2(0u, 0u);
                // (LATE_PRIME: the ring of a half in its first round is still on its way)
                const Row nn = LATE_PRIME ? unpack(v0 ? pre0 : z, v1 ? pre1 : z) : unpack(pre0, pre1);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    prev.v[c] = (o.v[c] & m) | (prev.v[c] & ~m);
                    cur.v[c] = (nxt.v[c] & m) | (cur.v[c] & ~m);
                    nxt.v[c] = (nn.v[c] & m) | (nxt.v[c] & ~m);
                }
            }
        };
        SN_TICK(2);
        if (lo.run && hi.run && K * lo.t + K < rows && K * hi.t + K < rows) {  // the steady state: no condition per row
#pragma unroll
            for (int u = 0; u < K; ++u) row_body(u, true, true, std::integral_constant<bool, true>{});
        } else {
#pragma unroll
            for (int u = 0; u < K; ++u) {
                const bool v0 = lo.run && K * lo.t + 1 + u < rows, v1 = hi.run && K * hi.t + 1 + u < rows;
                if (v0 || v1) row_body(u, v0, v1, std::integral_constant<bool, false>{});  // uniform
            }
            if constexpr (LATE_PRIME) {
                if (lo.fetch) prime(lo, false);
                if (hi.fetch) prime(hi, true);
            }
        }
        SN_TICK(3);
        const bool more0 = lo.run && K * (lo.t + 1) < rows - 1, more1 = hi.run && K * (hi.t + 1) < rows - 1;
        if ((more0 || more1) && (pub_right || pub_left)) {
            unsigned* to = pub_right ? mb_at((round + 1) & 1, wave + 1, 0, slot) : mb_at((round + 1) & 1, wave - 1, 1, slot);
#pragma unroll
            for (int c = 0; c < 8; ++c) to[c] = prev.v[c];
        }
        SN_TICK(4);
    }
#ifdef SN_CHAIN_TIMING
    if (lane == 0)
        for (int k = 0; k < 6; ++k) atomicAdd(&sn_chain_cycles[k], acc_[k]);
#endif
}

#ifdef SN_CHAIN_TIMING
extern "C" __attribute__((visibility("default"))) int sn_debug_chain_cycles(unsigned long long out[6], int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sn_chain_cycles), sizeof(unsigned long long) * 6) != hipSuccess) return 1;
    if (reset) {
        const unsigned long long z[6] = {0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sn_chain_cycles), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// The chain for 9..16-bit samples: the same schedule around the row body of k_smooth_u16_strips.  GROUPED: over several
// workgroups per buffer as k_smooth_u8_chain<true> -- with one pass per wave a workgroup of sixteen waves only holds
// 16 / nw passes, so here the workgroups ADD slots (up to the sixteen the lag of three rounds can use) before they
// shorten the rounds.
template <bool GROUPED>
__global__ void __launch_bounds__(kSmoothThreads) k_smooth_u16_chain(PoolArgs pool, ChainArgs ch, int nw, int lanes, int pass_rounds,
                                                                    int cycle)
{
    using namespace v3c;
    extern __shared__ __align__(16) unsigned char smem[];
    if (ch.only_if && *ch.only_if == 0) return;  // the guarded redo of a launch that did not time out: nothing to do
    const int se = pool.stride_e;
    const int nl = se >> 3;  // lanes that own columns
    const int groups = GROUPED ? ch.groups : 1, slack = GROUPED ? ch.slack : 0;
    const int b = GROUPED ? (int)blockIdx.x / groups : (int)blockIdx.x, grp = GROUPED ? (int)blockIdx.x % groups : 0;
    const size_t bufsz = (size_t)se * (pool.bh + 1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int ps = (tid >> 6) / nw, wave = (tid >> 6) % nw;  // which pass of the `lanes` in flight, which strip of it
    int gl;
    bool ghost;
    if (wave == 0) {
        gl = lane;
        ghost = nw > 1 && lane >= 64 - GH;
    } else {
        gl = kFirst + kInner * (wave - 1) + (lane - GH);
        ghost = lane < GH || (lane >= 64 - GH && wave < nw - 1);
    }
    const bool live = gl < nl, real = live && !ghost;
    const int x0 = live ? gl * 8 : 0;  // dead lanes shadow column 0 and store nothing
    const unsigned first_mask = live && gl == 0 ? 0xffffffffu : 0u, last_mask = live && gl == nl - 1 ? 0xffffffffu : 0u;
    // mailbox: [pass in flight][copy][wave][side][slot][8 registers]
    unsigned* mb = reinterpret_cast<unsigned*>(smem) + (size_t)ps * (2 * nw * 2 * GH * 8);
    auto mb_at = [&](int copy, int w, int side, int slot) { return mb + ((((copy * nw + w) * 2 + side) * GH + slot) * 8); };
    const bool pub_right = lane >= 64 - 2 * GH && lane < 64 - GH && wave < nw - 1;
    const bool pub_left = lane >= GH && lane < 2 * GH && wave > 0;
    const bool recv = ghost && live;
    const int slot = lane < GH ? lane : lane >= 64 - GH ? lane - (64 - GH) : pub_right ? lane - (64 - 2 * GH) : lane - GH;

    struct Row {
        unsigned v[8];
    };
    auto unpack = [](uint4 q) {
        Row r;
        r.v[0] = q.x & 0xffffu; r.v[1] = q.x >> 16;
        r.v[2] = q.y & 0xffffu; r.v[3] = q.y >> 16;
        r.v[4] = q.z & 0xffffu; r.v[5] = q.z >> 16;
        r.v[6] = q.w & 0xffffu; r.v[7] = q.w >> 16;
        return r;
    };
    const int rows = ch.rows > 0 && ch.rows < pool.bh ? ch.rows : pool.bh;  // rows 1 .. rows - 1 are smoothed
    const int origin = ch.origin + (int)blockIdx.y * ch.chain_step;           // blockIdx.y: one of several independent chains
    const int slots = lanes * groups;  // passes in flight per buffer
    const int last_slot = (ch.npass - 1) % slots;  // the chain is over when its last pass is: no rounds for slots nothing fills
    const int total = ((ch.npass - 1) / slots) * cycle + last_slot * kChainLag + (last_slot / lanes) * slack + pass_rounds;

    // state of the pass this wave is working on; addresses as in k_smooth_u8_chain: one buffer resource over the ring, the
    // slot, cost buffer and pool row in the scalar offset, 32 bits per lane (a row is ONE sixteen-byte access, coherent or not)
    constexpr unsigned kNoAccess = 0xffffffffu;
    constexpr int kAux = GROUPED ? 16 : 0;  // sc1
    const unsigned long long ring_bytes = (unsigned long long)pool.slot_bytes * (unsigned long long)pool.slot_mod;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pool.base, 0, ring_bytes < 0xfffffffeull ? (int)(unsigned)ring_bytes : (int)0xfffffffeu, 0x00020000);
    unsigned vo_fresh = kNoAccess;  // this lane's offset for rows 1 .. nr: in the pass's own slot where its k_prepare wrote ...
    unsigned vo_stale = kNoAccess;  // ... and for every other row: in the slot of the pass before it
    unsigned vo_out = kNoAccess;    // where it stores (kNoAccess: ghost and dead lanes)
    int nr = 0;
    Row prev{}, cur{}, nxt{};
    constexpr int kAhead = K;
    uint4 ring[kAhead] = {};
    auto load = [&](int row) {
        row = row <= pool.bh ? row : pool.bh;  // past the end: loaded, never used
        const bool fresh = row >= 1 && row <= nr;  // (uniform)
        const unsigned of = vo_fresh, os = vo_stale;  // (by value: a conditional between the two VARIABLES would take their addresses)
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(fresh ? of : os), row * se * 2, kAux);
        return make_uint4(q[0], q[1], q[2], q[3]);
    };

    ChainSync sync{};
    if constexpr (GROUPED) sync.init(ch, b, grp);
    const int gslot = grp * lanes + ps;  // the slot among all of the buffer's
    for (int round = 0; round < total; ++round) {
        if constexpr (GROUPED) chain_release_barrier();
        else __syncthreads();
        if constexpr (GROUPED) sync.enter(round, ps == 0, tid);
        const int rel = round - (gslot * kChainLag + grp * slack);
        if (rel < 0) continue;
        const int j = (rel / cycle) * slots + gslot, t = rel % cycle - 1;
        if (j >= ch.npass || t >= pass_rounds - 1) continue;
        if (t < 0) {  // the pass's first rows
            const int k = j % ch.pn;
            const int64_t s_own = (origin + 1 + j) % pool.slot_mod, s_before = (origin + j) % pool.slot_mod;
            const unsigned own_off = (unsigned)(s_own * pool.slot_bytes + (int64_t)b * (int64_t)bufsz * 2);        // (uniform; the ring is
            const unsigned before_off = (unsigned)(s_before * pool.slot_bytes + (int64_t)b * (int64_t)bufsz * 2);  // smaller than 4 GB)
            vo_stale = before_off + x0 * 2;
            vo_fresh = (x0 < ch.w[k] ? own_off : before_off) + x0 * 2;
            vo_out = real ? own_off + x0 * 2 : kNoAccess;
            nr = ch.nr[k];
            prev = unpack(load(0));
            cur = unpack(load(1));
            nxt = unpack(load(2));
#pragma unroll
            for (int u = 0; u < kAhead; ++u) ring[u] = load(3 + u);
            continue;
        }
        if (t > 0 && recv) {  // the ghosts take over what the seam lanes held after the block before
            const uint4* from = reinterpret_cast<const uint4*>(mb_at(t & 1, wave, lane < GH ? 0 : 1, slot));
            const uint4 lo = from[0], hi = from[1];
            prev.v[0] = lo.x; prev.v[1] = lo.y; prev.v[2] = lo.z; prev.v[3] = lo.w;
            prev.v[4] = hi.x; prev.v[5] = hi.y; prev.v[6] = hi.z; prev.v[7] = hi.w;
        }
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int r = K * t + 1 + u;
            if (r < rows) {  // uniform
                const uint4 pre = ring[u];
                ring[u] = load(r + 2 + kAhead);
                unsigned X[14];  // as k_smooth_u16_strips
#pragma unroll
                for (int i = 0; i < 8; ++i) X[3 + i] = prev.v[i] + cur.v[i] + nxt.v[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    X[i] = (first_mask & X[3]) | (~first_mask & dpp_from_left(X[8 + i]));
                    X[11 + i] = (last_mask & X[10]) | (~last_mask & dpp_from_right(X[3 + i]));
                }
                unsigned T = ((X[0] + X[1]) + X[2]) + ((X[3] + X[4]) + (X[5] + X[6]));
                Row o;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    o.v[m] = (T >> 4) & 0xffffu;
                    if (m < 7) T = (T - X[m]) + X[m + 7];
                }
                {
                    u32x4 q;
                    q[0] = o.v[0] | (o.v[1] << 16);
                    q[1] = o.v[2] | (o.v[3] << 16);
                    q[2] = o.v[4] | (o.v[5] << 16);
                    q[3] = o.v[6] | (o.v[7] << 16);
                    __builtin_amdgcn_raw_buffer_store_b128(q, rs, (int)vo_out, r * se * 2, kAux);  // (lanes that keep nothing: dropped by the range check)
                }
                prev = o;
                cur = nxt;
                nxt = unpack(pre);
            }
        }
        if (K * (t + 1) < rows - 1 && (pub_right || pub_left)) {
            unsigned* to = pub_right ? mb_at((t + 1) & 1, wave + 1, 0, slot) : mb_at((t + 1) & 1, wave - 1, 1, slot);
            reinterpret_cast<uint4*>(to)[0] = make_uint4(prev.v[0], prev.v[1], prev.v[2], prev.v[3]);
            reinterpret_cast<uint4*>(to)[1] = make_uint4(prev.v[4], prev.v[5], prev.v[6], prev.v[7]);
        }
    }
}

// The chain for float samples: the same schedule around the row body of k_smooth_f32_strips (every sum in the
// reference's order).
template <bool GROUPED>
__global__ void __launch_bounds__(kSmoothThreads) k_smooth_f32_chain(PoolArgs pool, ChainArgs ch, int nw, int lanes, int pass_rounds,
                                                                    int cycle)
{
    using namespace v3c;
    extern __shared__ __align__(16) unsigned char smem[];
    if (ch.only_if && *ch.only_if == 0) return;  // the guarded redo of a launch that did not time out: nothing to do
    const int se = pool.stride_e;
    const int nl = se >> 3;  // lanes that own columns
    const int groups = GROUPED ? ch.groups : 1, slack = GROUPED ? ch.slack : 0;
    const int b = GROUPED ? (int)blockIdx.x / groups : (int)blockIdx.x, grp = GROUPED ? (int)blockIdx.x % groups : 0;
    const size_t bufsz = (size_t)se * (pool.bh + 1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int ps = (tid >> 6) / nw, wave = (tid >> 6) % nw;  // which pass of the `lanes` in flight, which strip of it
    int gl;
    bool ghost;
    if (wave == 0) {
        gl = lane;
        ghost = nw > 1 && lane >= 64 - GH;
    } else {
        gl = kFirst + kInner * (wave - 1) + (lane - GH);
        ghost = lane < GH || (lane >= 64 - GH && wave < nw - 1);
    }
    const bool live = gl < nl, real = live && !ghost;
    const int x0 = live ? gl * 8 : 0;  // dead lanes shadow column 0 and store nothing
    const unsigned first_mask = live && gl == 0 ? 0xffffffffu : 0u, last_mask = live && gl == nl - 1 ? 0xffffffffu : 0u;
    // mailbox: [pass in flight][copy][wave][side][slot][8 registers]
    float* mb = reinterpret_cast<float*>(smem) + (size_t)ps * (2 * nw * 2 * GH * 8);
    auto mb_at = [&](int copy, int w, int side, int slot) { return mb + ((((copy * nw + w) * 2 + side) * GH + slot) * 8); };
    const bool pub_right = lane >= 64 - 2 * GH && lane < 64 - GH && wave < nw - 1;
    const bool pub_left = lane >= GH && lane < 2 * GH && wave > 0;
    const bool recv = ghost && live;
    const int slot = lane < GH ? lane : lane >= 64 - GH ? lane - (64 - GH) : pub_right ? lane - (64 - 2 * GH) : lane - GH;

    struct Row {
        float v[8];
    };
    struct Raw {
        float4 lo, hi;
    };
    auto unpack = [](const Raw& q) {
        Row r;
        r.v[0] = q.lo.x; r.v[1] = q.lo.y; r.v[2] = q.lo.z; r.v[3] = q.lo.w;
        r.v[4] = q.hi.x; r.v[5] = q.hi.y; r.v[6] = q.hi.z; r.v[7] = q.hi.w;
        return r;
    };
    auto pick = [](unsigned mask, float edge, float other) {  // bitwise, as in k_smooth_u8_strips
        return __uint_as_float((mask & __float_as_uint(edge)) | (~mask & __float_as_uint(other)));
    };
    const int rows = ch.rows > 0 && ch.rows < pool.bh ? ch.rows : pool.bh;  // rows 1 .. rows - 1 are smoothed
    const int origin = ch.origin + (int)blockIdx.y * ch.chain_step;           // blockIdx.y: one of several independent chains
    const int slots = lanes * groups;  // passes in flight per buffer
    const int last_slot = (ch.npass - 1) % slots;  // the chain is over when its last pass is: no rounds for slots nothing fills
    const int total = ((ch.npass - 1) / slots) * cycle + last_slot * kChainLag + (last_slot / lanes) * slack + pass_rounds;

    // state of the pass this wave is working on; addresses as in k_smooth_u8_chain: one buffer resource over the ring, the
    // slot, cost buffer and pool row in the scalar offset, 32 bits per lane (a row is ONE sixteen-byte access, coherent or not)
    constexpr unsigned kNoAccess = 0xffffffffu;
    constexpr int kAux = GROUPED ? 16 : 0;  // sc1
    const unsigned long long ring_bytes = (unsigned long long)pool.slot_bytes * (unsigned long long)pool.slot_mod;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pool.base, 0, ring_bytes < 0xfffffffeull ? (int)(unsigned)ring_bytes : (int)0xfffffffeu, 0x00020000);
    unsigned vo_fresh = kNoAccess;  // this lane's offset for rows 1 .. nr: in the pass's own slot where its k_prepare wrote ...
    unsigned vo_stale = kNoAccess;  // ... and for every other row: in the slot of the pass before it
    unsigned vo_out = kNoAccess;    // where it stores (kNoAccess: ghost and dead lanes)
    int nr = 0;
    Row prev{}, cur{}, nxt{};
    constexpr int kAhead = K;
    Raw ring[kAhead] = {};
    auto load = [&](int row) {
        row = row <= pool.bh ? row : pool.bh;  // past the end: loaded, never used
        const bool fresh = row >= 1 && row <= nr;  // (uniform)
        const unsigned of = vo_fresh, os = vo_stale;  // (by value, see k_smooth_u16_chain)
        const int voff = (int)(fresh ? of : os);
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, row * se * 4, kAux), c = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, row * se * 4 + 16, kAux);
        Raw q;
        q.lo = make_float4(__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]), __uint_as_float(a[3]));
        q.hi = make_float4(__uint_as_float(c[0]), __uint_as_float(c[1]), __uint_as_float(c[2]), __uint_as_float(c[3]));
        return q;
    };

    ChainSync sync{};
    if constexpr (GROUPED) sync.init(ch, b, grp);
    const int gslot = grp * lanes + ps;  // the slot among all of the buffer's
    for (int round = 0; round < total; ++round) {
        if constexpr (GROUPED) chain_release_barrier();
        else __syncthreads();
        if constexpr (GROUPED) sync.enter(round, ps == 0, tid);
        const int rel = round - (gslot * kChainLag + grp * slack);
        if (rel < 0) continue;
        const int j = (rel / cycle) * slots + gslot, t = rel % cycle - 1;
        if (j >= ch.npass || t >= pass_rounds - 1) continue;
        if (t < 0) {  // the pass's first rows
            const int k = j % ch.pn;
            const int64_t s_own = (origin + 1 + j) % pool.slot_mod, s_before = (origin + j) % pool.slot_mod;
            const unsigned own_off = (unsigned)(s_own * pool.slot_bytes + (int64_t)b * (int64_t)bufsz * 4);        // (uniform; the ring is
            const unsigned before_off = (unsigned)(s_before * pool.slot_bytes + (int64_t)b * (int64_t)bufsz * 4);  // smaller than 4 GB)
            vo_stale = before_off + x0 * 4;
            vo_fresh = (x0 < ch.w[k] ? own_off : before_off) + x0 * 4;
            vo_out = real ? own_off + x0 * 4 : kNoAccess;
            nr = ch.nr[k];
            prev = unpack(load(0));
            cur = unpack(load(1));
            nxt = unpack(load(2));
#pragma unroll
            for (int u = 0; u < kAhead; ++u) ring[u] = load(3 + u);
            continue;
        }
        if (t > 0 && recv) {  // the ghosts take over what the seam lanes held after the block before
            const float4* from = reinterpret_cast<const float4*>(mb_at(t & 1, wave, lane < GH ? 0 : 1, slot));
            const float4 lo = from[0], hi = from[1];
            prev.v[0] = lo.x; prev.v[1] = lo.y; prev.v[2] = lo.z; prev.v[3] = lo.w;
            prev.v[4] = hi.x; prev.v[5] = hi.y; prev.v[6] = hi.z; prev.v[7] = hi.w;
        }
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int r = K * t + 1 + u;
            if (r < rows) {  // uniform
                const Raw pre = ring[u];
                ring[u] = load(r + 2 + kAhead);
                float X[14];  // as k_smooth_f32_strips
#pragma unroll
                for (int i = 0; i < 8; ++i) X[3 + i] = (prev.v[i] + cur.v[i]) + nxt.v[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    X[i] = pick(first_mask, X[3], __uint_as_float(dpp_from_left(__float_as_uint(X[8 + i]))));
                    X[11 + i] = pick(last_mask, X[10], __uint_as_float(dpp_from_right(__float_as_uint(X[3 + i]))));
                }
                Row o;
#pragma unroll
                for (int m = 0; m < 8; ++m)  // left to right, SangNom2.cpp:152
                    o.v[m] = ((((((X[m] + X[m + 1]) + X[m + 2]) + X[m + 3]) + X[m + 4]) + X[m + 5]) + X[m + 6]) * 0.0625f;
                {
                    u32x4 qa, qb;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        qa[m] = __float_as_uint(o.v[m]);
                        qb[m] = __float_as_uint(o.v[4 + m]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(qa, rs, (int)vo_out, r * se * 4, kAux);  // (lanes that keep nothing: dropped by the range check)
                    __builtin_amdgcn_raw_buffer_store_b128(qb, rs, (int)vo_out, r * se * 4 + 16, kAux);
                }
                prev = o;
                cur = nxt;
                nxt = unpack(pre);
            }
        }
        if (K * (t + 1) < rows - 1 && (pub_right || pub_left)) {
            float* to = pub_right ? mb_at((t + 1) & 1, wave + 1, 0, slot) : mb_at((t + 1) & 1, wave - 1, 1, slot);
            reinterpret_cast<float4*>(to)[0] = make_float4(prev.v[0], prev.v[1], prev.v[2], prev.v[3]);
            reinterpret_cast<float4*>(to)[1] = make_float4(prev.v[4], prev.v[5], prev.v[6], prev.v[7]);
        }
    }
}

// 9..16-bit pools: eight columns per thread as well, one 32-bit sum per register (seven 3-row sums of 16-bit samples need 21
// bits), a sliding box (two instructions per further column instead of three three-operand adds), one 16-byte row
// access and two 16-byte LDS reads per thread and row: 55 vector instructions per 8 columns against 79 per 4
// (one 2160p pool 830 -> 613 us).
__global__ void __launch_bounds__(kSmoothThreads) k_smooth_u16x8(PoolArgs pool, int slot0)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int se = pool.stride_e;  // a multiple of 32
    const int nt = se >> 3;        // threads that own columns
    uint4* line0 = reinterpret_cast<uint4*>(smem);  // [thread][2]: sums of its columns 0..3 and 4..7
    uint4* line1 = line0 + 2 * nt;
    const int b = blockIdx.x;
    const int f = blockIdx.y;
    if (pool.guard && pool.guard[pool.guard_single ? 0 : f] == 0) return;
    const size_t bufsz = (size_t)se * (pool.bh + 1);
    uint16_t* buf = reinterpret_cast<uint16_t*>(pool.base + slot_of(pool, slot0, f) * pool.slot_bytes) + (size_t)b * bufsz;
    const int tid = threadIdx.x;
    const bool active = tid < nt;
    const int t = active ? tid : 0;  // idle lanes of the last wave shadow thread 0 and store nothing
    const bool first = t == 0, last = t == nt - 1;
    const int tl = first ? 0 : t - 1, tr = last ? t : t + 1;

    struct Row {
        unsigned v[8];
    };
    auto unpack = [](uint4 q) {
        Row r;
        r.v[0] = q.x & 0xffffu; r.v[1] = q.x >> 16;
        r.v[2] = q.y & 0xffffu; r.v[3] = q.y >> 16;
        r.v[4] = q.z & 0xffffu; r.v[5] = q.z >> 16;
        r.v[6] = q.w & 0xffffu; r.v[7] = q.w >> 16;
        return r;
    };
    auto load = [&](int row) { return *reinterpret_cast<const uint4*>(buf + (size_t)row * se + 8 * t); };
    auto row_or_last = [&](int row) { return row <= pool.bh ? row : pool.bh; };  // past the end: loaded, never used
    Row prev = unpack(load(0)), cur = unpack(load(1)), nxt = unpack(load(row_or_last(2)));
    const int rows = pool.rows > 0 && pool.rows < pool.bh ? pool.rows : pool.bh;  // rows 1 .. rows - 1 are smoothed
    constexpr int kAhead = 4;  // rows in flight, as in k_smooth_u8x2
    uint4 ring[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) ring[u] = load(row_or_last(3 + u));
    auto row_step = [&](int r, uint4 pre) {  // pre = row r + 2
        uint4* line = (r & 1) ? line1 : line0;
        unsigned X[14];  // sums of columns 8t - 3 .. 8t + 10
#pragma unroll
        for (int i = 0; i < 8; ++i) X[3 + i] = prev.v[i] + cur.v[i] + nxt.v[i];
        if (active) {
            line[2 * t] = make_uint4(X[3], X[4], X[5], X[6]);
            line[2 * t + 1] = make_uint4(X[7], X[8], X[9], X[10]);
        }
        __syncthreads();
        const uint4 lf = line[2 * tl + 1], rt = line[2 * tr];
        // the pool row is clamped at both ends (SangNom2.cpp:144-150)
        X[0] = first ? X[3] : lf.y;
        X[1] = first ? X[3] : lf.z;
        X[2] = first ? X[3] : lf.w;
        X[11] = last ? X[10] : rt.x;
        X[12] = last ? X[10] : rt.y;
        X[13] = last ? X[10] : rt.z;
        unsigned T = ((X[0] + X[1]) + X[2]) + ((X[3] + X[4]) + (X[5] + X[6]));
        Row o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            o.v[k] = (T >> 4) & 0xffffu;  // (sum / 16) wraps to uint16_t, SangNom2.cpp:152; integer sums: any order
            if (k < 7) T = (T - X[k]) + X[k + 7];
        }
        uint4 q;
        q.x = o.v[0] | (o.v[1] << 16);
        q.y = o.v[2] | (o.v[3] << 16);
        q.z = o.v[4] | (o.v[5] << 16);
        q.w = o.v[6] | (o.v[7] << 16);
        if (active) *reinterpret_cast<uint4*>(buf + (size_t)r * se + 8 * t) = q;
        prev = o;
        cur = nxt;
        nxt = unpack(pre);
    };
    for (int r = 1; r < rows; r += kAhead) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (r + u < rows) {  // uniform
                const uint4 pre = ring[u];
                ring[u] = load(row_or_last(r + u + 2 + kAhead));
                row_step(r + u, pre);
            }
        }
    }
}

// 9..16-bit pools of 512 columns and more in strips, as k_smooth_u8_strips: 60 lanes x 8 columns per wave, one 32-bit
// sum per register, the three sums on either side over DPP, ghost lanes refreshed from the mailbox every K rows.
__global__ void __launch_bounds__(kSmoothThreads) k_smooth_u16_strips(PoolArgs pool, int slot0)
{
    using namespace v3c;
    extern __shared__ __align__(16) unsigned char smem[];
    const int se = pool.stride_e;
    const int nl = se >> 3;             // lanes that own columns
    const int nw = (int)blockDim.x >> 6;
    const int b = blockIdx.x;
    const int f = blockIdx.y;
    if (pool.guard && pool.guard[pool.guard_single ? 0 : f] == 0) return;
    const size_t bufsz = (size_t)se * (pool.bh + 1);
    uint16_t* buf = reinterpret_cast<uint16_t*>(pool.base + slot_of(pool, slot0, f) * pool.slot_bytes) + (size_t)b * bufsz;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int gl;
    bool ghost;
    if (wave == 0) {
        gl = lane;
        ghost = nw > 1 && lane >= 64 - GH;
    } else {
        gl = kFirst + kInner * (wave - 1) + (lane - GH);
        ghost = lane < GH || (lane >= 64 - GH && wave < nw - 1);
    }
    const bool live = gl < nl, real = live && !ghost;
    const int x0 = live ? gl * 8 : 0;  // dead lanes shadow column 0 and store nothing
    const unsigned first_mask = live && gl == 0 ? 0xffffffffu : 0u, last_mask = live && gl == nl - 1 ? 0xffffffffu : 0u;
    // mailbox: [copy][wave][side][slot][8 registers]
    unsigned* mb = reinterpret_cast<unsigned*>(smem);
    auto mb_at = [&](int copy, int w, int side, int slot) { return mb + ((((copy * nw + w) * 2 + side) * GH + slot) * 8); };
    const bool pub_right = lane >= 64 - 2 * GH && lane < 64 - GH && wave < nw - 1;  // feeds the next wave's left ghosts
    const bool pub_left = lane >= GH && lane < 2 * GH && wave > 0;                   // ... the previous wave's right ghosts
    const bool recv = ghost && live;
    const int slot = lane < GH ? lane : lane >= 64 - GH ? lane - (64 - GH) : pub_right ? lane - (64 - 2 * GH) : lane - GH;

    struct Row {
        unsigned v[8];
    };
    auto unpack = [](uint4 q) {
        Row r;
        r.v[0] = q.x & 0xffffu; r.v[1] = q.x >> 16;
        r.v[2] = q.y & 0xffffu; r.v[3] = q.y >> 16;
        r.v[4] = q.z & 0xffffu; r.v[5] = q.z >> 16;
        r.v[6] = q.w & 0xffffu; r.v[7] = q.w >> 16;
        return r;
    };
    auto load = [&](int row) { return *reinterpret_cast<const uint4*>(buf + (size_t)row * se + x0); };
    auto row_or_last = [&](int row) { return row <= pool.bh ? row : pool.bh; };  // past the end: loaded, never used
    constexpr int kAhead = K;  // >= the drift between two waves + 1
    Row prev = unpack(load(0)), cur = unpack(load(1)), nxt = unpack(load(row_or_last(2)));
    uint4 ring[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) ring[u] = load(row_or_last(3 + u));
    __syncthreads();  // nobody stores before everybody has fetched its first rows
    const int rows = pool.rows > 0 && pool.rows < pool.bh ? pool.rows : pool.bh;  // rows 1 .. rows - 1 are smoothed
    auto row_step = [&](int r, uint4 pre) {  // pre = row r + 2
        if (r > 1 && (r - 1) % K == 0) {  // the ghosts take over what the seam lanes held after row r - 1
            __syncthreads();
            if (recv) {
                const uint4* from = reinterpret_cast<const uint4*>(mb_at((r / K) & 1, wave, lane < GH ? 0 : 1, slot));
                const uint4 a = from[0], c = from[1];
                prev.v[0] = a.x; prev.v[1] = a.y; prev.v[2] = a.z; prev.v[3] = a.w;
                prev.v[4] = c.x; prev.v[5] = c.y; prev.v[6] = c.z; prev.v[7] = c.w;
            }
        }
        unsigned X[14];  // sums of columns x0 - 3 .. x0 + 10
#pragma unroll
        for (int i = 0; i < 8; ++i) X[3 + i] = prev.v[i] + cur.v[i] + nxt.v[i];
        // the pool row is clamped at both ends (SangNom2.cpp:144-150); bitwise selects, as in k_smooth_u8_strips
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            X[i] = (first_mask & X[3]) | (~first_mask & dpp_from_left(X[8 + i]));
            X[11 + i] = (last_mask & X[10]) | (~last_mask & dpp_from_right(X[3 + i]));
        }
        unsigned T = ((X[0] + X[1]) + X[2]) + ((X[3] + X[4]) + (X[5] + X[6]));
        Row o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            o.v[k] = (T >> 4) & 0xffffu;  // (sum / 16) wraps to uint16_t, SangNom2.cpp:152; integer sums: any order
            if (k < 7) T = (T - X[k]) + X[k + 7];
        }
        if (real) {
            uint4 q;
            q.x = o.v[0] | (o.v[1] << 16);
            q.y = o.v[2] | (o.v[3] << 16);
            q.z = o.v[4] | (o.v[5] << 16);
            q.w = o.v[6] | (o.v[7] << 16);
            *reinterpret_cast<uint4*>(buf + (size_t)r * se + x0) = q;
        }
        prev = o;
        cur = nxt;
        nxt = unpack(pre);
        if (r % K == 0 && r < rows - 1 && (pub_right || pub_left)) {
            uint4* to = reinterpret_cast<uint4*>(pub_right ? mb_at(((r + 1) / K) & 1, wave + 1, 0, slot)
                                                           : mb_at(((r + 1) / K) & 1, wave - 1, 1, slot));
            to[0] = make_uint4(prev.v[0], prev.v[1], prev.v[2], prev.v[3]);
            to[1] = make_uint4(prev.v[4], prev.v[5], prev.v[6], prev.v[7]);
        }
    };
    for (int r = 1; r < rows; r += kAhead) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (r + u < rows) {  // uniform
                const uint4 pre = ring[u];
                ring[u] = load(row_or_last(r + u + 2 + kAhead));
                row_step(r + u, pre);
            }
        }
    }
}

// Float pools: eight columns per thread too.  The sums keep the reference's order -- (a + b) + c for the three rows, the
// seven taps left to right (SangNom2.cpp:144-152) -- so nothing slides; what is saved is the per-thread overhead (half
// the waves, one 32-byte row access, two 16-byte LDS reads per thread and row) and the row latency (four rows in flight).
__global__ void __launch_bounds__(kSmoothThreads) k_smooth_f32x8(PoolArgs pool, int slot0)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int se = pool.stride_e;  // a multiple of 32
    const int nt = se >> 3;        // threads that own columns
    float4* line0 = reinterpret_cast<float4*>(smem);  // [thread][2]: sums of its columns 0..3 and 4..7
    float4* line1 = line0 + 2 * nt;
    const int b = blockIdx.x;
    const int f = blockIdx.y;
    if (pool.guard && pool.guard[pool.guard_single ? 0 : f] == 0) return;
    const size_t bufsz = (size_t)se * (pool.bh + 1);
    float* buf = reinterpret_cast<float*>(pool.base + slot_of(pool, slot0, f) * pool.slot_bytes) + (size_t)b * bufsz;
    const int tid = threadIdx.x;
    const bool active = tid < nt;
    const int t = active ? tid : 0;  // idle lanes of the last wave shadow thread 0 and store nothing
    const bool first = t == 0, last = t == nt - 1;
    const int tl = first ? 0 : t - 1, tr = last ? t : t + 1;

    struct Row {
        float4 lo, hi;
        __device__ __forceinline__ float at(int i) const
        {
            return i == 0 ? lo.x : i == 1 ? lo.y : i == 2 ? lo.z : i == 3 ? lo.w : i == 4 ? hi.x : i == 5 ? hi.y : i == 6 ? hi.z : hi.w;
        }
    };
    auto load = [&](int row) {
        const float4* p = reinterpret_cast<const float4*>(buf + (size_t)row * se + 8 * t);
        Row r;
        r.lo = p[0];
        r.hi = p[1];
        return r;
    };
    auto row_or_last = [&](int row) { return row <= pool.bh ? row : pool.bh; };  // past the end: loaded, never used
    Row prev = load(0), cur = load(1), nxt = load(row_or_last(2));
    const int rows = pool.rows > 0 && pool.rows < pool.bh ? pool.rows : pool.bh;  // rows 1 .. rows - 1 are smoothed
    constexpr int kAhead = 4;  // rows in flight, as in k_smooth_u8x2
    Row ring[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) ring[u] = load(row_or_last(3 + u));
    auto row_step = [&](int r, const Row& pre) {  // pre = row r + 2
        float4* line = (r & 1) ? line1 : line0;
        float X[14];  // sums of columns 8t - 3 .. 8t + 10
#pragma unroll
        for (int i = 0; i < 8; ++i) X[3 + i] = (prev.at(i) + cur.at(i)) + nxt.at(i);
        if (active) {
            line[2 * t] = make_float4(X[3], X[4], X[5], X[6]);
            line[2 * t + 1] = make_float4(X[7], X[8], X[9], X[10]);
        }
        __syncthreads();
        const float4 lf = line[2 * tl + 1], rt = line[2 * tr];
        // the pool row is clamped at both ends (SangNom2.cpp:144-150)
        X[0] = first ? X[3] : lf.y;
        X[1] = first ? X[3] : lf.z;
        X[2] = first ? X[3] : lf.w;
        X[11] = last ? X[10] : rt.x;
        X[12] = last ? X[10] : rt.y;
        X[13] = last ? X[10] : rt.z;
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)  // left to right, SangNom2.cpp:152
            o[k] = ((((((X[k] + X[k + 1]) + X[k + 2]) + X[k + 3]) + X[k + 4]) + X[k + 5]) + X[k + 6]) * 0.0625f;
        Row out;
        out.lo = make_float4(o[0], o[1], o[2], o[3]);
        out.hi = make_float4(o[4], o[5], o[6], o[7]);
        if (active) {
            float4* q = reinterpret_cast<float4*>(buf + (size_t)r * se + 8 * t);
            q[0] = out.lo;
            q[1] = out.hi;
        }
        prev = out;
        cur = nxt;
        nxt = pre;
    };
    for (int r = 1; r < rows; r += kAhead) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (r + u < rows) {  // uniform
                const Row pre = ring[u];
                ring[u] = load(row_or_last(r + u + 2 + kAhead));
                row_step(r + u, pre);
            }
        }
    }
}

// Pools narrower than 1024 columns: one column per thread, columns strided by the workgroup size.  (Measured on
// 720-wide clips: 12 % faster per row than the NC = 1 instance of the kernel above, which wins from 1024 columns on.)
// Float pools of 512 columns and more in strips, as k_smooth_u16_strips; every sum in the reference's order.  What the
// ghost lanes of a strip's outer lanes compute after the seam (anything, NaN included) never reaches an owned column
// inside the K rows between two refreshes.
__global__ void __launch_bounds__(kSmoothThreads) k_smooth_f32_strips(PoolArgs pool, int slot0)
{
    using namespace v3c;
    extern __shared__ __align__(16) unsigned char smem[];
    const int se = pool.stride_e;
    const int nl = se >> 3;             // lanes that own columns
    const int nw = (int)blockDim.x >> 6;
    const int b = blockIdx.x;
    const int f = blockIdx.y;
    if (pool.guard && pool.guard[pool.guard_single ? 0 : f] == 0) return;
    const size_t bufsz = (size_t)se * (pool.bh + 1);
    float* buf = reinterpret_cast<float*>(pool.base + slot_of(pool, slot0, f) * pool.slot_bytes) + (size_t)b * bufsz;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int gl;
    bool ghost;
    if (wave == 0) {
        gl = lane;
        ghost = nw > 1 && lane >= 64 - GH;
    } else {
        gl = kFirst + kInner * (wave - 1) + (lane - GH);
        ghost = lane < GH || (lane >= 64 - GH && wave < nw - 1);
    }
    const bool live = gl < nl, real = live && !ghost;
    const int x0 = live ? gl * 8 : 0;  // dead lanes shadow column 0 and store nothing
    const unsigned first_mask = live && gl == 0 ? 0xffffffffu : 0u, last_mask = live && gl == nl - 1 ? 0xffffffffu : 0u;
    // mailbox: [copy][wave][side][slot][8 registers]
    float* mb = reinterpret_cast<float*>(smem);
    auto mb_at = [&](int copy, int w, int side, int slot) { return mb + ((((copy * nw + w) * 2 + side) * GH + slot) * 8); };
    const bool pub_right = lane >= 64 - 2 * GH && lane < 64 - GH && wave < nw - 1;  // feeds the next wave's left ghosts
    const bool pub_left = lane >= GH && lane < 2 * GH && wave > 0;                   // ... the previous wave's right ghosts
    const bool recv = ghost && live;
    const int slot = lane < GH ? lane : lane >= 64 - GH ? lane - (64 - GH) : pub_right ? lane - (64 - 2 * GH) : lane - GH;

    struct Row {
        float v[8];
    };
    auto from_vec = [](float4 a, float4 c) {
        Row r;
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
        r.v[4] = c.x; r.v[5] = c.y; r.v[6] = c.z; r.v[7] = c.w;
        return r;
    };
    struct Raw {
        float4 lo, hi;
    };
    auto load = [&](int row) {
        const float4* p = reinterpret_cast<const float4*>(buf + (size_t)row * se + x0);
        Raw q;
        q.lo = p[0];
        q.hi = p[1];
        return q;
    };
    auto unpack = [&](const Raw& q) { return from_vec(q.lo, q.hi); };
    auto row_or_last = [&](int row) { return row <= pool.bh ? row : pool.bh; };  // past the end: loaded, never used
    constexpr int kAhead = K;  // >= the drift between two waves + 1
    Row prev = unpack(load(0)), cur = unpack(load(1)), nxt = unpack(load(row_or_last(2)));
    Raw ring[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) ring[u] = load(row_or_last(3 + u));
    __syncthreads();  // nobody stores before everybody has fetched its first rows
    const int rows = pool.rows > 0 && pool.rows < pool.bh ? pool.rows : pool.bh;  // rows 1 .. rows - 1 are smoothed
    auto pick = [](unsigned mask, float edge, float other) {  // bitwise, as in k_smooth_u8_strips
        return __uint_as_float((mask & __float_as_uint(edge)) | (~mask & __float_as_uint(other)));
    };
    auto row_step = [&](int r, const Raw& pre) {  // pre = row r + 2
        if (r > 1 && (r - 1) % K == 0) {  // the ghosts take over what the seam lanes held after row r - 1
            __syncthreads();
            if (recv) {
                const float4* from = reinterpret_cast<const float4*>(mb_at((r / K) & 1, wave, lane < GH ? 0 : 1, slot));
                prev = from_vec(from[0], from[1]);
            }
        }
        float X[14];  // sums of columns x0 - 3 .. x0 + 10
#pragma unroll
        for (int i = 0; i < 8; ++i) X[3 + i] = (prev.v[i] + cur.v[i]) + nxt.v[i];
        // the pool row is clamped at both ends (SangNom2.cpp:144-150)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            X[i] = pick(first_mask, X[3], __uint_as_float(dpp_from_left(__float_as_uint(X[8 + i]))));
            X[11 + i] = pick(last_mask, X[10], __uint_as_float(dpp_from_right(__float_as_uint(X[3 + i]))));
        }
        Row o;
#pragma unroll
        for (int k = 0; k < 8; ++k)  // left to right, SangNom2.cpp:152
            o.v[k] = ((((((X[k] + X[k + 1]) + X[k + 2]) + X[k + 3]) + X[k + 4]) + X[k + 5]) + X[k + 6]) * 0.0625f;
        if (real) {
            float4* q = reinterpret_cast<float4*>(buf + (size_t)r * se + x0);
            q[0] = make_float4(o.v[0], o.v[1], o.v[2], o.v[3]);
            q[1] = make_float4(o.v[4], o.v[5], o.v[6], o.v[7]);
        }
        prev = o;
        cur = nxt;
        nxt = unpack(pre);
        if (r % K == 0 && r < rows - 1 && (pub_right || pub_left)) {
            float4* to = reinterpret_cast<float4*>(pub_right ? mb_at(((r + 1) / K) & 1, wave + 1, 0, slot)
                                                             : mb_at(((r + 1) / K) & 1, wave - 1, 1, slot));
            to[0] = make_float4(prev.v[0], prev.v[1], prev.v[2], prev.v[3]);
            to[1] = make_float4(prev.v[4], prev.v[5], prev.v[6], prev.v[7]);
        }
    };
    for (int r = 1; r < rows; r += kAhead) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (r + u < rows) {  // uniform
                const Raw pre = ring[u];
                ring[u] = load(row_or_last(r + u + 2 + kAhead));
                row_step(r + u, pre);
            }
        }
    }
}

template <class T, int NC>
__global__ void __launch_bounds__(kSmoothThreads) k_smooth_strided(PoolArgs pool, int slot0)
{
    using P = Px<T>;
    using W = typename P::W;
    extern __shared__ __align__(16) unsigned char smem[];
    const int se = pool.stride_e;
    W* line0 = reinterpret_cast<W*>(smem);
    W* line1 = line0 + se;
    const int b = blockIdx.x;
    const int f = blockIdx.y;
    if (pool.guard && pool.guard[pool.guard_single ? 0 : f] == 0) return;
    const size_t bufsz = (size_t)se * (pool.bh + 1);
    T* buf = reinterpret_cast<T*>(pool.base + slot_of(pool, slot0, f) * pool.slot_bytes) + (size_t)b * bufsz;
    const int tid = threadIdx.x;

    W prev[NC], cur[NC], nxt[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int x = tid + k * kSmoothThreads;
        const bool in = x < se;
        prev[k] = in ? (W)buf[x] : (W)0;
        cur[k] = in ? (W)buf[(size_t)se + x] : (W)0;
        nxt[k] = (in && pool.bh >= 2) ? (W)buf[(size_t)2 * se + x] : (W)0;
    }
    const int rows = pool.rows > 0 && pool.rows < pool.bh ? pool.rows : pool.bh;  // rows 1 .. rows - 1 are smoothed
    for (int r = 1; r < rows; ++r) {
        W pre[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int x = tid + k * kSmoothThreads;
            pre[k] = (x < se && r + 2 <= pool.bh) ? (W)buf[(size_t)(r + 2) * se + x] : (W)0;
        }
        W* line = (r & 1) ? line1 : line0;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int x = tid + k * kSmoothThreads;
            if (x < se) line[x] = P::sum3(prev[k], cur[k], nxt[k]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int x = tid + k * kSmoothThreads;
            if (x < se) {
                W s;
                {
                    const int q0 = x - 3 < 0 ? 0 : x - 3;
                    const int q1 = x - 2 < 0 ? 0 : x - 2;
                    s = line[q0] + line[q1];  // left-to-right, SangNom2.cpp:152
                }
                s = s + line[x - 1 < 0 ? 0 : x - 1];
                s = s + line[x];
                s = s + line[x + 1 > se - 1 ? se - 1 : x + 1];
                s = s + line[x + 2 > se - 1 ? se - 1 : x + 2];
                s = s + line[x + 3 > se - 1 ? se - 1 : x + 3];
                const W o = P::div16(s);
                buf[(size_t)r * se + x] = (T)o;
                prev[k] = o;
            }
            cur[k] = nxt[k];
            nxt[k] = pre[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Stage 3: finalizePlane_c, SangNom2.cpp:161-257.  One thread per interpolated pixel.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256)
k_finalize(PlaneArgs p, PoolArgs pool, int slot0, typename Px<T>::W thr)
{
    using P = Px<T>;
    using W = typename P::W;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int f = blockIdx.z;
    if (p.guard && p.guard[p.guard_single ? 0 : f] == 0) return;
    if (x >= p.w) return;
    uint8_t* plane = p.dst + (int64_t)f * p.dst_frame_stride;
    const T* cl = reinterpret_cast<const T*>(plane + (int64_t)(p.offset + 2 * y) * p.dst_pitch);
    const T* nl = reinterpret_cast<const T*>(plane + (int64_t)(p.offset + 2 * y + 2) * p.dst_pitch);
    T* ol = reinterpret_cast<T*>(plane + (int64_t)(p.offset + 2 * y + 1) * p.dst_pitch);
    Taps<T> t;
    t.load(cl, nl, x, p.w);
    const T* pb = reinterpret_cast<const T*>(pool.base + slot_of(pool, slot0, f) * pool.slot_bytes);
    const size_t bufsz = (size_t)pool.stride_e * (pool.bh + 1);
    const size_t at = (size_t)(y + 1) * pool.stride_e + x;
    W v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = (W)pb[i * bufsz + at];
    W m = v[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) m = v[i] < m ? v[i] : m;  // std::min(m, v[i])

    W r = P::avg(t.c[0], t.n[6]);                       // buf[0]: avg(c-3, n+3), lowest priority
    r = (v[8] == m) ? P::avg(t.c[6], t.n[0]) : r;       // buf[8]: avg(c+3, n-3)
    r = (v[1] == m) ? P::avg(t.c[1], t.n[5]) : r;
    r = (v[7] == m) ? P::avg(t.c[5], t.n[1]) : r;
    r = (v[2] == m) ? P::avg(t.c[2], t.n[4]) : r;
    r = (v[6] == m) ? P::avg(t.c[4], t.n[2]) : r;
    r = (v[3] == m) ? P::avg(t.f1, t.f2) : r;
    r = (v[5] == m) ? P::avg(t.b1, t.b2) : r;
    r = (v[4] == m || m > thr) ? P::avg(t.c[3], t.n[3]) : r;  // SangNom2.cpp:214
    ol[x] = (T)r;
}

template <class T>
static hipError_t launch_pool_plane_t(hipStream_t st, const PlaneArgs& p, const PoolArgs& pool,
                                      double threshold, int nframes, int slot0)
{
    using W = typename Px<T>::W;
    const int nr = p.h_out / 2 - 1;
    if (nr > 0) {
        dim3 grid((p.w + 255) / 256, nr, nframes), block(256);
        hipLaunchKernelGGL(k_prepare<T>, grid, block, 0, st, p, pool, slot0);
    }
    if (std::is_same<T, uint8_t>::value && pool.bh > 1 && pool.stride_e >= 512 && v3c::strips_for(pool.stride_e / 8) <= kSmoothThreads / 64) {
        const int nw = v3c::strips_for(pool.stride_e / 8);
        const size_t lds = (size_t)2 * nw * 2 * v3c::GH * 4 * sizeof(unsigned);
        hipLaunchKernelGGL(k_smooth_u8_strips, dim3(kBuffers, nframes), dim3(nw * 64), lds, st, pool, slot0);
    } else if (std::is_same<T, uint8_t>::value && pool.bh > 1 && pool.stride_e >= 256 && pool.stride_e <= 8 * kSmoothThreads) {
        const int threads = ((pool.stride_e / 8) + 63) / 64 * 64;
        const size_t lds = (size_t)2 * (pool.stride_e / 8) * sizeof(uint4);
        hipLaunchKernelGGL(k_smooth_u8x2, dim3(kBuffers, nframes), dim3(threads), lds, st, pool, slot0);
    } else if (std::is_same<T, uint16_t>::value && pool.bh > 1 && pool.stride_e >= 512 && v3c::strips_for(pool.stride_e / 8) <= kSmoothThreads / 64) {
        const int nw = v3c::strips_for(pool.stride_e / 8);
        const size_t lds = (size_t)2 * nw * 2 * v3c::GH * 8 * sizeof(unsigned);
        hipLaunchKernelGGL(k_smooth_u16_strips, dim3(kBuffers, nframes), dim3(nw * 64), lds, st, pool, slot0);
    } else if (std::is_same<T, uint16_t>::value && pool.bh > 1 && pool.stride_e >= 256 && pool.stride_e <= 8 * kSmoothThreads) {
        const int threads = ((pool.stride_e / 8) + 63) / 64 * 64;
        const size_t lds = (size_t)2 * 2 * (pool.stride_e / 8) * sizeof(uint4);
        hipError_t e = hipSuccess;
        if (lds > 48 * 1024) e = hipFuncSetAttribute((const void*)k_smooth_u16x8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_smooth_u16x8, dim3(kBuffers, nframes), dim3(threads), lds, st, pool, slot0);
    } else if (std::is_same<T, float>::value && pool.bh > 1 && pool.stride_e >= 512 && v3c::strips_for(pool.stride_e / 8) <= kSmoothThreads / 64 &&
               nframes <= 8) {
        // launches of more frames are bound by HBM, where the ghost lanes' second fetch of the seam columns costs more
        // than the barriers saved (sixteen 2160p frames: 6.3 k frames/s with k_smooth_f32x8, 5.9 k in strips)
        const int nw = v3c::strips_for(pool.stride_e / 8);
        const size_t lds = (size_t)2 * nw * 2 * v3c::GH * 8 * sizeof(float);
        hipLaunchKernelGGL(k_smooth_f32_strips, dim3(kBuffers, nframes), dim3(nw * 64), lds, st, pool, slot0);
    } else if (std::is_same<T, float>::value && pool.bh > 1 && pool.stride_e >= 256 && pool.stride_e <= 8 * kSmoothThreads) {
        const int threads = ((pool.stride_e / 8) + 63) / 64 * 64;
        const size_t lds = (size_t)2 * 2 * (pool.stride_e / 8) * sizeof(float4);
        hipError_t e = hipSuccess;
        if (lds > 48 * 1024) e = hipFuncSetAttribute((const void*)k_smooth_f32x8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_smooth_f32x8, dim3(kBuffers, nframes), dim3(threads), lds, st, pool, slot0);
    } else if (pool.bh > 1) {
        // columns per thread: what 1024 threads need, and 4 for every pool of 1024 columns or more (vector accesses,
        // fewer LDS round trips); narrower pools do best with one column per thread
        int nc = (pool.stride_e + kSmoothThreads - 1) / kSmoothThreads;
        nc = nc <= 1 ? 1 : nc <= 2 ? 2 : nc <= 4 ? 4 : 8;
        if (nc < 4 && pool.stride_e >= 1024) nc = 4;
        const int threads = ((pool.stride_e / nc) + 63) / 64 * 64;
        const size_t lds = (size_t)2 * pool.stride_e * sizeof(W);
        dim3 grid(kBuffers, nframes), block(threads);
        hipError_t e = hipSuccess;
#define SN_SMOOTH(NC)                                                                              \
    do {                                                                                           \
        if (lds > 48 * 1024)                                                                       \
            e = hipFuncSetAttribute((const void*)k_smooth<T, NC>,                                  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        if (e == hipSuccess) hipLaunchKernelGGL((k_smooth<T, NC>), grid, block, lds, st, pool, slot0); \
    } while (0)
        if (nc <= 1) {
            if (e == hipSuccess) hipLaunchKernelGGL((k_smooth_strided<T, 1>), grid, dim3(kSmoothThreads), lds, st, pool, slot0);
        } else if (nc <= 2) SN_SMOOTH(2);
        else if (nc <= 4) SN_SMOOTH(4);
        else SN_SMOOTH(8);
#undef SN_SMOOTH
        if (e != hipSuccess) return e;
    }
    if (nr > 0) {
        dim3 grid((p.w + 255) / 256, nr, nframes), block(256);
        W thr;
        if constexpr (sizeof(T) == 4) thr = (float)threshold; else thr = (W)threshold;
        hipLaunchKernelGGL(k_finalize<T>, grid, block, 0, st, p, pool, slot0, thr);
    }
    return hipGetLastError();
}

// ---- the chain of a history-carrying stream: stage 1 of all passes, one stage 2, stage 3 of all passes ----
int pool_chain_lanes(int bytes, int stride_e)
{
    if (stride_e < 64) return 0;
    const int nw = v3c::strips_for(stride_e / 8);
    if (bytes == 1) return nw <= kChain8Threads / 64 ? 2 * (kChain8Threads / 64 / nw) : 0;  // two passes per set of nw waves (k_smooth_u8_chain)
    return nw <= kSmoothThreads / 128 ? kSmoothThreads / 64 / nw : 0;                     // at least two passes in flight
}

template <class T>
static hipError_t launch_pool_prepare_t(hipStream_t st, const PlaneArgs& p, const PoolArgs& pool, int nframes, int slot0)
{
    const int nr = p.h_out / 2 - 1;
    if (nr > 0) hipLaunchKernelGGL(k_prepare<T>, dim3((p.w + 255) / 256, nr, nframes), dim3(256), 0, st, p, pool, slot0);
    return hipGetLastError();
}

template <class T>
static hipError_t launch_pool_finalize_t(hipStream_t st, const PlaneArgs& p, const PoolArgs& pool, double threshold, int nframes,
                                         int slot0)
{
    using W = typename Px<T>::W;
    const int nr = p.h_out / 2 - 1;
    W thr;
    if constexpr (sizeof(T) == 4) thr = (float)threshold; else thr = (W)threshold;
    if (nr > 0) hipLaunchKernelGGL(k_finalize<T>, dim3((p.w + 255) / 256, nr, nframes), dim3(256), 0, st, p, pool, slot0, thr);
    return hipGetLastError();
}

hipError_t launch_pool_prepare(hipStream_t st, const PlaneArgs& p, const PoolArgs& pool, int bytes, int nframes, int slot0)
{
    switch (bytes) {
    case 1: return launch_pool_prepare_t<uint8_t>(st, p, pool, nframes, slot0);
    case 2: return launch_pool_prepare_t<uint16_t>(st, p, pool, nframes, slot0);
    case 4: return launch_pool_prepare_t<float>(st, p, pool, nframes, slot0);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_pool_finalize(hipStream_t st, const PlaneArgs& p, const PoolArgs& pool, int bytes, double threshold, int nframes,
                                int slot0)
{
    switch (bytes) {
    case 1: return launch_pool_finalize_t<uint8_t>(st, p, pool, threshold, nframes, slot0);
    case 2: return launch_pool_finalize_t<uint16_t>(st, p, pool, threshold, nframes, slot0);
    case 4: return launch_pool_finalize_t<float>(st, p, pool, threshold, nframes, slot0);
    }
    return hipErrorInvalidValue;
}

// Workgroups per buffer a chain can be spread over (`want` or fewer).  8-bit: each gets 1 / groups of the sixteen waves
// and needs at least one pair of passes (nw waves).  16-bit and float (one pass per wave): thirty-two waves in all, sixteen
// per workgroup at most, and a workgroup needs a pass (nw waves).
static int chain_waves(int bytes, int groups)
{
    const int all = bytes == 1 ? kChain8Threads / 64 : 2 * (kSmoothThreads / 64);
    const int w = all / groups;
    return w > kSmoothThreads / 64 ? kSmoothThreads / 64 : w;
}
int pool_chain_groups(int bytes, int stride_e, int want)
{
    if (want <= 1) return 1;
    const int nw = v3c::strips_for(stride_e / 8);
    int g = 1;
    while (g * 2 <= want && g * 2 <= kChainMaxGroups && chain_waves(bytes, g * 2) >= nw) g *= 2;
    return g;
}

__global__ void k_chain_redo_count(const uint32_t* fault, uint32_t* redone, uint32_t* host_mirror)
{
    if (*fault != 0) *redone += 1;  // (launches of one context are ordered: one thread, no atomics)
    *host_mirror = *redone;
}

hipError_t launch_chain_redo_count(hipStream_t st, const uint32_t* fault, uint32_t* redone, uint32_t* host_mirror)
{
    hipLaunchKernelGGL(k_chain_redo_count, dim3(1), dim3(1), 0, st, fault, redone, host_mirror);
    return hipGetLastError();
}

hipError_t launch_pool_chain(hipStream_t st, const PoolArgs& pool, const ChainArgs& chain, int bytes)
{
    const int groups = chain.groups > 1 ? chain.groups : 1;
    int lanes = pool_chain_lanes(bytes, pool.stride_e);
    const int nchains = chain.nchains > 1 ? chain.nchains : 1;
    if (lanes < 2 || chain.npass < 1 || pool.bh < 2) return hipErrorInvalidValue;
    if (nchains == 1 ? pool.slot_mod <= chain.npass
                     : (groups > 1 || chain.chain_step <= chain.npass || chain.origin + (int64_t)nchains * chain.chain_step > pool.slot_mod))
        return hipErrorInvalidValue;  // (several chains: each has its slots to itself, none wraps around the ring, one workgroup per buffer)
    for (int k = 0; k < chain.pn; ++k)
        if (chain.w[k] % 8 != 0 || chain.nr[k] >= pool.bh) return hipErrorInvalidValue;
    const int nw = v3c::strips_for(pool.stride_e / 8);
    if ((int64_t)pool.slot_bytes * pool.slot_mod > (int64_t)0xfffffffe) return hipErrorInvalidValue;  // 32-bit offsets into the ring
    if (groups > 1) {
        if (groups != pool_chain_groups(bytes, pool.stride_e, groups) || !chain.flags || !chain.status || chain.slack < 0) return hipErrorInvalidValue;
        lanes = (bytes == 1 ? 2 : 1) * (chain_waves(bytes, groups) / nw);  // per workgroup (8-bit: two passes per set of nw waves)
    }
    const int rows = chain.rows > 0 && chain.rows < pool.bh ? chain.rows : pool.bh;
    const int pass_rounds = 1 + (rows - 1 + v3c::K - 1) / v3c::K;
    const int busy = lanes * groups * kChainLag + groups * (groups > 1 ? chain.slack : 0);  // rounds until a slot may take its next pass
    const int cycle = pass_rounds > busy ? pass_rounds : busy;
    if (groups == 1 && chain.npass < lanes) lanes = bytes == 1 ? (chain.npass + 1) & ~1 : chain.npass;  // a short chain: no idle waves at the barrier
    const int sets = bytes == 1 ? lanes / 2 : lanes;  // sets of nw waves (8-bit: a set carries two passes)
    const size_t lds = (size_t)sets * 2 * nw * 2 * v3c::GH * 8 * sizeof(unsigned);
    const dim3 grid(kBuffers * groups, nchains), block(sets * nw * 64);
    if (bytes == 4 && groups > 1)
        hipLaunchKernelGGL(k_smooth_f32_chain<true>, grid, block, lds, st, pool, chain, nw, lanes, pass_rounds, cycle);
    else if (bytes == 4)
        hipLaunchKernelGGL(k_smooth_f32_chain<false>, grid, block, lds, st, pool, chain, nw, lanes, pass_rounds, cycle);
    else if (bytes == 2 && groups > 1)
        hipLaunchKernelGGL(k_smooth_u16_chain<true>, grid, block, lds, st, pool, chain, nw, lanes, pass_rounds, cycle);
    else if (bytes == 2)
        hipLaunchKernelGGL(k_smooth_u16_chain<false>, grid, block, lds, st, pool, chain, nw, lanes, pass_rounds, cycle);
    else if (groups > 1)
        hipLaunchKernelGGL((k_smooth_u8_chain<true, kChain8Threads / 2>), grid, block, lds, st, pool, chain, nw, lanes, pass_rounds, cycle);
    else if ((int)block.x <= kChain8Threads / 2)
        hipLaunchKernelGGL((k_smooth_u8_chain<false, kChain8Threads / 2>), grid, block, lds, st, pool, chain, nw, lanes, pass_rounds, cycle);
    else
        hipLaunchKernelGGL((k_smooth_u8_chain<false, kChain8Threads>), grid, block, lds, st, pool, chain, nw, lanes, pass_rounds, cycle);
    return hipGetLastError();
}

hipError_t launch_pool_plane(hipStream_t st, const PlaneArgs& p, const PoolArgs& pool, int bytes,
                             double threshold, int nframes, int slot0)
{
    switch (bytes) {
    case 1: return launch_pool_plane_t<uint8_t>(st, p, pool, threshold, nframes, slot0);
    case 2: return launch_pool_plane_t<uint16_t>(st, p, pool, threshold, nframes, slot0);
    default: return launch_pool_plane_t<float>(st, p, pool, threshold, nframes, slot0);
    }
}

}  // namespace sn
