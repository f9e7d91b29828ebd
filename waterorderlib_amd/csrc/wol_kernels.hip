// wol_kernels.hip -- gfx950 kernels of libwol.so (included by wol_api.hip).
//
// Per batch of frames the q pipeline is
//   k_bin_count   waters -> cell id + arrival rank (int atomics on the per-frame cell counters)
//   k_cell_scan   per-frame exclusive scan of the counters -> cell_start[]
//   k_scatter     counting-sort scatter: sorted[f][start[cell]+rank] = (x, y, z, original index)
//   k_search_*    per water: walk the cells shell by shell, exact minimum-image arithmetic,
//                 register top-4, early exit once the 4th key is inside the radius the visited
//                 shells guarantee; fused epilogue: six angles, q, q/distance histograms in LDS
//   k_frame_stats per-frame mean / population variance of q (deterministic tree reduction)
// Nothing N^2-sized is ever stored (the reference materialises an N x N int32 mask,
// waterlib.f90:836).  k_brute_* are LDS-tiled all-pairs kernels: the f2py-compatible masks, frames
// with a non-periodic axis, and tiny inputs.
#include "wol_device.h"

#define WOL_BLOCK 256

// =============================================================================================
// binning
// =============================================================================================
template <typename T>
struct FramesView {
    const T* base;
    int64_t atom_stride;   // in elements of T
    int64_t frame_stride;  // in elements of T
    int n;
};

template <typename T>
__global__ void __launch_bounds__(WOL_BLOCK)
k_bin_count(FramesView<T> fv, const FrameGrid* __restrict__ grids, int* __restrict__ cell_count,
            int* __restrict__ cellid, int* __restrict__ rank) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * WOL_BLOCK + threadIdx.x;
    if (i >= fv.n) return;
    const FrameGrid& g = grids[f];
    const T* p = fv.base + (int64_t)f * fv.frame_stride + (int64_t)i * fv.atom_stride;
    const double x = (double)p[0], y = (double)p[1], z = (double)p[2];
    const int cx = wol_cell_coord(x, g.box.ib[0], g.nc[0], nullptr);
    const int cy = wol_cell_coord(y, g.box.ib[1], g.nc[1], nullptr);
    const int cz = wol_cell_coord(z, g.box.ib[2], g.nc[2], nullptr);
    const int c = (cz * g.nc[1] + cy) * g.nc[0] + cx;
    const int r = atomicAdd(&cell_count[g.cell_base + c], 1);
    cellid[(int64_t)f * fv.n + i] = c;
    rank[(int64_t)f * fv.n + i] = r;
}

// exclusive scan of cell_count[0..ncell) in place; cell_count[ncell] receives the total.  One block per frame goes
// through the table in tiles of 4096 (four consecutive entries per thread: the block's accesses are one contiguous
// stream), wave scans by shuffle, a running carry between tiles.
__global__ void __launch_bounds__(1024)
k_cell_scan(const FrameGrid* __restrict__ grids, int* __restrict__ cell_count) {
    __shared__ int wsum[16];
    const FrameGrid& g = grids[blockIdx.x];
    int* c = cell_count + g.cell_base;
    const int ncell = g.ncell;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 0;
    for (int t0 = 0; t0 < ncell; t0 += 4096) {
        const int i = t0 + 4 * tid;
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i + u < ncell) ? c[i + u] : 0;
        const int s = (v[0] + v[1]) + (v[2] + v[3]);
        const int inc = wol_wave_scan_incl(s);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int x = wsum[w]; total += x; if (w < wave) before += x; }
        int run = carry + before + inc - s;
#pragma unroll
        for (int u = 0; u < 4; ++u) { if (i + u < ncell) c[i + u] = run; run += v[u]; }
        carry += total;
        __syncthreads();
    }
    if (tid == 0) c[ncell] = carry;
}

// The same scan for big tables (a 500k-water frame binned for a short reach has ~0.5 M cells): every tile of 4096
// gets its own block (local scan + tile total), one block per frame scans the tile totals, a third launch adds them.
__global__ void __launch_bounds__(1024)
k_cell_scan_tiles(const FrameGrid* __restrict__ grids, int* __restrict__ cell_count, int* __restrict__ tile_tot, int tt_stride) {
    __shared__ int wsum[16];
    const FrameGrid& g = grids[blockIdx.y];
    int* c = cell_count + g.cell_base;
    const int ncell = g.ncell, t0 = blockIdx.x * 4096;
    if (t0 >= ncell) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = t0 + 4 * tid;
    int v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (i + u < ncell) ? c[i + u] : 0;
    const int s = (v[0] + v[1]) + (v[2] + v[3]);
    const int inc = wol_wave_scan_incl(s);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int x = wsum[w]; total += x; if (w < wave) before += x; }
    int run = before + inc - s;
#pragma unroll
    for (int u = 0; u < 4; ++u) { if (i + u < ncell) c[i + u] = run; run += v[u]; }
    if (tid == 0) tile_tot[(int64_t)blockIdx.y * tt_stride + blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024)
k_cell_scan_totals(const FrameGrid* __restrict__ grids, int* __restrict__ cell_count, int* __restrict__ tile_tot, int tt_stride) {
    __shared__ int wsum[16];
    const FrameGrid& g = grids[blockIdx.x];
    int* tt = tile_tot + (int64_t)blockIdx.x * tt_stride;
    const int ntile = (g.ncell + 4095) / 4096;   // <= 1024 tiles: 4 M cells
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v = tid < ntile ? tt[tid] : 0;
    const int inc = wol_wave_scan_incl(v);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int x = wsum[w]; total += x; if (w < wave) before += x; }
    if (tid < ntile) tt[tid] = before + inc - v;
    if (tid == 0) cell_count[g.cell_base + g.ncell] = total;
}

__global__ void __launch_bounds__(1024)
k_cell_scan_add(const FrameGrid* __restrict__ grids, int* __restrict__ cell_count, const int* __restrict__ tile_tot, int tt_stride) {
    const FrameGrid& g = grids[blockIdx.y];
    const int add = tile_tot[(int64_t)blockIdx.y * tt_stride + blockIdx.x];
    int* c = cell_count + g.cell_base;
    const int i = blockIdx.x * 4096 + 4 * threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (i + u < g.ncell) c[i + u] += add;
}

template <typename T>
__global__ void __launch_bounds__(WOL_BLOCK)
k_scatter(FramesView<T> fv, const FrameGrid* __restrict__ grids, const int* __restrict__ cell_start,
          const int* __restrict__ cellid, const int* __restrict__ rank, double4* __restrict__ sorted,
          float4* __restrict__ sorted32) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * WOL_BLOCK + threadIdx.x;
    if (i >= fv.n) return;
    const FrameGrid& g = grids[f];
    const T* p = fv.base + (int64_t)f * fv.frame_stride + (int64_t)i * fv.atom_stride;
    const int64_t o = (int64_t)f * fv.n;
    const int cell = cellid[o + i];
    const int dst = cell_start[g.cell_base + cell] + rank[o + i];
    double4 v;
    v.x = (double)p[0]; v.y = (double)p[1]; v.z = (double)p[2];
    v.w = __longlong_as_double((long long)i);
    sorted[o + dst] = v;
    if (sorted32) {
        // fp32 copy for the filtered search, already in key units: position inside the own cell (the
        // staging adds the cell's offset in the tile frame, so magnitudes -- and rounding -- stay small)
        const int nx = g.nc[0], ny = g.nc[1];
        const int cx = cell % nx, cy = (cell / nx) % ny, cz = cell / (nx * ny);
        double u;
        float4 w;
        u = v.x * g.box.ib[0]; u -= floor(u); w.x = (float)((u * (double)nx - (double)cx) * g.cw[0] * (double)g.scale);
        u = v.y * g.box.ib[1]; u -= floor(u); w.y = (float)((u * (double)ny - (double)cy) * g.cw[1] * (double)g.scale);
        u = v.z * g.box.ib[2]; u -= floor(u); w.z = (float)((u * (double)g.nc[2] - (double)cz) * g.cw[2] * (double)g.scale);
        w.w = (float)cx;  // cell number along x (k_search_rows builds tile coordinates from it)
        sorted32[o + dst] = w;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused counting sort of ONE frame per 1024-thread block, cell counters in LDS (dynamic, ncell+1
// ints, up to the CU's 160 KiB): count with LDS atomics (the global-atomic form above is bound by
// the ~33 G/s rate of memory-side atomics), block scan in place, scatter.  Used when a batch has
// enough frames to fill the chip with one block each and the cell table fits.
// ---------------------------------------------------------------------------------------------
#ifndef WOL_SORT_BLOCK
#define WOL_SORT_BLOCK 1024
#endif
template <typename T>
__global__ void __launch_bounds__(WOL_SORT_BLOCK)
k_sort_frame_lds(FramesView<T> fv, const FrameGrid* __restrict__ grids, int* __restrict__ cell_start,
                 int* __restrict__ rank, double4* __restrict__ sorted, float4* __restrict__ sorted32) {
    extern __shared__ int cnt[];          // [ncell + 1]
    __shared__ int part[WOL_SORT_BLOCK];
    const int f = blockIdx.x;
    const FrameGrid& g = grids[f];
    const int ncell = g.ncell, tid = threadIdx.x;
    const int nx = g.nc[0], ny = g.nc[1], nz = g.nc[2];
    const T* __restrict__ base = fv.base + (int64_t)f * fv.frame_stride;
    int* __restrict__ rk = rank + (int64_t)f * fv.n;
    for (int c = tid; c <= ncell; c += WOL_SORT_BLOCK) cnt[c] = 0;
    __syncthreads();
    // ---- pass 1: cell of every water, arrival rank inside the cell (8 waters per thread in flight:
    //      one block owns the whole frame, so memory-level parallelism has to come from unrolling) ----
    constexpr int U = 8;
    for (int i0 = tid; i0 < fv.n; i0 += WOL_SORT_BLOCK * U) {
        double x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * WOL_SORT_BLOCK;
            if (i < fv.n) {
                const T* p = base + (int64_t)i * fv.atom_stride;
                x[u] = (double)p[0]; y[u] = (double)p[1]; z[u] = (double)p[2];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * WOL_SORT_BLOCK;
            if (i < fv.n) {
                const int cx = wol_cell_coord(x[u], g.box.ib[0], nx, nullptr);
                const int cy = wol_cell_coord(y[u], g.box.ib[1], ny, nullptr);
                const int cz = wol_cell_coord(z[u], g.box.ib[2], nz, nullptr);
                rk[i] = atomicAdd(&cnt[(cz * ny + cy) * nx + cx], 1);
            }
        }
    }
    __syncthreads();
    // ---- exclusive scan of the counters in place ----
    const int per = (ncell + WOL_SORT_BLOCK - 1) / WOL_SORT_BLOCK;
    const int b = min(tid * per, ncell), e = min(b + per, ncell);
    int s = 0;
    for (int k = b; k < e; ++k) s += cnt[k];
    wol_block_scan_incl(s, part);   // part[tid] = inclusive prefix over the block's threads
    int run = part[tid] - s;
    for (int k = b; k < e; ++k) { const int v = cnt[k]; cnt[k] = run; run += v; }
    if (tid == WOL_SORT_BLOCK - 1) cnt[ncell] = part[WOL_SORT_BLOCK - 1];
    __syncthreads();
    int* __restrict__ cs = cell_start + g.cell_base;
    for (int c = tid; c <= ncell; c += WOL_SORT_BLOCK) cs[c] = cnt[c];
    // ---- pass 2: scatter (coordinates re-read; they are still in L2 / Infinity Cache) ----
    double4* __restrict__ so = sorted + (int64_t)f * fv.n;
    float4* __restrict__ so32 = sorted32 ? sorted32 + (int64_t)f * fv.n : nullptr;
    for (int i0 = tid; i0 < fv.n; i0 += WOL_SORT_BLOCK * U) {
        double x[U], y[U], z[U];
        int r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * WOL_SORT_BLOCK;
            if (i < fv.n) {
                const T* p = base + (int64_t)i * fv.atom_stride;
                x[u] = (double)p[0]; y[u] = (double)p[1]; z[u] = (double)p[2];
                r[u] = rk[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * WOL_SORT_BLOCK;
            if (i >= fv.n) continue;
            double4 v;
            v.x = x[u]; v.y = y[u]; v.z = z[u];
            v.w = __longlong_as_double((long long)i);
            double ux = v.x * g.box.ib[0]; ux -= floor(ux);
            double uy = v.y * g.box.ib[1]; uy -= floor(uy);
            double uz = v.z * g.box.ib[2]; uz -= floor(uz);
            // same arithmetic as wol_cell_coord
            int cx = (int)(ux * (double)nx); cx = cx >= nx ? nx - 1 : (cx < 0 ? 0 : cx);
            int cy = (int)(uy * (double)ny); cy = cy >= ny ? ny - 1 : (cy < 0 ? 0 : cy);
            int cz = (int)(uz * (double)nz); cz = cz >= nz ? nz - 1 : (cz < 0 ? 0 : cz);
            const int dst = cnt[(cz * ny + cy) * nx + cx] + r[u];
            so[dst] = v;
            if (so32) {
                float4 w;
                w.x = (float)((ux * (double)nx - (double)cx) * g.cw[0] * (double)g.scale);
                w.y = (float)((uy * (double)ny - (double)cy) * g.cw[1] * (double)g.scale);
                w.z = (float)((uz * (double)nz - (double)cz) * g.cw[2] * (double)g.scale);
                w.w = (float)cx;  // cell number along x (k_search_rows builds tile coordinates from it)
                so32[dst] = w;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming two-level sort for large batches.  A random-order frame makes a direct scatter into
// the cell-sorted arrays a storm of partial-line HBM writes; here every global access is a stream:
//   level 1 (one block per frame): histogram of the (z, y) cell rows in LDS, then every water is
//           appended to its row's bucket in tmp[] -- ~1000 write streams whose open lines stay in L2;
//   level 2 (one wave per row): the bucket (~100 waters, second read served by L1) is counting-sorted
//           by cx with per-wave LDS counters straight into the final arrays; consecutive rows are
//           consecutive in the output, so those writes stream too.
// ---------------------------------------------------------------------------------------------
#ifndef WOL_XSUB
#define WOL_XSUB 4            // fine x bins per cell (see k_sort_in_row)
#endif
// Index records ("IDX" forms of the sort, float64 frames of bare O rows on their way to the row-merged q pipeline): the first
// level already knows everything the proposal kernel wants of a water -- its position inside its cell as fp32, its cell along
// x and fine bin -- so it writes THAT (12 bytes) plus a key word = original index | cell << 20 | fine bin << 27 instead of a
// 32-byte fp64 record; the second level orders these 16-byte records and writes the fp32 array and one int per water (the
// original index by sorted slot).  No fp64 copy of the frame is made: the certification and the exact search read the 24-byte
// rows of the SOURCE frame by original index (an O-only frame of 100 000 waters is 2.4 MB: it stays in the L2 of the XCD
// that works on it).  76 instead of 136 bytes per water through the sort.
#define WOL_IDX_BITS 20
__device__ __forceinline__ unsigned wol_idx_key(int orig, int cx, int sub) { return (unsigned)orig | ((unsigned)cx << WOL_IDX_BITS) | ((unsigned)sub << (WOL_IDX_BITS + 7)); }

#define WOL_L1_MAXROWS 4080   // nz * ny rows counted in LDS by level 1 (three tables + the chunk staging within 160 KB)
#define WOL_L2_MAXNX 128      // cells of a row handled by one wave of level 2
#ifndef WOL_L1_CHUNK
#define WOL_L1_CHUNK 3            // waters per thread and chunk of level 1 (chunk = 3 * 1024 waters: 96 KB of staging)
#endif
#define WOL_L1_BLOCK_PIPE 512     // level-1 block when the sort runs beside the proposal kernel of another batch ...
#define WOL_L1_CHUNK_PIPE 2       // ... and its waters per thread and chunk: 8 waves, 37 KB of staging
#define WOL_L1_CHUNK_BIG 2        // ... for frames with more rows than WOL_L1_MAXROWS (tables up to WOL_L1_MAXROWS_BIG rows)
#define WOL_L1_MAXROWS_BIG 6800
// dynamic LDS of the level-1 kernels for a frame with nrow cell rows and chunks of ch * 1024 waters
__host__ __device__ inline size_t wol_sort_rows_lds(int nrow, int ch, int block = WOL_SORT_BLOCK, size_t rec = sizeof(double4)) {
    return (size_t)ch * block * (rec + sizeof(int)) + 3 * sizeof(int) * (size_t)(nrow + 1);
}

// The scatter of level 1 for waters [i_begin, i_end) of a frame, cnt[] holding the row cursors in tmp[]:
// a lane's 32-byte store to its water's row bucket is a transaction of its own (64 per wave instruction) -- the
// scatter alone cost 0.23 ms of 0.61.  So a chunk of CH * 1024 waters is first ordered by row inside LDS (returning
// atomic = rank inside the row, a scan over the rows) and then written out in that order: a row's waters of the
// chunk are consecutive lanes writing consecutive addresses.  The next chunk's coordinates are fetched while the
// current one goes through LDS.  icnt[] must be zero on entry (and is on exit).
template <typename T, int CH, int SB = WOL_SORT_BLOCK, class REC = double4>
__device__ __forceinline__ void wol_sort_rows_chunks(const T* __restrict__ base, int64_t atom_stride, int i_begin, int i_end,
                                                     const FrameGrid& g, int* cnt, int* icnt, int* delta, REC* stage,
                                                     int* dest, int* wtot, REC* __restrict__ out) {
    constexpr int C = CH * SB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ny = g.nc[1], nz = g.nc[2], nrow = ny * nz;
    const int per = (nrow + SB - 1) / SB;
    const int sb = min(tid * per, nrow), se = min(sb + per, nrow);
    double nx_[CH], ny_[CH], nz_[CH];
    if (i_begin >= i_end) return;
#pragma unroll
    for (int u = 0; u < CH; ++u) {   // (clamped, unconditional loads: see wol_sort_rows_count)
        const int i = min(i_begin + u * SB + tid, i_end - 1);
        const T* p = base + (int64_t)i * atom_stride;
        nx_[u] = (double)p[0]; ny_[u] = (double)p[1]; nz_[u] = (double)p[2];
    }
    for (int c0 = i_begin; c0 < i_end; c0 += C) {
        const int cn = min(C, i_end - c0);
        // (1) the chunk's waters: row and rank inside the row
        double x[CH], y[CH], z[CH];
        int row[CH], lr[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) { x[u] = nx_[u]; y[u] = ny_[u]; z[u] = nz_[u]; }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int i = min(c0 + C + u * SB + tid, i_end - 1);
            const T* p = base + (int64_t)i * atom_stride;
            nx_[u] = (double)p[0]; ny_[u] = (double)p[1]; nz_[u] = (double)p[2];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int i = c0 + u * SB + tid;
            row[u] = -1;
            if (i < i_end) {
                row[u] = wol_cell_coord(z[u], g.box.ib[2], nz, nullptr) * ny + wol_cell_coord(y[u], g.box.ib[1], ny, nullptr);
                lr[u] = atomicAdd(&icnt[row[u]], 1);
            }
        }
        __syncthreads();
        // (2) starts of the rows inside the chunk (thread t owns rows [sb, se)); cursors advance by the chunk's counts
        int s2 = 0;
        for (int k = sb; k < se; ++k) s2 += icnt[k];
        const int inc = wol_wave_scan_incl(s2);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int before = 0;
        for (int w2 = 0; w2 < wave; ++w2) before += wtot[w2];
        int st = before + inc - s2;
        for (int k = sb; k < se; ++k) {
            const int v = icnt[k];
            icnt[k] = st;               // start of row k in the staged chunk
            delta[k] = cnt[k] - st;     // slot in tmp[] = staged position + delta
            cnt[k] += v;
            st += v;
        }
        __syncthreads();
        // (3) into the staging area in row order
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (row[u] >= 0) {
                const int k = icnt[row[u]] + lr[u];
                stage[k] = wol_rec<REC>::make(x[u], y[u], z[u], c0 + u * SB + tid);
                dest[k] = k + delta[row[u]];
            }
        __syncthreads();
        // (4) out, in staged order; the owners clear the chunk counters for the next round
#pragma unroll
        for (int u = 0; u < CH; ++u) {   // (a fixed number of stores per chunk: the wait for the next chunk's
            const int k = tid + u * SB;   //  coordinates, fetched before them, need not drain them)
            if (k < cn) out[dest[k]] = stage[k];
        }
        for (int k = sb; k < se; ++k) icnt[k] = 0;
        __syncthreads();
    }
}

// histogram of the (z, y) cell rows of waters [i_begin, i_end) into cnt[] (LDS, zeroed by the caller)
template <typename T, int SB = WOL_SORT_BLOCK>
__device__ __forceinline__ void wol_sort_rows_count(const T* __restrict__ base, int64_t atom_stride, int i_begin, int i_end,
                                                    const FrameGrid& g, int* cnt) {
#ifndef WOL_SORT_U
#define WOL_SORT_U 8
#endif
    constexpr int U = WOL_SORT_U;
    const int tid = threadIdx.x, ny = g.nc[1], nz = g.nc[2];
    for (int i0 = i_begin + tid; i0 < i_end; i0 += SB * U) {
        double y[U], z[U];
        // (loads are unconditional -- the index is clamped -- so the compiler issues all U before the first wait: behind
        //  a per-element `if` it waited for every load on its own, one memory round trip after the other)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(i0 + u * SB, i_end - 1);
            const T* p = base + (int64_t)i * atom_stride;
            y[u] = (double)p[1]; z[u] = (double)p[2];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + u * SB < i_end)
                atomicAdd(&cnt[wol_cell_coord(z[u], g.box.ib[2], nz, nullptr) * ny + wol_cell_coord(y[u], g.box.ib[1], ny, nullptr)], 1);
    }
}

// one block per frame: count, scan, scatter (the batch has enough frames to fill the chip)
template <typename T, int SB, int CH, class REC = double4>
__global__ void __launch_bounds__(SB)
k_sort_rows(FramesView<T> fv, const FrameGrid* __restrict__ grids, int* __restrict__ row_off, int ro_stride,
            REC* __restrict__ tmp, int nrow_cap) {
    extern __shared__ double4 lds_sort_rows[];
    constexpr int C = CH * SB;
    REC* stage = reinterpret_cast<REC*>(lds_sort_rows);             // [C] the chunk in row order
    int* dest = reinterpret_cast<int*>(stage + C);                   // [C] its slots in tmp[]
    int* cnt = dest + C;                                             // [nrow + 1] row cursors in tmp[]
    int* icnt = cnt + (nrow_cap + 1);                                // [nrow + 1] the chunk's waters per row, then starts
    int* delta = icnt + (nrow_cap + 1);                              // [nrow + 1] cursor - start of the row in this chunk
    __shared__ int part[SB];
    __shared__ int wtot[SB / 64];
    const int f = blockIdx.x, tid = threadIdx.x;
    const FrameGrid& g = grids[f];
    const int nrow = g.nc[1] * g.nc[2];
    const T* __restrict__ base = fv.base + (int64_t)f * fv.frame_stride;
    for (int c = tid; c <= nrow; c += SB) { cnt[c] = 0; icnt[c] = 0; }
    __syncthreads();
    wol_sort_rows_count<T, SB>(base, fv.atom_stride, 0, fv.n, g, cnt);
    __syncthreads();
    // exclusive scan of the row counters in place
    const int per = (nrow + SB - 1) / SB;
    const int sb = min(tid * per, nrow), se = min(sb + per, nrow);
    int s = 0;
    for (int k = sb; k < se; ++k) s += cnt[k];
    wol_block_scan_incl(s, part);   // part[tid] = inclusive prefix over the block's threads
    int* ro = row_off + (int64_t)f * ro_stride;
    int run = part[tid] - s;
    for (int k = sb; k < se; ++k) { const int v = cnt[k]; cnt[k] = run; ro[k] = run; run += v; }
    if (tid == SB - 1) ro[nrow] = part[SB - 1];
    __syncthreads();
    wol_sort_rows_chunks<T, CH, SB, REC>(base, fv.atom_stride, 0, fv.n, g, cnt, icnt, delta, stage, dest, wtot,
                                         tmp + (int64_t)f * fv.n);
}

// ---------------------------------------------------------------------------------------------
// First level, single-pass form ("segments"): one block per frame.  The frame is cut into chunks of CH * SB waters; a
// chunk is ordered by (z, y) cell row inside LDS (returning atomic = rank inside the row, a scan over the rows) and
// written out AS IT LIES THERE -- one contiguous 32-byte-record stream per block, no read of the frame beforehand to size
// row buckets -- together with the start of every row inside the chunk (tab[chunk][row], 16 bit).  The second level's
// wave then collects its row from the frame's chunks (~3 waters of the row in each).  The row totals, summed as the
// chunks go by, give the rows' places in the final arrays (row_off), as before.
// ---------------------------------------------------------------------------------------------
// Big frames in small batches: P blocks per frame (grid (P, F)), block p taking chunks p, p + P, ...; the blocks' row totals
// go to cntb[f][p][row] and k_sort_rows_scan turns them into the rows' places (with P == 1 the block does that itself).
template <typename T, int SB, int CH, class REC = double4, bool IDX = false>   // (IDX: REC = float4, see wol_idx_key)
__global__ void __launch_bounds__(SB)
k_sort_rows_seg(FramesView<T> fv, const FrameGrid* __restrict__ grids, int* __restrict__ row_off, int ro_stride,
                REC* __restrict__ tmp, unsigned short* __restrict__ tab, int tab_rows, int nchunk_cap, int nrow_cap,
                int P, int* __restrict__ cntb, int cnt_stride) {
    extern __shared__ double4 lds_sort_rows[];
    constexpr int C = CH * SB;
    REC* stage = reinterpret_cast<REC*>(lds_sort_rows);              // [C] the chunk in row order
    int* tot = reinterpret_cast<int*>(stage + C);                    // [nrow + 1] waters of the row so far
    int* icnt = tot + (nrow_cap + 1);                                // [nrow + 1] the chunk's waters per row, then starts
    __shared__ int part[SB];
    __shared__ int wtot[SB / 64];
    const int p = blockIdx.x, f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FrameGrid& g = grids[f];
    const int ny = g.nc[1], nz = g.nc[2], nrow = ny * nz;
    const T* __restrict__ base = fv.base + (int64_t)f * fv.frame_stride;
    REC* __restrict__ out = tmp + (int64_t)f * fv.n;
    unsigned short* __restrict__ tb = tab + (int64_t)f * tab_rows * nchunk_cap;
    const int per = (nrow + SB - 1) / SB;
    const int sb = min(tid * per, nrow), se = min(sb + per, nrow);
    for (int c = tid; c <= nrow; c += SB) { tot[c] = 0; icnt[c] = 0; }
    const int i_end = fv.n;
    double nx_[CH], ny_[CH], nz_[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {   // (clamped, unconditional loads: see wol_sort_rows_count)
        const int i = min(p * C + u * SB + tid, i_end - 1);
        const T* q = base + (int64_t)i * fv.atom_stride;
        nx_[u] = (double)q[0]; ny_[u] = (double)q[1]; nz_[u] = (double)q[2];
    }
    __syncthreads();
    for (int chunk = p; (int64_t)chunk * C < i_end; chunk += P) {
        const int c0 = chunk * C;
        const int cn = min(C, i_end - c0);
        double x[CH], y[CH], z[CH];
        int row[CH], lr[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) { x[u] = nx_[u]; y[u] = ny_[u]; z[u] = nz_[u]; }
#pragma unroll
        for (int u = 0; u < CH; ++u) {   // the block's next chunk's coordinates travel while this one goes through LDS
            const int i = (int)min((int64_t)c0 + (int64_t)P * C + u * SB + tid, (int64_t)i_end - 1);
            const T* q = base + (int64_t)i * fv.atom_stride;
            nx_[u] = (double)q[0]; ny_[u] = (double)q[1]; nz_[u] = (double)q[2];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int i = c0 + u * SB + tid;
            row[u] = -1;
            if (i < i_end) {
                row[u] = wol_cell_coord(z[u], g.box.ib[2], nz, nullptr) * ny + wol_cell_coord(y[u], g.box.ib[1], ny, nullptr);
                lr[u] = atomicAdd(&icnt[row[u]], 1);
            }
        }
        __syncthreads();
        // starts of the rows inside the chunk (thread t owns rows [sb, se)): into LDS for the placement, into the table
        // for the second level; the row totals advance
        int s2 = 0;
        for (int k = sb; k < se; ++k) s2 += icnt[k];
        const int inc = wol_wave_scan_incl(s2);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int before = 0;
        for (int w2 = 0; w2 < wave; ++w2) before += wtot[w2];
        int st = before + inc - s2;
        for (int k = sb; k < se; ++k) {
            const int v = icnt[k];
            icnt[k] = st;
            tb[(int64_t)chunk * tab_rows + k] = (unsigned short)st;     // chunk-major: the owners' stores are one 2-KB run
            tot[k] += v;
            st += v;
        }
        if (tid == SB - 1) tb[(int64_t)chunk * tab_rows + nrow] = (unsigned short)cn;   // (row nrow: the end of the last row)
        __syncthreads();
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (row[u] >= 0) {
                if constexpr (IDX) {
                    // the fp32 record of k_sort_in_row, statement for statement (cy, cz are the row's, as there)
                    const int nxc = g.nc[0];
                    const int cyr = row[u] % ny, czr = row[u] / ny;
                    double ux = x[u] * g.box.ib[0]; ux -= floor(ux);
                    double uy = y[u] * g.box.ib[1]; uy -= floor(uy);
                    double uz = z[u] * g.box.ib[2]; uz -= floor(uz);
                    int cx = (int)(ux * (double)nxc); cx = cx >= nxc ? nxc - 1 : (cx < 0 ? 0 : cx);
                    const double sx = g.cw[0] * (double)g.scale, sy = g.cw[1] * (double)g.scale, sz = g.cw[2] * (double)g.scale;
                    float4 w;
                    w.x = (float)((ux * (double)nxc - (double)cx) * sx);
                    w.y = (float)((uy * (double)ny - (double)cyr) * sy);
                    w.z = (float)((uz * (double)nz - (double)czr) * sz);
                    const float inv_wf = (float)WOL_XSUB / (float)(g.cw[0] * (double)g.scale);
                    const int sub = min(WOL_XSUB - 1, (int)(w.x * inv_wf));
                    w.w = __uint_as_float(wol_idx_key(c0 + u * SB + tid, cx, sub));
                    stage[icnt[row[u]] + lr[u]] = w;
                } else {
                    stage[icnt[row[u]] + lr[u]] = wol_rec<REC>::make(x[u], y[u], z[u], c0 + u * SB + tid);
                }
            }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int k = tid + u * SB;
            if (k < cn) out[c0 + k] = stage[k];
        }
        for (int k = sb; k < se; ++k) icnt[k] = 0;
        __syncthreads();
    }
    if (P > 1) {   // this block's share of the row totals; k_sort_rows_scan adds the shares up and scans them
        int* __restrict__ o = cntb + ((int64_t)f * P + p) * cnt_stride;
        for (int k = sb; k < se; ++k) o[k] = tot[k];
        return;
    }
    // exclusive scan of the row totals: the rows' places in the final arrays
    int s = 0;
    for (int k = sb; k < se; ++k) s += tot[k];
    wol_block_scan_incl(s, part);   // part[tid] = inclusive prefix over the block's threads
    int* ro = row_off + (int64_t)f * ro_stride;
    int run = part[tid] - s;
    for (int k = sb; k < se; ++k) { ro[k] = run; run += tot[k]; }
    if (tid == SB - 1) ro[nrow] = part[SB - 1];
}

// ---- the same with every frame split over P blocks (big frames in small batches): three launches ----
// (a) block (p, f): rows histogram of its share of the frame -> cntb[f][p][row]
template <typename T>
__global__ void __launch_bounds__(WOL_SORT_BLOCK)
k_sort_rows_count(FramesView<T> fv, const FrameGrid* __restrict__ grids, int* __restrict__ cntb, int stride, int P) {
    extern __shared__ int lds_rows_count[];
    const int p = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const FrameGrid& g = grids[f];
    const int nrow = g.nc[1] * g.nc[2];
    for (int c = tid; c < nrow; c += WOL_SORT_BLOCK) lds_rows_count[c] = 0;
    __syncthreads();
    const int per = (fv.n + P - 1) / P;
    wol_sort_rows_count<T>(fv.base + (int64_t)f * fv.frame_stride, fv.atom_stride, min(p * per, fv.n), min((p + 1) * per, fv.n), g,
                           lds_rows_count);
    __syncthreads();
    int* o = cntb + ((int64_t)f * P + p) * stride;
    for (int c = tid; c < nrow; c += WOL_SORT_BLOCK) o[c] = lds_rows_count[c];
}

// (b) block f: row offsets of the frame and, in place of the counts, every block's first slot in every row
__global__ void __launch_bounds__(WOL_SORT_BLOCK)
k_sort_rows_scan(const FrameGrid* __restrict__ grids, int* __restrict__ cntb, int stride, int P, int* __restrict__ row_off,
                 int ro_stride) {
    __shared__ int part[WOL_SORT_BLOCK];
    const int f = blockIdx.x, tid = threadIdx.x;
    const FrameGrid& g = grids[f];
    const int nrow = g.nc[1] * g.nc[2];
    const int per = (nrow + WOL_SORT_BLOCK - 1) / WOL_SORT_BLOCK;
    const int sb = min(tid * per, nrow), se = min(sb + per, nrow);
    int* cb = cntb + (int64_t)f * P * stride;
    int s = 0;
    for (int k = sb; k < se; ++k)
        for (int p = 0; p < P; ++p) s += cb[(int64_t)p * stride + k];
    wol_block_scan_incl(s, part);   // part[tid] = inclusive prefix over the block's threads
    int* ro = row_off + (int64_t)f * ro_stride;
    int run = part[tid] - s;
    for (int k = sb; k < se; ++k) {
        ro[k] = run;
        for (int p = 0; p < P; ++p) { const int v = cb[(int64_t)p * stride + k]; cb[(int64_t)p * stride + k] = run; run += v; }
    }
    if (tid == WOL_SORT_BLOCK - 1) ro[nrow] = part[WOL_SORT_BLOCK - 1];
}

// (c) block (p, f): its share of the frame into the row buckets, from its cursors
template <typename T, int CH, class REC = double4>
__global__ void __launch_bounds__(WOL_SORT_BLOCK)
k_sort_rows_scatter(FramesView<T> fv, const FrameGrid* __restrict__ grids, const int* __restrict__ cntb, int stride, int P,
                    REC* __restrict__ tmp, int nrow_cap) {
    extern __shared__ double4 lds_sort_rows[];
    constexpr int C = CH * WOL_SORT_BLOCK;
    REC* stage = reinterpret_cast<REC*>(lds_sort_rows);
    int* dest = reinterpret_cast<int*>(stage + C);
    int* cnt = dest + C;
    int* icnt = cnt + (nrow_cap + 1);
    int* delta = icnt + (nrow_cap + 1);
    __shared__ int wtot[WOL_SORT_BLOCK / 64];
    const int p = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const FrameGrid& g = grids[f];
    const int nrow = g.nc[1] * g.nc[2];
    const int* cur = cntb + ((int64_t)f * P + p) * stride;
    for (int c = tid; c < nrow; c += WOL_SORT_BLOCK) { cnt[c] = cur[c]; icnt[c] = 0; }
    __syncthreads();
    const int per = (fv.n + P - 1) / P;
    wol_sort_rows_chunks<T, CH, WOL_SORT_BLOCK, REC>(fv.base + (int64_t)f * fv.frame_stride, fv.atom_stride, min(p * per, fv.n),
                                                     min((p + 1) * per, fv.n), g, cnt, icnt, delta, stage, dest, wtot, tmp + (int64_t)f * fv.n);
}

// SLOTS waters per lane of a row are sorted through registers and LDS (rows of up to 64 * SLOTS waters; longer
// rows take the direct form): 2 for the usual ~100-water rows, 4 for the long rows of big boxes.
//
// The sort key is the FINE x bin (WOL_XSUB bins per cell, the bins of the proposal kernel's windows): every row of the
// cell-sorted arrays is bin-sorted, each water's fp32 record carries its cell, sub-bin and rank inside the bin
// (wol_pack_cell_tag), and cnt4[cell] holds the four sub-bin populations of the cell as bytes (saturating at 255; the
// proposal kernel checks them against the cell population).  k_search_rows then places a staged water with table
// look-ups only -- no counting pass, no atomics, one trip to memory per tile (wol_search_rows.hip).
#ifndef WOL_XSUB
#define WOL_XSUB 4            // fine x bins per cell
#endif
static_assert(WOL_XSUB == 4, "cnt4 packs the sub-bin populations of a cell into the four bytes of a word");

// tag word of a sorted32 record (stored in .w as raw bits): cell number along x, sub-bin, rank inside the fine bin
__device__ __forceinline__ float wol_pack_cell_tag(int cx, int sub, int rank) {
    return __uint_as_float((unsigned)cx | ((unsigned)sub << 8) | ((unsigned)rank << 10));
}

#define WOL_SEG_MAXCHUNK 256  // chunks of a frame the segment form of the first level may write (a power of two)

// where the rows' waters lie when the first level wrote segments (k_sort_rows_seg): chunk c of the frame holds the
// waters [tab[c][row], tab[c][row + 1]) of every row, chunks are `chunk_waters` apart; tab == null: rows are contiguous
struct RowPieces {
    const unsigned short* tab;
    int tab_rows, nchunk_cap, chunk_waters;
    int ny_cap;   // rows of a cell layer the block's LDS copy of the table has room for (+ 1)
};

template <bool IDX, class REC> struct wol_sorted_out { typedef REC type; };
template <class REC> struct wol_sorted_out<true, REC> { typedef int type; };   // index records: the original index by sorted slot

template <int SLOTS, class REC = double4, bool IDX = false>
__global__ void __launch_bounds__(WOL_BLOCK)
k_sort_in_row(const FrameGrid* __restrict__ grids, const int* __restrict__ row_off, int ro_stride,
              const REC* __restrict__ tmp, int n, int* __restrict__ cell_start,
              typename wol_sorted_out<IDX, REC>::type* __restrict__ sorted, float4* __restrict__ sorted32, unsigned* __restrict__ cnt4,
              int nb_cap, RowPieces rp) {
    typedef typename wol_sorted_out<IDX, REC>::type OUT;
    extern __shared__ int lds_row_bins[];   // per wave: cnt[nb_cap] (per fine bin: population, then first slot in the row) and
                                            // cur[nb_cap] (long rows: running cursors); nb_cap >= nx * WOL_XSUB of every frame
    constexpr int WOL_L2_CAP = 64 * SLOTS;
    __shared__ OUT st64_all[(WOL_BLOCK / 64) * WOL_L2_CAP];       // the row in bin order, written out linearly:
    __shared__ float4 st32_all[(WOL_BLOCK / 64) * WOL_L2_CAP];    // full-line stores instead of a 32-byte scatter
    const int f = blockIdx.y, cz = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const FrameGrid& g = grids[f];
    const int nx = g.nc[0], ny = g.nc[1], nz = g.nc[2];
    if (cz >= nz || g.maxshell < 0 || nx * WOL_XSUB > nb_cap) return;   // (frames of the all-pairs path have no grid)
    int* cnt = lds_row_bins + wave * 2 * nb_cap;
    int* cur = cnt + nb_cap;
    OUT* st64 = st64_all + wave * WOL_L2_CAP;
    float4* st32 = st32_all + wave * WOL_L2_CAP;
    const int* ro = row_off + (int64_t)f * ro_stride;
    const REC* __restrict__ in = tmp + (int64_t)f * n;
    const unsigned short* __restrict__ tb = rp.tab ? rp.tab + (int64_t)f * rp.tab_rows * rp.nchunk_cap : nullptr;
    const int nchunk = rp.tab ? (n + rp.chunk_waters - 1) / rp.chunk_waters : 0;
    // segment form, per wave (dynamic LDS behind the bin counters): waters of the row in earlier chunks, and where water i of
    // the row lies, minus i
    int* cpre = lds_row_bins + (WOL_BLOCK / 64) * 2 * nb_cap + wave * (2 * rp.nchunk_cap + 1);
    int* csrc = cpre + rp.nchunk_cap + 1;
    // the block's rows are the ny rows of cell layer cz: their table entries (and the next row's) are ny + 1 consecutive
    // 16-bit values per chunk -- fetched once, chunk by chunk, into LDS behind the waves' bin counters
    unsigned short* tabl = reinterpret_cast<unsigned short*>(lds_row_bins + (WOL_BLOCK / 64) * (2 * nb_cap + (rp.tab ? 2 * rp.nchunk_cap + 1 : 0)));
    if (tb) {
        const int w = ny + 1;
        for (int idx = threadIdx.x; idx < nchunk * w; idx += WOL_BLOCK) {
            const int c = idx / w, j = idx - c * w;
            tabl[idx] = tb[(int64_t)c * rp.tab_rows + cz * ny + j];
        }
        __syncthreads();
    }
    OUT* __restrict__ so = sorted + (int64_t)f * n;
    float4* __restrict__ so32 = sorted32 ? sorted32 + (int64_t)f * n : nullptr;
    const double sx = g.cw[0] * (double)g.scale, sy = g.cw[1] * (double)g.scale, sz = g.cw[2] * (double)g.scale;
    // the proposal kernel's own expression for the fine bin of a staged coordinate (wol_search_rows.hip)
    const float inv_wf = (float)WOL_XSUB / (float)(g.cw[0] * (double)g.scale);
    const int nb = nx * WOL_XSUB;                    // fine bins of a row
    const int cpl = (nx + 63) >> 6;                  // cells per lane in the scan (1 or 2: nx <= 128)
    for (int cy = wave; cy < ny; cy += WOL_BLOCK / 64) {  // one wave per row, no block-level barriers
        const int row = cz * ny + cy;
        const int b = ro[row], e = ro[row + 1];
        if (tb) {   // the row's pieces: a lane per chunk, 64 chunks at a time
            int run = 0;
            for (int c0 = 0; c0 < nchunk; c0 += 64) {
                const int c = c0 + lane;
                const int t0 = c < nchunk ? (int)tabl[c * (ny + 1) + cy] : 0;
                const int len = c < nchunk ? (int)tabl[c * (ny + 1) + cy + 1] - t0 : 0;
                const int inc = wol_wave_scan_incl(len);
                if (c < nchunk) {
                    cpre[c] = run + inc - len;
                    csrc[c] = c * rp.chunk_waters + t0 - (run + inc - len);
                }
                run += __builtin_amdgcn_readlane(inc, 63);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // water i of the row (0 <= i < e - b) in the first level's output
        auto src_of = [&](int i) -> int {
            if (!tb) return b + i;
            int c = 0;
#pragma unroll
            for (int step = WOL_SEG_MAXCHUNK / 2; step >= 1; step >>= 1)
                if (c + step < nchunk && cpre[c + step] <= i) c += step;
            return csrc[c] + i;
        };
        int* __restrict__ cs = cell_start + g.cell_base + (int64_t)row * nx;
        unsigned* __restrict__ c4 = cnt4 ? cnt4 + g.cell_base + (int64_t)row * nx : nullptr;
        for (int c = lane; c < nb; c += 64) cnt[c] = 0;
        __builtin_amdgcn_wave_barrier();
        const bool fits = e - b <= WOL_L2_CAP && e > b;
        REC r[SLOTS];
        double4 v[SLOTS];
        float4 w[SLOTS];
        int bins[SLOTS], rank[SLOTS];
        if (fits) {
            // ---- the usual row: read once (the waters stay in registers), rank = returning LDS atomic ----
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) r[u] = in[src_of(min(lane + 64 * u, e - b - 1))];   // all in flight together (e > b here)
            if constexpr (IDX) {
#pragma unroll
                for (int u = 0; u < SLOTS; ++u) {
                    const int i = b + lane + 64 * u;
                    bins[u] = -1;
                    if (i < e) {
                        const unsigned key = __float_as_uint(r[u].w);
                        w[u].x = r[u].x; w[u].y = r[u].y; w[u].z = r[u].z;
                        bins[u] = (int)((key >> WOL_IDX_BITS) & 0x7Fu) * WOL_XSUB + (int)(key >> (WOL_IDX_BITS + 7));
                        rank[u] = atomicAdd(&cnt[bins[u]], 1);
                    }
                }
            } else {
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) v[u] = wol_rec<REC>::wide(r[u]);
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) {
                const int i = b + lane + 64 * u;
                bins[u] = -1;
                if (i < e) {
                    double ux = v[u].x * g.box.ib[0]; ux -= floor(ux);
                    double uy = v[u].y * g.box.ib[1]; uy -= floor(uy);
                    double uz = v[u].z * g.box.ib[2]; uz -= floor(uz);
                    int cx = (int)(ux * (double)nx); cx = cx >= nx ? nx - 1 : (cx < 0 ? 0 : cx);  // as wol_cell_coord
                    w[u].x = (float)((ux * (double)nx - (double)cx) * sx);
                    w[u].y = (float)((uy * (double)ny - (double)cy) * sy);
                    w[u].z = (float)((uz * (double)nz - (double)cz) * sz);
                    const int sub = min(WOL_XSUB - 1, (int)(w[u].x * inv_wf));
                    bins[u] = cx * WOL_XSUB + sub;
                    rank[u] = atomicAdd(&cnt[bins[u]], 1);
                }
            }
            }
        } else {
            // ---- long rows: count first ----
            for (int i = lane; i < e - b; i += 64) {
                if constexpr (IDX) {
                    const unsigned key = __float_as_uint(in[src_of(i)].w);
                    atomicAdd(&cnt[(int)((key >> WOL_IDX_BITS) & 0x7Fu) * WOL_XSUB + (int)(key >> (WOL_IDX_BITS + 7))], 1);
                    continue;
                }
                const double4 p = wol_rec_load(in, src_of(i));
                double ux = p.x * g.box.ib[0]; ux -= floor(ux);
                int cx = (int)(ux * (double)nx); cx = cx >= nx ? nx - 1 : (cx < 0 ? 0 : cx);
                const float wx = (float)((ux * (double)nx - (double)cx) * sx);
                const int bin = cx * WOL_XSUB + min(WOL_XSUB - 1, (int)(wx * inv_wf));
                atomicAdd(&cnt[bin], 1);
            }
        }
        __builtin_amdgcn_wave_barrier();
        {   // exclusive scan over the row's fine bins (a lane takes whole cells: four bins each), cell_start and the
            // packed sub-bin populations of the row's cells
            int c[2][WOL_XSUB], s = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int j = 0; j < WOL_XSUB; ++j) {
                    const int cell = lane * cpl + k;
                    c[k][j] = (k < cpl && cell < nx) ? cnt[cell * WOL_XSUB + j] : 0;
                    s += c[k][j];
                }
            const int inc = wol_wave_scan_incl(s);
            int run = inc - s;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int cell = lane * cpl + k;
                if (k < cpl && cell < nx) {
                    cs[cell] = b + run;
                    if (c4) c4[cell] = (unsigned)min(c[k][0], 255) | ((unsigned)min(c[k][1], 255) << 8) |
                                       ((unsigned)min(c[k][2], 255) << 16) | ((unsigned)min(c[k][3], 255) << 24);
#pragma unroll
                    for (int j = 0; j < WOL_XSUB; ++j) {
                        cnt[cell * WOL_XSUB + j] = run;
                        if (!fits) cur[cell * WOL_XSUB + j] = run;
                        run += c[k][j];
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (fits) {
#pragma unroll
            for (int u = 0; u < SLOTS; ++u)
                if (bins[u] >= 0) {
                    const int d = cnt[bins[u]] + rank[u];
                    if constexpr (IDX) st64[d] = (int)(__float_as_uint(r[u].w) & ((1u << WOL_IDX_BITS) - 1u));
                    else st64[d] = r[u];
                    w[u].w = wol_pack_cell_tag(bins[u] / WOL_XSUB, bins[u] % WOL_XSUB, rank[u]);
                    st32[d] = w[u];
                }
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < e - b; i += 64) {
                so[b + i] = st64[i];
                if (so32) so32[b + i] = st32[i];
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        // ---- long rows: scatter straight into the final arrays ----
        for (int i = lane; i < e - b; i += 64) {
            const REC praw = in[src_of(i)];
            if constexpr (IDX) {
                const unsigned key = __float_as_uint(praw.w);
                const int cx = (int)((key >> WOL_IDX_BITS) & 0x7Fu), sub = (int)(key >> (WOL_IDX_BITS + 7));
                const int bin = cx * WOL_XSUB + sub;
                const int dst = atomicAdd(&cur[bin], 1);
                so[b + dst] = (int)(key & ((1u << WOL_IDX_BITS) - 1u));
                if (so32) {
                    float4 q = praw;
                    q.w = wol_pack_cell_tag(cx, sub, dst - cnt[bin]);
                    so32[b + dst] = q;
                }
                continue;
            }
            const double4 p = wol_rec<REC>::wide(praw);
            double ux = p.x * g.box.ib[0]; ux -= floor(ux);
            double uy = p.y * g.box.ib[1]; uy -= floor(uy);
            double uz = p.z * g.box.ib[2]; uz -= floor(uz);
            int cx = (int)(ux * (double)nx); cx = cx >= nx ? nx - 1 : (cx < 0 ? 0 : cx);  // as wol_cell_coord
            float4 q;
            q.x = (float)((ux * (double)nx - (double)cx) * sx);
            q.y = (float)((uy * (double)ny - (double)cy) * sy);
            q.z = (float)((uz * (double)nz - (double)cz) * sz);
            const int sub = min(WOL_XSUB - 1, (int)(q.x * inv_wf));
            const int bin = cx * WOL_XSUB + sub;
            const int dst = atomicAdd(&cur[bin], 1);
            if constexpr (!IDX) so[b + dst] = praw;
            if (so32) {
                q.w = wol_pack_cell_tag(cx, sub, dst - cnt[bin]);
                so32[b + dst] = q;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (cz == nz - 1 && threadIdx.x == 0) cell_start[g.cell_base + (int64_t)nz * ny * nx] = n;
}

// row offsets of cell-sorted frames: lets the in-row sort above run behind the other (cell-level) sorts too
__global__ void __launch_bounds__(WOL_BLOCK)
k_rows_from_cells(const FrameGrid* __restrict__ grids, const int* __restrict__ cell_start, int* __restrict__ row_off, int ro_stride) {
    const FrameGrid& g = grids[blockIdx.y];
    const int nrow = g.nc[1] * g.nc[2];
    const int r = blockIdx.x * WOL_BLOCK + threadIdx.x;
    if (r <= nrow) row_off[(int64_t)blockIdx.y * ro_stride + r] = cell_start[g.cell_base + (int64_t)r * g.nc[0]];
}

// =============================================================================================
// epilogue shared by the search kernels
// =============================================================================================
struct QOut {
    double* q;          // [F][ns] or null
    float* q32;         // [F][ns] or null (float32 storage variant)
    int32_t* nn_idx;    // [F][ns][4] or null
    double* nn_dist;    // [F][ns][4] or null
    double* ang;        // [F][ns][6] or null
    unsigned long long* qhist;  // [qbins] or null
    unsigned long long* dhist;  // [dbins] or null
    int qbins, dbins;
    double dbinwidth;
    // per-frame sums of q and q^2 as exact fixed-point integers, [F][4] = {q_hi, q_lo, q2_hi, q2_lo}:
    // value = hi * 2^-30 + lo * 2^-70.  Integer addition is associative, so mean/var do not depend on
    // the order in which waters arrive (the cell sort's arrival ranks are not reproducible).  Or null.
    unsigned long long* stat;
};

// q (or q^2) -> (hi, lo) with x = hi * 2^-30 + lo * 2^-70, hi = floor(x * 2^30) (two's complement for
// negative q), 0 <= lo < 2^40.  Exact for |x| >= 2^-22, truncated below 2^-70 otherwise.
__device__ __forceinline__ void wol_fix_split(double x, long long* hi, long long* lo) {
    const double t = x * 1073741824.0;            // 2^30, exact scaling
    const double fl = floor(t);
    *hi = (long long)fl;
    *lo = (long long)((t - fl) * 1099511627776.0);  // 2^40
}

// the same split without the emulated double -> int64 conversions: adding 1.5 * 2^52 to an integer-valued double
// of magnitude < 2^51 leaves 2^51 + x in the mantissa field
__device__ __forceinline__ long long wol_small_int_of(double x) {
    const long long b = __double_as_longlong(x + 6755399441055744.0);
    return (b & 0x000FFFFFFFFFFFFFll) - 0x0008000000000000ll;
}
__device__ __forceinline__ void wol_fix_split_fast(double x, long long* hi, long long* lo) {
    const double t = x * 1073741824.0;            // 2^30, exact scaling; |t| < 2^34 for any q or q^2
    const double fl = floor(t);
    *hi = wol_small_int_of(fl);
    *lo = wol_small_int_of(floor((t - fl) * 1099511627776.0));  // 2^40: truncation of a non-negative value
}

// rare paths: one lane adds its water straight to the frame's accumulators
__device__ __forceinline__ void wol_stat_add_lane(const QOut& o, int f, double q) {
    if (!o.stat) return;
    long long h, l, h2, l2;
    wol_fix_split(q, &h, &l);
    wol_fix_split(q * q, &h2, &l2);
    unsigned long long* a = o.stat + 4 * (size_t)f;
    atomicAdd(&a[0], (unsigned long long)h); atomicAdd(&a[1], (unsigned long long)l);
    atomicAdd(&a[2], (unsigned long long)h2); atomicAdd(&a[3], (unsigned long long)l2);
}

// mean and population variance of every frame from the accumulators (np.mean / np.var, ddof = 0)
// one wave per frame
__global__ void __launch_bounds__(64)
k_stat_finalize(const unsigned long long* __restrict__ stat, const long long* __restrict__ part, int slots,
                int F, int n, double* __restrict__ mean, double* __restrict__ var) {
    const int f = blockIdx.x, lane = threadIdx.x;
    if (f >= F) return;
    long long s[4] = {0, 0, 0, 0};
    if (part) {   // the certification kernel's per-wave shares (integer sums: exact in any order)
        const longlong4* p = reinterpret_cast<const longlong4*>(part) + (size_t)f * slots;
        for (int k = lane; k < slots; k += 64) { const longlong4 v = p[k]; s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w; }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += __shfl_xor(s[j], off, 64);
    }
    if (lane != 0) return;
    const long long* a = reinterpret_cast<const long long*>(stat) + 4 * (size_t)f;
    s[0] += a[0]; s[1] += a[1]; s[2] += a[2]; s[3] += a[3];
    const double s1 = (double)s[0] * 9.313225746154785e-10 + (double)s[1] * 8.470329472543003e-22;   // 2^-30, 2^-70
    const double s2 = (double)s[2] * 9.313225746154785e-10 + (double)s[3] * 8.470329472543003e-22;
    const double m = s1 / (double)n;
    if (mean) mean[f] = m;
    if (var) var[f] = s2 / (double)n - m * m;
}

// P* are the coordinates of the selected neighbours as stored (not yet re-imaged)
__device__ __forceinline__ void wol_finish_water(const double a[3], const Top4& t, const double N0[3],
                                                 const double N1[3], const double N2[3], const double N3[3],
                                                 const Box3& b, const QOut& o, int64_t slot, int f,
                                                 int* __restrict__ lds_qh, int* __restrict__ lds_dh) {
    double P0[3] = {0, 0, 0}, P1[3] = {0, 0, 0}, P2[3] = {0, 0, 0}, P3[3] = {0, 0, 0};
    const int nb = t.n;
    if (nb >= 1) wol_neighbour_point(a, N0[0], N0[1], N0[2], b, P0);
    if (nb >= 2) wol_neighbour_point(a, N1[0], N1[1], N1[2], b, P1);
    if (nb >= 3) wol_neighbour_point(a, N2[0], N2[1], N2[2], b, P2);
    if (nb >= 4) wol_neighbour_point(a, N3[0], N3[1], N3[2], b, P3);
    double ang[6];
    const double q = wol_q_from_points(a, P0, P1, P2, P3, nb, ang);
    if (o.q) o.q[slot] = q;
    if (o.q32) o.q32[slot] = (float)q;
    wol_stat_add_lane(o, f, q);
    const double nanv = __builtin_nan("");
    if (o.nn_idx) {
        int4 v;
        v.x = nb >= 1 ? t.j0 : -1; v.y = nb >= 2 ? t.j1 : -1; v.z = nb >= 3 ? t.j2 : -1; v.w = nb >= 4 ? t.j3 : -1;
        reinterpret_cast<int4*>(o.nn_idx)[slot] = v;
    }
    if (o.nn_dist) {
        double* d = o.nn_dist + slot * 4;
        d[0] = nb >= 1 ? t.d0 : nanv; d[1] = nb >= 2 ? t.d1 : nanv;
        d[2] = nb >= 3 ? t.d2 : nanv; d[3] = nb >= 4 ? t.d3 : nanv;
    }
    if (o.ang) {
        double* d = o.ang + slot * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = ang[k];
    }
    if (lds_qh) {
        const int bin = wol_np_hist_bin(q, o.qbins, 0.0, 1.0);
        if (bin >= 0) atomicAdd(&lds_qh[bin], 1);
    }
    if (lds_dh) {
        int bin;
        if (nb >= 1 && (bin = wol_dist_bin(t.d0, o.dbinwidth, o.dbins)) >= 0) atomicAdd(&lds_dh[bin], 1);
        if (nb >= 2 && (bin = wol_dist_bin(t.d1, o.dbinwidth, o.dbins)) >= 0) atomicAdd(&lds_dh[bin], 1);
        if (nb >= 3 && (bin = wol_dist_bin(t.d2, o.dbinwidth, o.dbins)) >= 0) atomicAdd(&lds_dh[bin], 1);
        if (nb >= 4 && (bin = wol_dist_bin(t.d3, o.dbinwidth, o.dbins)) >= 0) atomicAdd(&lds_dh[bin], 1);
    }
}

__device__ __forceinline__ void wol_hist_zero(int* lds, int nbins) {
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) lds[k] = 0;
}
__device__ __forceinline__ void wol_hist_flush(const int* lds, int nbins, unsigned long long* g) {
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) {
        const int v = lds[k];
        if (v) atomicAdd(&g[k], (unsigned long long)v);
    }
}

// =============================================================================================
// exact cell-list search
// =============================================================================================
struct SearchArgs {
    const FrameGrid* grids;
    const double4* sorted;   // [F][n]
    const int* cell_start;
    int n, F;
    const double* sub;       // external queries (single frame) or null = the waters themselves
    int64_t sub_stride;
    int ns;                  // queries per frame
    double lowsq, highsq, high;
    QOut out;
    int tiles_per_frame;
    int xcd_affine;
    // index records (see FastArgs): sidx[F][n] or null, the caller's frames of bare O rows
    const int* sidx;
    const double* src;
    int64_t src_fstride, src_astride;
};

// tile -> (frame, tile in frame).  With >= 8 frames in flight the 8 XCDs (blocks b and b+8 share
// one, dealt round-robin) each walk their own frames so a frame's sorted array stays in that
// XCD's 4 MiB L2; speed only, never correctness.
__device__ __forceinline__ bool wol_tile_of(const SearchArgs& a, int it, int* frame, int* tile) {
    if (a.xcd_affine) {
        const int xcd = blockIdx.x & 7;
        const int k = blockIdx.x >> 3;
        const int K = gridDim.x >> 3;
        const int64_t m = (int64_t)it * K + k;
        const int fi = (int)(m / a.tiles_per_frame);
        const int f = fi * 8 + xcd;
        if (f >= a.F) return false;
        *frame = f;
        *tile = (int)(m % a.tiles_per_frame);
        return true;
    }
    const int64_t m = (int64_t)it * gridDim.x + blockIdx.x;
    if (m >= (int64_t)a.F * a.tiles_per_frame) return false;
    *frame = (int)(m / a.tiles_per_frame);
    *tile = (int)(m % a.tiles_per_frame);
    return true;
}

__device__ __forceinline__ int wol_wrap_cell(int c, int nc) {
    c += (c < 0) ? nc : 0;
    c -= (c >= nc) ? nc : 0;
    return c;
}

// exact search of one query point c against frame grid g; returns the top-4 (slots into sp)
__device__ __forceinline__ void wol_search_exact_one(const double c[3], const FrameGrid& g,
                                                     const double4* __restrict__ sp,
                                                     const int* __restrict__ cs, double lowsq,
                                                     double highsq, double high, Top4& t) {
    double fr[3];
    int cc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) cc[k] = wol_cell_coord(c[k], g.box.ib[k], g.nc[k], &fr[k]);
    top4_init(t);
    for (int shell = 0; shell <= g.maxshell; ++shell) {
        const int zl = max(-shell, -g.lo[2]), zh = min(shell, g.hi[2]);
        const int yl = max(-shell, -g.lo[1]), yh = min(shell, g.hi[1]);
        const int xl = max(-shell, -g.lo[0]), xh = min(shell, g.hi[0]);
        for (int dz = zl; dz <= zh; ++dz) {
            const int cz = wol_wrap_cell(cc[2] + dz, g.nc[2]);
            for (int dy = yl; dy <= yh; ++dy) {
                const int cy = wol_wrap_cell(cc[1] + dy, g.nc[1]);
                const bool face = (abs(dz) == shell) || (abs(dy) == shell);
                for (int dx = xl; dx <= xh; ++dx) {
                    if (!face && abs(dx) != shell) continue;  // interior of the cube: done in earlier shells
                    const int cx = wol_wrap_cell(cc[0] + dx, g.nc[0]);
                    const int cell = (cz * g.nc[1] + cy) * g.nc[0] + cx;
                    const int b = cs[cell], e = cs[cell + 1];
                    for (int j = b; j < e; ++j) {
                        const double4 v = sp[j];
                        double dist;
                        bool in_mask;
                        if (wol_eval_pair(c, v.x, v.y, v.z, g.box, lowsq, highsq, t.bound2, &dist, &in_mask))
                            top4_insert(t, dist, (int)__double_as_longlong(v.w), j);
                    }
                }
            }
        }
        // radius guaranteed complete after this shell: along each axis the visited slab reaches
        // shell*cw + (distance to the nearer wall of the own cell); an axis whose cells have all
        // been visited imposes no limit
        double rg = __builtin_huge_val();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool all = (shell >= g.lo[k]) && (shell >= g.hi[k]) && (g.lo[k] + g.hi[k] + 1 >= g.nc[k]);
            if (!all) {
                const double wall = fmin(fr[k], 1.0 - fr[k]) * g.cw[k];
                rg = fmin(rg, (double)shell * g.cw[k] + wall);
            }
        }
        rg -= 1e-6;  // binning rounds at cell walls; a micro-angstrom of slack covers it
        if (rg >= high) break;
        if (t.n == 4 && t.d3 <= rg) break;
    }
}

__global__ void __launch_bounds__(WOL_BLOCK)
k_search_exact(SearchArgs a) {
    extern __shared__ int lds_hist[];
    int* lds_qh = a.out.qhist ? lds_hist : nullptr;
    int* lds_dh = a.out.dhist ? lds_hist + (a.out.qhist ? a.out.qbins : 0) : nullptr;
    if (lds_qh) wol_hist_zero(lds_qh, a.out.qbins);
    if (lds_dh) wol_hist_zero(lds_dh, a.out.dbins);
    __syncthreads();

    for (int it = 0;; ++it) {
        int f, tile;
        if (!wol_tile_of(a, it, &f, &tile)) break;
        const int s = tile * WOL_BLOCK + threadIdx.x;
        if (s >= a.ns) continue;
        const FrameGrid& g = a.grids[f];
        if (g.maxshell < 0) continue;  // frame handled by the all-pairs kernel
        const double4* __restrict__ sp = a.sorted + (int64_t)f * a.n;
        const int* __restrict__ cs = a.cell_start + g.cell_base;
        double c[3];
        int out_row;
        if (a.sub) {
            const double* p = a.sub + (int64_t)s * a.sub_stride;
            c[0] = p[0]; c[1] = p[1]; c[2] = p[2];
            out_row = s;
        } else {
            const double4 v = sp[s];
            c[0] = v.x; c[1] = v.y; c[2] = v.z;
            out_row = (int)__double_as_longlong(v.w);
        }
        Top4 t;
        wol_search_exact_one(c, g, sp, cs, a.lowsq, a.highsq, a.high, t);
        const double4 n0 = sp[t.s0], n1 = sp[t.s1], n2 = sp[t.s2], n3 = sp[t.s3];
        const double N0[3] = {n0.x, n0.y, n0.z}, N1[3] = {n1.x, n1.y, n1.z};
        const double N2[3] = {n2.x, n2.y, n2.z}, N3[3] = {n3.x, n3.y, n3.z};
        wol_finish_water(c, t, N0, N1, N2, N3, g.box, a.out, (int64_t)f * a.ns + out_row, f, lds_qh, lds_dh);
    }
    __syncthreads();
    if (lds_qh) wol_hist_flush(lds_qh, a.out.qbins, a.out.qhist);
    if (lds_dh) wol_hist_flush(lds_dh, a.out.dbins, a.out.dhist);
}

// =============================================================================================
// LDS-tiled all-pairs kernels (exact arithmetic, any box incl. non-periodic axes)
// =============================================================================================
template <typename T>
struct PointsView {
    const T* base;
    int64_t stride;  // elements of T between rows
    int n;
};

template <typename T>
__global__ void __launch_bounds__(WOL_BLOCK)
k_brute_q(PointsView<T> pos, PointsView<T> sub, Box3 box, double lowsq, double highsq, QOut out,
          int64_t out_base, int frame) {
    extern __shared__ int lds_hist[];
    __shared__ double tx[WOL_BLOCK], ty[WOL_BLOCK], tz[WOL_BLOCK];
    int* lds_qh = out.qhist ? lds_hist : nullptr;
    int* lds_dh = out.dhist ? lds_hist + (out.qhist ? out.qbins : 0) : nullptr;
    if (lds_qh) wol_hist_zero(lds_qh, out.qbins);
    if (lds_dh) wol_hist_zero(lds_dh, out.dbins);

    const int s = blockIdx.x * WOL_BLOCK + threadIdx.x;
    const bool active = s < sub.n;
    double c[3] = {0.0, 0.0, 0.0};
    if (active) {
        const T* p = sub.base + (int64_t)s * sub.stride;
        c[0] = (double)p[0]; c[1] = (double)p[1]; c[2] = (double)p[2];
    }
    Top4 t;
    top4_init(t);
    for (int j0 = 0; j0 < pos.n; j0 += WOL_BLOCK) {
        __syncthreads();
        const int j = j0 + threadIdx.x;
        if (j < pos.n) {
            const T* p = pos.base + (int64_t)j * pos.stride;
            tx[threadIdx.x] = (double)p[0]; ty[threadIdx.x] = (double)p[1]; tz[threadIdx.x] = (double)p[2];
        }
        __syncthreads();
        const int cnt = min(WOL_BLOCK, pos.n - j0);
        if (active) {
            for (int k = 0; k < cnt; ++k) {
                double dist;
                bool in_mask;
                if (wol_eval_pair(c, tx[k], ty[k], tz[k], box, lowsq, highsq, t.bound2, &dist, &in_mask))
                    top4_insert(t, dist, j0 + k, j0 + k);
            }
        }
    }
    if (active) {
        double N[4][3];
        const int sl[4] = {t.s0, t.s1, t.s2, t.s3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const T* p = pos.base + (int64_t)(k < t.n ? sl[k] : 0) * pos.stride;
            N[k][0] = (double)p[0]; N[k][1] = (double)p[1]; N[k][2] = (double)p[2];
        }
        wol_finish_water(c, t, N[0], N[1], N[2], N[3], box, out, out_base + s, frame, lds_qh, lds_dh);
    }
    __syncthreads();
    if (lds_qh) wol_hist_flush(lds_qh, out.qbins, out.qhist);
    if (lds_dh) wol_hist_flush(lds_dh, out.dbins, out.dhist);
}

// f2py-compatible mask: int32, Fortran order (ns, n): element (i,j) at i + j*ns.
// waterlib.f90:710-743 / 830-862 (the all-pairs form is the same predicate, exactly symmetric).
#define WOL_MASK_TJ 64
__global__ void __launch_bounds__(WOL_BLOCK)
k_brute_mask(PointsView<double> pos, PointsView<double> sub, Box3 box, double lowsq, double highsq,
             int32_t* __restrict__ mask) {
    __shared__ double tx[WOL_MASK_TJ], ty[WOL_MASK_TJ], tz[WOL_MASK_TJ];
    const int i = blockIdx.x * WOL_BLOCK + threadIdx.x;
    const int j0 = blockIdx.y * WOL_MASK_TJ;
    if (threadIdx.x < WOL_MASK_TJ && j0 + threadIdx.x < pos.n) {
        const double* p = pos.base + (int64_t)(j0 + threadIdx.x) * pos.stride;
        tx[threadIdx.x] = p[0]; ty[threadIdx.x] = p[1]; tz[threadIdx.x] = p[2];
    }
    __syncthreads();
    if (i >= sub.n) return;
    const double* q = sub.base + (int64_t)i * sub.stride;
    const double a0 = q[0], a1 = q[1], a2 = q[2];
    const int cnt = min(WOL_MASK_TJ, pos.n - j0);
    for (int k = 0; k < cnt; ++k) {
        const double dx = wol_minimg1(tx[k] - a0, box.L[0], box.ib[0]);
        const double dy = wol_minimg1(ty[k] - a1, box.L[1], box.ib[1]);
        const double dz = wol_minimg1(tz[k] - a2, box.L[2], box.ib[2]);
        const double d2 = wol_sumsq(dx, dy, dz);
        mask[(int64_t)i + (int64_t)(j0 + k) * sub.n] = (d2 > lowsq && d2 <= highsq) ? 1 : 0;
    }
}

// =============================================================================================
// level-1 waterlib routines (tiny; kept on the GPU so the product has no CPU compute path)
// =============================================================================================
__global__ void k_reimage(const double* __restrict__ pos, int n, double r0, double r1, double r2, Box3 box,
                          double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ref[3] = {r0, r1, r2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double d = wol_minimg1(pos[3 * (int64_t)i + k] - ref[k], box.L[k], box.ib[k]);
        out[3 * (int64_t)i + k] = ref[k] + d;  // waterlib.f90:45
    }
}

__global__ void k_tetra_cos_ang(const double* __restrict__ neigh, int n, double r0, double r1, double r2,
                                Box3 box, double* __restrict__ allangs) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t)n * n) return;
    const int i = (int)(id / n), j = (int)(id % n);
    if (i == j) { allangs[id] = 0.0; return; }
    const int u = min(i, j), v = max(i, j);
    const double ref[3] = {r0, r1, r2};
    double p1[3], p2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p1[k] = ref[k] + wol_minimg1(neigh[3 * (int64_t)u + k] - ref[k], box.L[k], box.ib[k]);
        p2[k] = ref[k] + wol_minimg1(neigh[3 * (int64_t)v + k] - ref[k], box.L[k], box.ib[k]);
    }
    allangs[id] = wol_cos_angle3(p1, ref, p2);
}

__global__ void k_lsi_dists(const double* __restrict__ neigh, int n, double r0, double r1, double r2, Box3 box,
                            double* __restrict__ dist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double dx = wol_minimg1(neigh[3 * (int64_t)i] - r0, box.L[0], box.ib[0]);
    const double dy = wol_minimg1(neigh[3 * (int64_t)i + 1] - r1, box.L[1], box.ib[1]);
    const double dz = wol_minimg1(neigh[3 * (int64_t)i + 2] - r2, box.L[2], box.ib[2]);
    dist[i] = sqrt(wol_sumsq(dx, dy, dz));
}

// =============================================================================================
// water orientation (waterlib.f90:954-1044: AngBetween, watOrient, watOHvec).
// One lane per water: 72 B in (O, H1, H2), 16 B (two angles) or 24 B (unit dipole) out -- HBM-bound streaming.
// =============================================================================================
// AngBetween (waterlib.f90:954-965) behind its acos: the wrap into (-pi, pi] and the conversion to degrees
__device__ __forceinline__ double wol_ang_wrap_deg(double phi) {
    const double pi = WOL_PI;
    const double DegPerRad = 180.0 / pi;
    double r = fmod(phi + pi, pi * 2.0) - pi;
    if (r < -pi) r = r + pi * 2.0;
    return r * DegPerRad;
}

__device__ __forceinline__ double wol_ang_between(const double a[3], const double b[3]) {
    const double phi = fmin(1.0, fmax(-1.0, (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]));
    return wol_ang_wrap_deg(acos(phi));
}

// A cut-off angle prepared on the host for exact decisions `AngBetween(...) < angCut` (generalHbonds,
// waterlib.f90:1200): A = the cut in radians as a double, cos A as a double-double, sin A.  The device's acos may
// differ from the host libm's in the last place; within 1e-9 degrees of the cut the CORRECTLY ROUNDED acos is formed
// instead, from the expansion of acos about A (cos(A + e) = phi  =>  e = (cos A - phi)/sin A - (cos A / 2 sin A) e^2,
// |e| < 2e-11: the next term is below 1e-33), and the reference's own chain (wrap, degrees, compare) is applied to it.
struct AngCut {
    double cut_deg;
    double A, cA_hi, cA_lo, sA;
    int refine;     // 0: the cut is too close to 0 or 180 degrees for the expansion (sin A ~ 0); device acos decides
};

__device__ __forceinline__ bool wol_ang_below_cut(const double a[3], const double b[3], const AngCut& k) {
    const double phi = fmin(1.0, fmax(-1.0, (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]));
    double ang = wol_ang_wrap_deg(acos(phi));
    if (k.refine && fabs(ang - k.cut_deg) < 1e-9) {
        const double e1 = ((k.cA_hi - phi) + k.cA_lo) / k.sA;            // cA_hi - phi is exact (Sterbenz)
        const double e = e1 - (k.cA_hi / (2.0 * k.sA)) * (e1 * e1);
        ang = wol_ang_wrap_deg(k.A + e);                                 // one rounding: the correctly rounded acos(phi)
    }
    return ang < k.cut_deg;
}

__device__ __forceinline__ void wol_oh_vectors(const double* __restrict__ opos, const double* __restrict__ hpos, int i,
                                               const Box3& b, double v1[3], double v2[3], double dip[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double o = opos[3 * (int64_t)i + k];
        v1[k] = wol_minimg1(hpos[6 * (int64_t)i + k] - o, b.L[k], b.ib[k]);
        v2[k] = wol_minimg1(hpos[6 * (int64_t)i + 3 + k] - o, b.L[k], b.ib[k]);
        dip[k] = wol_minimg1(v1[k] + v2[k], b.L[k], b.ib[k]);
    }
}

__global__ void __launch_bounds__(256)
k_wat_orient(const double* __restrict__ opos, const double* __restrict__ hpos, int n, double r0, double r1, double r2,
             Box3 b, double* __restrict__ angdip, double* __restrict__ angplane) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double rn = sqrt(wol_sumsq(r0, r1, r2));
    const double ref[3] = {r0 / rn, r1 / rn, r2 / rn};
    double v1[3], v2[3], dip[3], u[3];
    wol_oh_vectors(opos, hpos, i, b, v1, v2, dip);
    const double nd = sqrt(wol_sumsq(dip[0], dip[1], dip[2]));
    u[0] = dip[0] / nd; u[1] = dip[1] / nd; u[2] = dip[2] / nd;
    angdip[i] = wol_ang_between(u, ref);
    const double pl[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
    const double np_ = sqrt(wol_sumsq(pl[0], pl[1], pl[2]));
    u[0] = pl[0] / np_; u[1] = pl[1] / np_; u[2] = pl[2] / np_;
    angplane[i] = wol_ang_between(u, ref);
}

__global__ void __launch_bounds__(256)
k_wat_oh_vec(const double* __restrict__ opos, const double* __restrict__ hpos, int n, Box3 b, double* __restrict__ normdip) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v1[3], v2[3], dip[3];
    wol_oh_vectors(opos, hpos, i, b, v1, v2, dip);
    const double nd = sqrt(wol_sumsq(dip[0], dip[1], dip[2]));
#pragma unroll
    for (int k = 0; k < 3; ++k) normdip[3 * (int64_t)i + k] = dip[k] / nd;
}
