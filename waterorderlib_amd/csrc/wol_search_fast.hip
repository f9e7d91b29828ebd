// wol_search_fast.hip -- the filtered neighbour search (included by wol_api.hip): the shared pieces (FastArgs, the
// branch-free top-5 insert), the 3-D tile proposal kernel k_search_fast, the certification kernel k_finish_fast and
// the exact kernel for the waters the filter cannot certify, k_search_exact_list.  The default proposal kernel is
// k_search_rows (wol_search_rows.hip); k_search_fast is the fallback for frames of more than 2^20 waters and the
// second opinion in the tests.
//
// k_search_fast: a 512-thread workgroup owns a 3-D TILE of the cell grid (sx x ry x sz cells, ~480 waters = one lane
// each) and stages the tile plus its one-cell halo -- periodic images already shifted into the tile's local frame --
// from the cell-sorted fp32 array into LDS as pair records (x0,x1,y0,y1)(z0,z1,id0,id1), pre-scaled so that a squared
// distance comes out in integer key units.  Each lane sweeps the 27 cells around its water as nine contiguous LDS
// spans, per candidate pair: 3 v_pk_add_f32, 3 v_pk_fma_f32 (the first adds 2^23, which turns the float into a
// fixed-point integer), 2 x (v_min saturate, v_lshl_or key = d2 << 12 | LDS index) and two branch-free sorted top-5
// inserts (one v_min_u32 + four v_med3_u32 each).
// The fp32 keys only PROPOSE the four nearest neighbours.  They are accepted when the proposal is provably the exact
// answer -- the 5th key is more than 2*eps above the 4th (eps = bound on the fp32 error in key units) and, after
// re-evaluating the four finalists from the fp64 coordinates with the reference's arithmetic (k_finish_fast), the 4th
// distance lies inside the radius the proposal kernel's stencil guarantees to be complete.  Everything else
// (near-ties, sparse neighbourhoods, dense cells that overflow the tile buffer) is appended to a list that
// k_search_exact_list finishes with the exact shell-walking search.  So the results are those of the exact path by
// construction; the filter only decides who needs it.
#pragma once

// Geometry of the proposal kernel (overridable at build time for tuning runs).
#ifndef WOL_FAST_BLOCK
#define WOL_FAST_BLOCK 512    // waves of a block share one staged tile; each sweeps one batch of 64 waters
#endif
#ifndef WOL_FAST_CAPP
#define WOL_FAST_CAPP 1024    // staged PAIRS per tile (cells are padded to an even count); with the tables
                              // this keeps a block under 53 KiB of LDS: three blocks (24 waves) per CU
#endif
#ifndef WOL_FAST_FILL
#define WOL_FAST_FILL 97      // per cent of the lanes a tile may fill on average (a denser tile takes a second sweep)
#endif
#define WOL_FAST_TARGET ((WOL_FAST_BLOCK * WOL_FAST_FILL) / 100)  // waters per tile aimed at (one lane each)
#define WOL_FAST_PADP 40      // head-room (pairs) incl. the dummy pair
#ifndef WOL_FAST_MAXROWS
#define WOL_FAST_MAXROWS 96   // (sz+2)*(ry+2) <= 96
#endif
#define WOL_FAST_MAXRY 30
#define WOL_FAST_MAXSZ 4
#define WOL_FAST_MAXQROWS 64   // sz * ry rows of own waters
#define WOL_FAST_MAXSX 12
#define WOL_CP 16             // row stride of the cell tables: sx + 3 <= 15
#define WOL_FAST_MAXCELL 20   // a cell with more waters sends its tile to the exact path (span <= 64 entries)
#define WOL_ID_BITS 12        // key = d2 (20 bits) << 12 | entry index in the staged tile
#ifndef WOL_XSUB
#define WOL_XSUB 4            // fine x bins per cell of the row-merged kernel (wol_search_rows.hip)
#endif

// Diagnostic build only (-DWOL_PHASE_STAMPS): per-phase shader-clock totals of k_search_fast, added
// by lane 0 of every block into a buffer that nothing else reads.  The product build has no stamps.
#ifdef WOL_PHASE_STAMPS
#define WOL_STAMP(slot)                                                                 \
    do {                                                                                \
        if (threadIdx.x == 0) {                                                         \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();               \
            atomicAdd(&a.stamps[slot], now_ - stamp_prev_);                             \
            stamp_prev_ = now_;                                                         \
        }                                                                               \
    } while (0)
#else
#define WOL_STAMP(slot) do { } while (0)
#endif

struct FastArgs {
    const FrameGrid* grids;
    const double4* sorted;  // [F][n]
    const float4* sorted32; // [F][n] fp32 copy in key units (k_scatter)
    const int* cell_start;
    int n, F;
    double highsq, high;
    QOut out;
    int tiles_per_frame;    // max over the frames of the batch
    int xcd_affine;
    unsigned long long* slow_list;  // (frame << 32 | sorted slot) of waters left to the exact kernel
    int* slow_count;
    int fast_q;             // 1: q straight from the cosines (no acos/deg/cos round trip); needs out.ang == null
    const int* level;       // [F] density level picked by k_pick_level (index into FrameGrid::lev_*) in the low byte; bit 8: the
                            // frame has (layer, tile row) pairs without any water (tile_any says which)
    const int* tile_any;    // [F][tile_any_stride] k_search_rows: 1 where tile row ty of layer cz (index cz * nty + ty) holds water, or null
    int tile_any_stride;
    int4* prop;             // [F][n] by sorted slot: sorted slots of the four proposed neighbours
    int* prop_key;          // [F][n]: 4th key (d2 in key units) or -1 = not certifiable by keys
    int* work;              // k_search_rows: eight tile counters (one per XCD, 64 B apart; zeroed by the host) or null = static tile order
    unsigned long long* stamps;  // diagnostic builds only
    int rows;               // 1: proposals come from k_search_rows (x reach counted in fine bins, its own eps)
    const unsigned* cnt4;   // per cell: populations of its four fine x bins as bytes (k_sort_in_row); k_search_rows only
    long long* stat_part;   // [F][stat_slots][4]: per-wave partial sums of q, q^2 of k_finish_fast (zeroed by the host), or null
    int stat_slots;
    // index records (wol_idx_key): no fp64 copy of the frames exists; sidx[F][n] = original index by sorted slot, the
    // proposals name ORIGINAL indices, and coordinates are read from the caller's frames (bare O rows: 3 doubles per water)
    const int* sidx;        // or null = classic records in `sorted`
    const double* src;      // water i of frame f at src + f * src_fstride + i * src_astride
    int64_t src_fstride, src_astride;
};

// where the certification and the exact search get a water's fp64 coordinates from: the sorted records (REC = double4 /
// float4), or -- index records -- the source frame through the original index
template <class REC> struct wol_rec_src {
    const REC* p;
    __device__ __forceinline__ double4 by_slot(int j) const { return wol_rec<REC>::wide(p[j]); }
    __device__ __forceinline__ double4 by_prop(int j) const { return wol_rec<REC>::wide(p[j]); }   // (proposals name slots)
};
struct wol_idx_src {
    const int* sidx;      // of the frame
    const double* rows;   // of the frame
    int64_t astride;
    __device__ __forceinline__ double4 by_prop(int idx) const {   // (proposals name original indices)
        const double* q = rows + astride * (int64_t)idx;
        double4 v;
        v.x = q[0]; v.y = q[1]; v.z = q[2]; v.w = __longlong_as_double((long long)idx);
        return v;
    }
    __device__ __forceinline__ double4 by_slot(int j) const { return by_prop(sidx[j]); }
};
template <class REC, bool IDX> struct wol_src_of { typedef wol_rec_src<REC> type; };
template <class REC> struct wol_src_of<REC, true> { typedef wol_idx_src type; };
template <class REC, bool IDX, class ARGS>
__device__ __forceinline__ typename wol_src_of<REC, IDX>::type wol_frame_src(const ARGS& a, int f) {
    if constexpr (IDX) return wol_idx_src{a.sidx + (int64_t)f * a.n, a.src + (int64_t)f * a.src_fstride, a.src_astride};
    else return wol_rec_src<REC>{reinterpret_cast<const REC*>(a.sorted) + (int64_t)f * a.n};
}

// sorted insert of `key` into k0 <= k1 <= ... <= k4 (the largest falls off): new k_i is the median of
// (old k_{i-1}, old k_i, key).  Done from the top down so every op still sees the old lower
// neighbour and the update is in place: one v_min_u32 + four v_med3_u32, no moves.
__device__ __forceinline__ void wol_top5_insert(unsigned key, unsigned& k0, unsigned& k1, unsigned& k2,
                                                unsigned& k3, unsigned& k4) {
    // one asm statement with all five keys tied in place: the register allocator then keeps them in
    // fixed registers across the loop instead of copying around per-instruction constraints
    asm("v_med3_u32 %4, %3, %4, %5\n\t"
        "v_med3_u32 %3, %2, %3, %5\n\t"
        "v_med3_u32 %2, %1, %2, %5\n\t"
        "v_med3_u32 %1, %0, %1, %5\n\t"
        "v_min_u32 %0, %0, %5"
        : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3), "+v"(k4)
        : "v"(key));
}

typedef float wol_f2 __attribute__((ext_vector_type(2)));

// One span of candidate PAIRS against the lane's water.  A staged pair is two float4:
// (x0, x1, y0, y1) and (z0, z1, id0, id1); the three differences and the three fused multiply-adds
// are packed (v_pk_add_f32 / v_pk_fma_f32: two candidates per instruction).  The +2^23 bias makes the
// float's mantissa the integer squared distance in key units; it is saturated and shifted above the
// entry id.  The trip count is wave-uniform (vote); a lane past the end of its own span is pointed
// at the tile's dummy pair, whose far-away coordinates saturate, so no per-element masks are needed.
template <bool CENTRE>
__device__ __forceinline__ void wol_fast_span(const char* __restrict__ rec_bytes, int pb, int np, int dummy_off,
                                              wol_f2 xi, wol_f2 yi, wol_f2 zi, unsigned self_id,
                                              unsigned tmaxbits, unsigned& k0, unsigned& k1, unsigned& k2,
                                              unsigned& k3, unsigned& k4) {
    const wol_f2 bias = {8388608.0f, 8388608.0f};
    int off = pb * 32;              // byte offset of the lane's current pair record
    const int end = (pb + np) * 32;
    // wave-uniform trip count: the longest span among the 64 lanes (butterfly max, then scalar)
    int tmax_np = np;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) tmax_np = max(tmax_np, __shfl_xor(tmax_np, sh, 64));
    const int T = __builtin_amdgcn_readfirstlane(tmax_np);
    for (int t = 0; t < T; ++t, off += 32) {
        const bool v = off < end;
        const int o = v ? off : dummy_off;
        const float4 a = *reinterpret_cast<const float4*>(rec_bytes + o);
        const float4 b = *reinterpret_cast<const float4*>(rec_bytes + o + 16);
        const wol_f2 dx = (wol_f2){a.x, a.y} - xi, dy = (wol_f2){a.z, a.w} - yi, dz = (wol_f2){b.x, b.y} - zi;
        wol_f2 tt = __builtin_elementwise_fma(dx, dx, bias);
        tt = __builtin_elementwise_fma(dy, dy, tt);
        tt = __builtin_elementwise_fma(dz, dz, tt);
        const unsigned id0 = __float_as_uint(b.z), id1 = __float_as_uint(b.w);
        unsigned key0 = (min(__float_as_uint(tt.x), tmaxbits) << WOL_ID_BITS) | id0;
        unsigned key1 = (min(__float_as_uint(tt.y), tmaxbits) << WOL_ID_BITS) | id1;
        if (CENTRE) {
            key0 = (id0 == self_id) ? 0xFFFFFFFFu : key0;
            key1 = (id1 == self_id) ? 0xFFFFFFFFu : key1;
        }
        wol_top5_insert(key0, k0, k1, k2, k3, k4);
        wol_top5_insert(key1, k0, k1, k2, k3, k4);
    }
}

// q straight from the clamped cosines: identical to wol_q_from_points up to the rounding of the
// acos -> degrees -> cos round trip (|dq| < 2e-14); only for four real neighbours
__device__ __forceinline__ double wol_cos_at(const double p1[3], const double p2[3], const double p3[3]) {
    if ((p1[0] == p2[0] && p1[1] == p2[1] && p1[2] == p2[2]) ||
        (p2[0] == p3[0] && p2[1] == p3[1] && p2[2] == p3[2]))
        return 1.0;  // CosAngle3 returns 0 degrees
    double v21[3], v23[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { v21[k] = p1[k] - p2[k]; v23[k] = p3[k] - p2[k]; }
    const double nrm = sqrt(wol_sumsq(v21[0], v21[1], v21[2]) * wol_sumsq(v23[0], v23[1], v23[2]));
    const double dot = (v21[0] * v23[0] + v21[1] * v23[1]) + v21[2] * v23[2];
    return fmin(1.0, fmax(-1.0, dot / nrm));
}

__device__ __forceinline__ double wol_q_from_cosines(const double a[3], const double P0[3], const double P1[3],
                                                     const double P2[3], const double P3[3]) {
    const double c01 = wol_cos_at(P0, a, P1) + (1.0 / 3.0), c02 = wol_cos_at(P0, a, P2) + (1.0 / 3.0);
    const double c03 = wol_cos_at(P0, a, P3) + (1.0 / 3.0), c12 = wol_cos_at(P1, a, P2) + (1.0 / 3.0);
    const double c13 = wol_cos_at(P1, a, P3) + (1.0 / 3.0), c23 = wol_cos_at(P2, a, P3) + (1.0 / 3.0);
    double s = -0.0;
    s += c01 * c01; s += c02 * c02; s += c03 * c03; s += c12 * c12; s += c13 * c13; s += c23 * c23;
    return 1.0 - (3.0 / 8.0) * s;
}

// finish for four real neighbours without the angle outputs
__device__ __forceinline__ void wol_finish_water_fastq(const double a[3], const Top4& t, const double N0[3],
                                                       const double N1[3], const double N2[3], const double N3[3],
                                                       const Box3& b, const QOut& o, int64_t slot, int f,
                                                       int* __restrict__ lds_qh, int* __restrict__ lds_dh) {
    double P0[3], P1[3], P2[3], P3[3];
    wol_neighbour_point(a, N0[0], N0[1], N0[2], b, P0);
    wol_neighbour_point(a, N1[0], N1[1], N1[2], b, P1);
    wol_neighbour_point(a, N2[0], N2[1], N2[2], b, P2);
    wol_neighbour_point(a, N3[0], N3[1], N3[2], b, P3);
    const double q = wol_q_from_cosines(a, P0, P1, P2, P3);
    if (o.q) o.q[slot] = q;
    if (o.q32) o.q32[slot] = (float)q;
    wol_stat_add_lane(o, f, q);
    if (o.nn_idx) {
        int4 v; v.x = t.j0; v.y = t.j1; v.z = t.j2; v.w = t.j3;
        reinterpret_cast<int4*>(o.nn_idx)[slot] = v;
    }
    if (o.nn_dist) {
        double* d = o.nn_dist + slot * 4;
        d[0] = t.d0; d[1] = t.d1; d[2] = t.d2; d[3] = t.d3;
    }
    if (lds_qh) {
        const int bin = wol_np_hist_bin(q, o.qbins, 0.0, 1.0);
        if (bin >= 0) atomicAdd(&lds_qh[bin], 1);
    }
    if (lds_dh) {
        int bin;
        if ((bin = wol_dist_bin(t.d0, o.dbinwidth, o.dbins)) >= 0) atomicAdd(&lds_dh[bin], 1);
        if ((bin = wol_dist_bin(t.d1, o.dbinwidth, o.dbins)) >= 0) atomicAdd(&lds_dh[bin], 1);
        if ((bin = wol_dist_bin(t.d2, o.dbinwidth, o.dbins)) >= 0) atomicAdd(&lds_dh[bin], 1);
        if ((bin = wol_dist_bin(t.d3, o.dbinwidth, o.dbins)) >= 0) atomicAdd(&lds_dh[bin], 1);
    }
}

// ---------------------------------------------------------------------------------------------
// kernel 0: density level of every frame.  The fullest 4-row x 2-layer window of cell rows gives the
// local waters-per-cell the tiles have to cope with (a slab with vacuum, a droplet: far above the box
// average); the first prepared tile shape that fits it is selected.  One block per frame.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WOL_BLOCK)
k_pick_level(const FrameGrid* __restrict__ grids, const int* __restrict__ cell_start, const int* __restrict__ row_off,
             int ro_stride, int* __restrict__ level, int* __restrict__ tile_any, int tile_any_stride, int rows_cap) {
    extern __shared__ int rows[];     // [rows_cap] waters per cell row (dynamic: small frames come in batches of thousands, and
                                      // the kernel runs beside the second sort level -- a block must not hold 48 KB for 81 rows)
    __shared__ int smax[WOL_BLOCK];
    const int f = blockIdx.x, tid = threadIdx.x;
    const FrameGrid& g = grids[f];
    const int nx = g.nc[0], ny = g.nc[1], nz = g.nc[2], nrow = ny * nz;
    const int* __restrict__ cs = cell_start + g.cell_base;
    const bool ok = g.maxshell >= 0 && nrow <= rows_cap;
    // waters per (z, y) cell row: from the first sort level's row offsets where it ran (the kernel then runs beside the
    // second level), else from the finished cell table
    if (ok && row_off) {
        const int* __restrict__ ro = row_off + (int64_t)f * ro_stride;
        for (int r = tid; r < nrow; r += WOL_BLOCK) rows[r] = ro[r + 1] - ro[r];
    } else if (ok)
        for (int r = tid; r < nrow; r += WOL_BLOCK) rows[r] = cs[(r + 1) * nx] - cs[r * nx];
    __syncthreads();
    // densest window of 4 rows x 2 layers (about the size of a tile; a single row is too noisy)
    int m = 0;
    const int wy = min(4, ny), wz = min(2, nz);
    if (ok)
        for (int r = tid; r < nrow; r += WOL_BLOCK) {
            const int cz = r / ny, cy = r - cz * ny;
            int sum = 0;
            for (int dz = 0; dz < wz; ++dz)
                for (int dy = 0; dy < wy; ++dy)
                    sum += rows[wol_wrap_cell(cz + dz, nz) * ny + wol_wrap_cell(cy + dy, ny)];
            m = max(m, sum);
        }
    smax[tid] = m;
    __syncthreads();
    for (int off = WOL_BLOCK >> 1; off > 0; off >>= 1) {
        if (tid < off) smax[tid] = max(smax[tid], smax[tid + off]);
        __syncthreads();
    }
    __shared__ int s_lev, s_empty;
    if (tid == 0) {
        int lev = g.nlev - 1;
        if (ok) {
            const float wpc = (float)smax[0] / (float)(nx * wy * wz);
            for (int l = g.nlev - 1; l >= 0; --l)
                if (wpc <= g.lev_wpc[l]) lev = l;
        }
        s_lev = lev;
        s_empty = 0;
    }
    __syncthreads();
    const int lev = s_lev;
    // tile rows of the row-merged proposal kernel (ry cell rows of one layer) without any water -- the vacuum of a slab, the
    // space around a droplet: flagged here, skipped there before a single table is built
    if (tile_any && ok && nz * (int)g.row_nty[lev] <= tile_any_stride) {
        const int ry = g.row_ry[lev], nty = g.row_nty[lev];
        int* __restrict__ ta = tile_any + (int64_t)f * tile_any_stride;
        for (int idx = tid; idx < nz * nty; idx += WOL_BLOCK) {
            const int cz = idx / nty, ty = idx - cz * nty;
            const int y0 = ty * ry, re = min(ry, ny - y0);
            int sum = 0;
            for (int y = 0; y < re; ++y) sum += rows[cz * ny + y0 + y];
            ta[idx] = sum > 0 ? 1 : 0;
            if (sum == 0) s_empty = 1;
        }
        __syncthreads();
    }
    if (tid == 0) level[f] = lev | ((tile_any && ok && s_empty) ? 0x100 : 0);
}

// ---------------------------------------------------------------------------------------------
// kernel 1: proposal.  Small register footprint (no fp64), occupancy limited by LDS only.
// Writes per water (indexed by its slot in the sorted array) the sorted slots of the four proposed
// neighbours and the 4th key (or -1 when the proposal cannot be certified by keys alone).
// ---------------------------------------------------------------------------------------------
#ifndef WOL_FAST_WPS
#define WOL_FAST_WPS 6   // waves per SIMD the register allocation must allow (3 blocks x 8 waves / 4 SIMDs)
#endif
__global__ void __launch_bounds__(WOL_FAST_BLOCK, WOL_FAST_WPS)
k_search_fast(FastArgs a) {
    __shared__ float4 rec[2 * (WOL_FAST_CAPP + WOL_FAST_PADP)];     // pair records
    __shared__ int slot[2 * WOL_FAST_CAPP];                         // entry -> slot in the sorted array
    __shared__ int cellsrc[WOL_FAST_MAXROWS * WOL_CP];              // staged cell -> first slot
    __shared__ unsigned short cellpos[WOL_FAST_MAXROWS * WOL_CP];   // staged cell -> first entry (even)
    __shared__ unsigned char cellcnt[WOL_FAST_MAXROWS * WOL_CP];    // staged cell -> waters in it
    __shared__ int rowlen[WOL_FAST_MAXROWS];
    __shared__ int qrow[WOL_FAST_MAXQROWS + 2];                     // own rows -> first query number
    __shared__ int s_flag, s_total;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
#ifdef WOL_PHASE_STAMPS
    unsigned long long stamp_prev_ = __builtin_amdgcn_s_memtime();
#endif
    // tiles are dealt without divisions: with >= 8 frames in flight each XCD (blocks b, b+8, ...
    // share one) walks its own frames so a frame's arrays stay in that XCD's L2 (speed only)
    const bool aff = a.xcd_affine != 0;
    const int fstep = aff ? 8 : 1;
    const int tstep = aff ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    int f = aff ? (int)(blockIdx.x & 7) : 0;
    int tile = aff ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;

    // tiles of frame f at the level the device picked for it (0 for frames this kernel skips)
    auto tiles_of = [&](int ff) -> int {
        if (ff >= a.F) return 0;
        const FrameGrid& gg = a.grids[ff];
        return (gg.maxshell < 0 || !gg.fast_ok) ? 0 : gg.lev_ntiles[a.level[ff] & 0xFF];
    };
    int nt_f = tiles_of(f);
    for (;; tile += tstep) {
        while (f < a.F && tile >= nt_f) { tile -= nt_f; f += fstep; nt_f = tiles_of(f); }
        if (f >= a.F) break;
        const FrameGrid& g = a.grids[f];
        const int lev = a.level[f] & 0xFF;
        const int nx = g.nc[0], ny = g.nc[1], nz = g.nc[2];
        const int gsx = g.lev_sx[lev], gry = g.lev_ry[lev], gsz = g.lev_sz[lev];
        const int gntx = g.lev_ntx[lev], gnty = g.lev_nty[lev];
        // tile -> (tx, ty, tz) through float reciprocals (exact for these small operands)
        const int ntxy = gntx * gnty;
        const int tz = (int)(((float)tile + 0.5f) / (float)ntxy);
        const int trem = tile - tz * ntxy;
        const int ty = (int)(((float)trem + 0.5f) / (float)gntx);
        const int tx = trem - ty * gntx;
        const int x0 = tx * gsx, sxe = min(gsx, nx - x0);
        const int y0 = ty * gry, re = min(gry, ny - y0);
        const int z0 = tz * gsz, sze = min(gsz, nz - z0);
        const int rpl = re + 2;           // staged rows per z layer
        const int nrows = (sze + 2) * rpl, ncr = sxe + 2;
        const int nqr = sze * re;         // rows holding the tile's own waters
        const float inv_rpl = 1.0f / (float)rpl, inv_re = 1.0f / (float)re;
        const float4* __restrict__ sp32 = a.sorted32 + (int64_t)f * a.n;
        const int* __restrict__ cs = a.cell_start + g.cell_base;
        const int64_t frame_off = (int64_t)f * a.n;
        const float fx = (float)(g.cw[0] * (double)g.scale), fy = (float)(g.cw[1] * (double)g.scale);
        const float fz = (float)(g.cw[2] * (double)g.scale);

        if (tid == 0) s_flag = 0;  // (every thread copied the previous tile's flag before its last barrier)
        __syncthreads();           // the previous tile is done with the LDS tables
        WOL_STAMP(0);
        // ---- phase 1: 16 lanes per staged row, one per cell (k = 0 left halo cell, 1..sxe the tile's
        //      own, sxe+1 right halo; halos wrap): population and first slot, one L2 round trip for
        //      the whole table; the row length (cells padded to even) by a 16-lane butterfly ----
        {
            const int k = tid & 15;
            for (int r = tid >> 4; r < ((nrows + 1) & ~1); r += WOL_FAST_BLOCK >> 4) {
                int cnt = 0, src = 0;
                if (r < nrows && k < ncr) {
                    const int zi = (int)(((float)r + 0.5f) * inv_rpl), yi = r - zi * rpl;
                    const int gz = wol_wrap_cell(z0 + zi - 1, nz), gy = wol_wrap_cell(y0 + yi - 1, ny);
                    const int c = (gz * ny + gy) * nx + wol_wrap_cell(x0 + k - 1, nx);
#ifdef WOL_EXP_NO_CS_LOAD   // ablation build: timing only, results are wrong
                    src = c * 2; cnt = 2 + (c & 1);
#else
                    src = cs[c];
                    cnt = cs[c + 1] - src;
#endif
                    cellsrc[r * WOL_CP + k] = src;
                    cellcnt[r * WOL_CP + k] = (unsigned char)min(cnt, 255);
                    if (cnt > WOL_FAST_MAXCELL) s_flag = 1;
                }
                int len = (cnt + 1) & ~1;
                len += __shfl_xor(len, 1, 64); len += __shfl_xor(len, 2, 64);
                len += __shfl_xor(len, 4, 64); len += __shfl_xor(len, 8, 64);
                if (k == 0 && r < nrows) rowlen[r] = len;
            }
        }
        __syncthreads();
        WOL_STAMP(1);
        // ---- phase 2: row bases by a wave scan, then every row lays out its cells (even starts) ----
        if (wave == 0) {
            int run = 0;  // nrows <= 96: two rounds of 64
            for (int r0 = 0; r0 < nrows; r0 += 64) {
                const int r = r0 + lane;
                const int len = (r < nrows) ? rowlen[r] : 0;
                int inc = len;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int v = __shfl_up(inc, off, 64);
                    if (lane >= off) inc += v;
                }
                if (r < nrows) rowlen[r] = run + inc - len;  // exclusive
                run += __shfl(inc, 63, 64);
            }
            if (lane == 0) {
                if (run > 2 * WOL_FAST_CAPP) s_flag = 1;
                s_total = run;  // total entries incl. padding
            }
        }
        __syncthreads();
        if (tid < nrows) {
            int pos = rowlen[tid];
            for (int k = 0; k < ncr; ++k) {
                cellpos[tid * WOL_CP + k] = (unsigned short)pos;
                pos += ((int)cellcnt[tid * WOL_CP + k] + 1) & ~1;
            }
            cellpos[tid * WOL_CP + ncr] = (unsigned short)pos;
        }
        // own rows (layers 1..sze, rows 1..re): running number of the tile's waters (queries)
        if (wave == 1) {
            int c = 0;
            if (lane < nqr) {
                const int lz = (int)(((float)lane + 0.5f) * inv_re), ly = lane - lz * re;
                const int r = (lz + 1) * rpl + ly + 1;
                for (int k = 1; k <= sxe; ++k) c += cellcnt[r * WOL_CP + k];
            }
            int inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(inc, off, 64);
                if (lane >= off) inc += v;
            }
            qrow[lane + 1] = inc;
            if (lane == 0) qrow[0] = 0;
        }
        __syncthreads();
        const bool overflow = s_flag != 0;
        const int total_e = s_total;
        const int dummy_p = (total_e >> 1) + WOL_FAST_PADP - 2;  // behind the read head-room
        WOL_STAMP(2);
        // ---- phase 3: stage.  16 lanes per row, one per cell: two waters at a time become one pair
        //      record; an odd cell is completed (and the head-room filled) with far-away dummies ----
        if (!overflow) {
            const int k = tid & 15;
            for (int r = tid >> 4; r < nrows; r += WOL_FAST_BLOCK >> 4) {
                if (k >= ncr) continue;
                const int zi = (int)(((float)r + 0.5f) * inv_rpl), yi = r - zi * rpl;
                const int src = cellsrc[r * WOL_CP + k], cnt = cellcnt[r * WOL_CP + k];
                const int pos = cellpos[r * WOL_CP + k];
                // cell-local coordinates -> tile frame centred on the tile (periodic images of the halo
                // cells come out shifted automatically: the offset is in staged-cell units)
                const float offx = ((float)(k - 1) - 0.5f * (float)sxe) * fx;
                const float offy = ((float)(yi - 1) - 0.5f * (float)re) * fy;
                const float offz = ((float)(zi - 1) - 0.5f * (float)sze) * fz;
                for (int e0 = 0; e0 < cnt; e0 += 6) {
                    float4 v[6];
#pragma unroll
                    for (int u = 0; u < 6; ++u)  // up to six waters of the cell in flight together
#ifdef WOL_EXP_NO_STAGE_LOAD   // ablation build: timing only, results are wrong
                        if (e0 + u < cnt) { v[u].x = (float)(src + u); v[u].y = (float)(k + u); v[u].z = (float)r; v[u].w = 0.f; }
#else
                        if (e0 + u < cnt) v[u] = sp32[src + e0 + u];
#endif
#pragma unroll
                    for (int u = 0; u < 6; u += 2) {
                        const int e = e0 + u;
                        if (e < cnt) {
                            const bool two = e + 1 < cnt;
                            float4 ra, rb;
                            ra.x = v[u].x + offx; ra.z = v[u].y + offy; rb.x = v[u].z + offz;
                            ra.y = two ? v[u + 1].x + offx : 1.0e7f;
                            ra.w = two ? v[u + 1].y + offy : 1.0e7f;
                            rb.y = two ? v[u + 1].z + offz : 1.0e7f;
                            rb.z = __int_as_float(pos + e); rb.w = __int_as_float(pos + e + 1);
                            const int p = (pos + e) >> 1;
                            rec[2 * p] = ra; rec[2 * p + 1] = rb;
                            slot[pos + e] = src + e;
                            slot[pos + e + 1] = two ? src + e + 1 : src + e;
                        }
                    }
                }
            }
            if (tid < WOL_FAST_PADP) {  // head-room pairs (incl. the dummy pair): nothing near
                const int p = (total_e >> 1) + tid;
                float4 far;
                far.x = far.y = far.z = far.w = 1.0e7f;
                rec[2 * p] = far;
                far.z = far.w = __int_as_float(0);
                rec[2 * p + 1] = far;
            }
        }
        __syncthreads();
        const int nq = qrow[nqr];
        WOL_STAMP(3);

        for (int qb = wave * 64; qb < nq; qb += WOL_FAST_BLOCK) {
            const int qi = qb + lane;
            if (qi >= nq) continue;
            int qr = 0;  // own row of this water: largest qr with qrow[qr] <= qi (binary search, nqr <= 64)
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1)
                if (qr + step < nqr && qrow[qr + step] <= qi) qr += step;
            const int lz = (int)(((float)qr + 0.5f) * inv_re), ly = qr - lz * re;
            const int rq = (lz + 1) * rpl + ly + 1;
            // own cell and entry: walk the row's cells
            int lx = 0, left = qi - qrow[qr];
            while (lx + 1 < sxe && left >= (int)cellcnt[rq * WOL_CP + lx + 1]) { left -= cellcnt[rq * WOL_CP + lx + 1]; ++lx; }
            const int self_slot = cellsrc[rq * WOL_CP + lx + 1] + left;
            int4 prop;
            prop.x = prop.y = prop.z = prop.w = 0;
            int kd3 = -1;
            if (!overflow) {
                const int self_e = cellpos[rq * WOL_CP + lx + 1] + left;
                const float4 ma = rec[2 * (self_e >> 1)], mb = rec[2 * (self_e >> 1) + 1];
                const bool odd = self_e & 1;
                const float mx = odd ? ma.y : ma.x, my = odd ? ma.w : ma.z, mz = odd ? mb.y : mb.x;
                const wol_f2 xi = {mx, mx}, yi = {my, my}, zi = {mz, mz};
                const unsigned tmaxbits = __float_as_uint(g.tmax);
                unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu, k3 = 0xFFFFFFFFu, k4 = 0xFFFFFFFFu;
#pragma unroll
                for (int dzi = 0; dzi < 3; ++dzi) {
#pragma unroll
                    for (int dyi = 0; dyi < 3; ++dyi) {
                        const int r = (lz + dzi) * rpl + ly + dyi;
                        const int pb = cellpos[r * WOL_CP + lx] >> 1, pe = cellpos[r * WOL_CP + lx + 3] >> 1;
                        if (dzi == 1 && dyi == 1)
                            wol_fast_span<true>(reinterpret_cast<const char*>(rec), pb, pe - pb, dummy_p * 32, xi, yi, zi, (unsigned)self_e, tmaxbits, k0, k1, k2, k3, k4);
                        else
                            wol_fast_span<false>(reinterpret_cast<const char*>(rec), pb, pe - pb, dummy_p * 32, xi, yi, zi, (unsigned)self_e, tmaxbits, k0, k1, k2, k3, k4);
                    }
                }
                // the proposal stands only if it is separated from the 5th candidate by more than the
                // fp32 error and is not saturated
                const unsigned d3 = k3 >> WOL_ID_BITS, d4 = k4 >> WOL_ID_BITS;
                const unsigned sat = (1u << (32 - WOL_ID_BITS)) - 1u - (unsigned)g.eps_units;
                if (k3 != 0xFFFFFFFFu && d3 < sat && d4 - d3 > 2u * (unsigned)g.eps_units) {
                    const unsigned m = (1u << WOL_ID_BITS) - 1u;
                    prop.x = slot[k0 & m]; prop.y = slot[k1 & m]; prop.z = slot[k2 & m]; prop.w = slot[k3 & m];
                    kd3 = (int)d3;
                }
            }
#ifdef WOL_EXP_NO_STORE        // ablation build: timing only
            if (kd3 == 123456789) a.prop_key[frame_off + self_slot] = prop.x;
#else
            a.prop[frame_off + self_slot] = prop;
            a.prop_key[frame_off + self_slot] = kd3;
#endif
        }
        WOL_STAMP(4);
    }
}

// ---------------------------------------------------------------------------------------------
// Diagnostic (context option "check_proposals"): the row kernel is trusted to write a key for EVERY water of every tile it
// is given -- a stale key from an earlier batch would pass a certification that proves "these four are close", not "these
// are the four closest".  With the option on the host poisons prop / prop_key before the proposal kernel; this kernel
// then counts the keys that were written and the proposals that name a slot outside the frame, and turns keys still
// poisoned into "not certifiable" so that a violated invariant is reported instead of gathered through.
// ---------------------------------------------------------------------------------------------
#define WOL_PROP_POISON 0x7E7E7E7E
__global__ void __launch_bounds__(256)
k_check_props(int* __restrict__ prop_key, const int4* __restrict__ prop, int64_t total, int n, unsigned long long* __restrict__ counters) {
    unsigned long long written = 0, bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = prop_key[i];
        if (k == WOL_PROP_POISON) { prop_key[i] = -1; continue; }
        ++written;
        if (k >= 0) {
            const int4 p = prop[i];
            if ((unsigned)p.x >= (unsigned)n || (unsigned)p.y >= (unsigned)n || (unsigned)p.z >= (unsigned)n || (unsigned)p.w >= (unsigned)n) ++bad;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { written += __shfl_xor(written, off, 64); bad += __shfl_xor(bad, off, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (written) atomicAdd(&counters[0], written);
        if (bad) atomicAdd(&counters[1], bad);
    }
}

// radius around a water inside which the proposal kernel is guaranteed to have seen every water: one cell
// width plus the distance to the nearest wall of its cell -- along x of its fine bin when the proposals
// come from k_search_rows, whose x window is counted in bins (fp32 binning there vs fp64 here: 1e-5 A margin)
__device__ __forceinline__ double wol_guaranteed_radius(const double c[3], const FrameGrid& g, int rows) {
    double rg = __builtin_huge_val();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double fr;
        (void)wol_cell_coord(c[k], g.box.ib[k], g.nc[k], &fr);
        double wall = fmin(fr, 1.0 - fr) * g.cw[k];
        if (k == 0 && rows) {
            const double ff = fr * (double)WOL_XSUB - floor(fr * (double)WOL_XSUB);
            wall = fmin(ff, 1.0 - ff) * (g.cw[0] / (double)WOL_XSUB);
        }
        rg = fmin(rg, g.cw[k] + wall);
    }
    return rg - 1e-5;
}

// ---------------------------------------------------------------------------------------------
// The trajectory mode's work on one water whose four proposed neighbours are at hand (fp64 coordinates): certify them,
// form q from their minimum-image vectors in proposal order -- no ranking, no second re-image, shared norms -- bin q and
// the four distances, add to the frame sums.  Shared by k_finish_fast<true> (neighbours gathered from global memory) and
// k_finish_tiles (from the tile's LDS copy): the arithmetic is the same statement for statement.  Returns true when the
// water has to take the exact path instead (nothing has been written for it then).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool wol_lean_water(const FastArgs& a, const FrameGrid& g, int f, const double c[3],
                                               const double nbx[4], const double nby[4], const double nbz[4],
                                               int kd3, int eps_u, double dinv, int* lds_qh, int* lds_dh, int orig,
                                               long long* acc_h, long long* acc_l, long long* acc_h2, long long* acc_l2) {
    bool slow = false;
    // The minimum-image vectors d themselves stand in for the reference's (apos + d) - apos (wp:372-373) and
    // for tetraCosAng's second re-image: they differ by rounding of order ulp(apos), 1e-15 relative, which q
    // tolerates (|dq| < 1e-14) and the distance bins do not notice except on a bin edge -- there the
    // reference's chain is evaluated.  One rsqrt per neighbour gives both its norm and its distance.
    // Per neighbour only its unit vector and its distance bin stay live (register pressure decides how many
    // waves cover the gathers' latency).
    double ux[4], uy[4], uz[4];
    int dbin[4];
    double dmax = 0.0;
    bool all_in = true;
    // |d - ((c + d) - c)| per component: two roundings at the magnitude of the coordinates (which may be
    // unwrapped, i.e. far larger than the box); the bin-edge guards below widen with it
    const double cguard = 1e-14 * (fmax(fabs(c[0]), fmax(fabs(c[1]), fabs(c[2]))) + a.high);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // waterlib.f90:733-735 with rint for anint: they differ only on exact half-integers, i.e. a neighbour
        // at exactly half a box length, which the emax test below sends to the exact path anyway
        const double ex = nbx[k] - c[0], ey = nby[k] - c[1], ez = nbz[k] - c[2];
        const double dx = ex - g.box.L[0] * rint(ex * g.box.ib[0]);
        const double dy = ey - g.box.L[1] * rint(ey * g.box.ib[1]);
        const double dz = ez - g.box.L[2] * rint(ez * g.box.ib[2]);
        const double d2 = wol_sumsq(dx, dy, dz);
        // q from d instead of (c + d) - c is off by ~ 4 ulp(c) / d: where that could pass 4e-13 (coordinates
        // thousands of box lengths from the origin, or a neighbour a few thousandths of an Angstrom away: d below
        // a thousandth of the coordinates' magnitude) the reference's chain is evaluated, on the exact path
        const double dlow = cguard * 1e11;
        all_in = all_in && (d2 > dlow * dlow) && (d2 <= a.highsq);
        const double rn = rsqrt(d2);
        const double dist = d2 * rn;
        dmax = fmax(dmax, dist);
        ux[k] = dx * rn; uy[k] = dy * rn; uz[k] = dz * rn;
        // nbin = ceiling(d/binwidth) (waterlib.f90:219-222): multiply by the reciprocal; when the product sits
        // on an integer, the reference's distance and the true division decide
        const double tq = dist * dinv;
        double cb = ceil(tq);
        if (fabs(tq - rint(tq)) < 1e-9 * tq + cguard * dinv) {
            const double vx = (c[0] + dx) - c[0], vy = (c[1] + dy) - c[1], vz = (c[2] + dz) - c[2];
            cb = ceil(sqrt(wol_sumsq(vx, vy, vz)) / a.out.dbinwidth);
        }
        dbin[k] = (cb >= 1.0 && cb <= (double)a.out.dbins) ? (int)cb - 1 : -1;
    }
    // no component of a neighbour vector reaches 0.49 box lengths if the longest vector stays below 0.49 of
    // the shortest box edge (sufficient; tiny boxes may take the exact path a little more often)
    const double emax = dmax * fmax(g.box.ib[0], fmax(g.box.ib[1], g.box.ib[2]));
    // every neighbour inside the smallest cell width is inside the guaranteed radius: the walls only matter beyond
    const double cwmin = fmin(g.cw[0], fmin(g.cw[1], g.cw[2]));
    const double rg = (dmax > cwmin - 1e-4) ? wol_guaranteed_radius(c, g, a.rows) : cwmin;
    const double d3s = (dmax * dmax) * g.S;
    // emax < 0.49: tetraCosAng's second re-image (waterlib.f90:881-883) is the identity
#ifdef WOL_EXP_FINISH_COALESCED
    if (!(dmax > -1.0) || d3s == 12345.678) {
#else
    if (!all_in || !(dmax <= rg) || fabs(d3s - (double)kd3) > (double)eps_u || !(emax < 0.49)) {
#endif
        slow = true;
    } else {
        // the six terms are summed in ascending order of their values: the result must not depend
        // on the order in which the proposal lists the neighbours (which can follow the arrival
        // order of waters inside a cell when two fp32 keys tie)
        double tm[6];
        {
            int m = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = i + 1; j < 4; ++j) {
                    const double dot = (ux[i] * ux[j] + uy[i] * uy[j]) + uz[i] * uz[j];
                    const double cs_ = fmin(1.0, fmax(-1.0, dot)) + (1.0 / 3.0);
                    tm[m++] = cs_ * cs_;
                }
        }
#define WOL_CE(i, j) { const double lo_ = fmin(tm[i], tm[j]), hi_ = fmax(tm[i], tm[j]); tm[i] = lo_; tm[j] = hi_; }
        WOL_CE(0, 5) WOL_CE(1, 3) WOL_CE(2, 4)   // 12-comparator sorting network for 6 keys
        WOL_CE(1, 2) WOL_CE(3, 4)
        WOL_CE(0, 3) WOL_CE(2, 5)
        WOL_CE(0, 1) WOL_CE(2, 3) WOL_CE(4, 5)
        WOL_CE(1, 2) WOL_CE(3, 4)
#undef WOL_CE
        double ssum = -0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m) ssum += tm[m];
        const double q = 1.0 - (3.0 / 8.0) * ssum;
        // This q is within 2e-14 of the reference's acos -> degrees -> cos chain (wp:376-388), which is what
        // np.histogram bins (opl:1493).  Where that could matter -- q within 1e-8 of a bin width from a bin edge
        // (0 and 1 included) -- the water goes to the exact kernel, which evaluates the reference's chain in
        // exact rank order: the integer histogram is the reference's by construction.
        const double tq_ = q * (double)a.out.qbins;
        if (lds_qh && fabs(tq_ - rint(tq_)) < 1e-8 + cguard * (double)a.out.qbins && tq_ > -0.5 &&
            tq_ < (double)a.out.qbins + 0.5) {
            slow = true;
        } else {
            if (a.out.q || a.out.q32) {
                const int64_t orow = (int64_t)f * a.n + orig;
                if (a.out.q) a.out.q[orow] = q;
                if (a.out.q32) a.out.q32[orow] = (float)q;
            }
            wol_fix_split_fast(q, acc_h, acc_l);
            wol_fix_split_fast(q * q, acc_h2, acc_l2);
            if (lds_qh) {
                const int bin = wol_np_hist_bin(q, a.out.qbins, 0.0, 1.0);
                if (bin >= 0) atomicAdd(&lds_qh[bin], 1);
            }
            if (lds_dh) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (dbin[k] >= 0) atomicAdd(&lds_dh[dbin[k]], 1);
            }
        }
    }
    return slow;
}

// ---------------------------------------------------------------------------------------------
// kernel 2: certification + epilogue, one lane per water in sorted order.  Re-evaluates the four
// proposed neighbours from the fp64 coordinates with the reference's arithmetic, accepts when the
// 4th distance is inside the radius the 27-cell stencil guarantees (and agrees with the fp32 key),
// computes angles / q / histograms; everything else goes to the exact list.
// ---------------------------------------------------------------------------------------------
#ifndef WOL_FINISH_WPS
#define WOL_FINISH_WPS 4
#endif
template <bool LEAN, class REC = double4, bool IDX = false>
__global__ void __launch_bounds__(WOL_BLOCK, WOL_FINISH_WPS)
k_finish_fast(FastArgs a) {
    extern __shared__ int lds_hist[];
    int* lds_qh = a.out.qhist ? lds_hist : nullptr;
    int* lds_dh = a.out.dhist ? lds_hist + (a.out.qhist ? a.out.qbins : 0) : nullptr;
    if (lds_qh) wol_hist_zero(lds_qh, a.out.qbins);
    if (lds_dh) wol_hist_zero(lds_dh, a.out.dbins);
    __syncthreads();
    constexpr bool lean = LEAN;  // host: fast_q && no per-neighbour outputs
    const double dinv = (a.out.dbinwidth > 0.0) ? 1.0 / a.out.dbinwidth : 0.0;
    // frames <-> XCDs as in the proposal kernel, so a frame's sorted array is read through one L2
    const bool aff = a.xcd_affine != 0;
    const int fstep = aff ? 8 : 1;
    int kb = aff ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int KB = aff ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    // small frames need fewer blocks than a group has: the group then works on several frames at a time
    const int need = (a.n + WOL_BLOCK - 1) / WOL_BLOCK;
    int fsub = 0, fgroups = 1;
    if (need < KB) { fgroups = KB / need; fsub = kb / need; kb -= fsub * need; KB = need; }
    __shared__ unsigned long long blk_stat[4];
    for (int f = (aff ? (int)(blockIdx.x & 7) : 0) + fstep * fsub; f < a.F && fsub < fgroups; f += fstep * fgroups) {
    long long sum_h = 0, sum_l = 0, sum_h2 = 0, sum_l2 = 0;  // this lane's share of frame f's sums of q, q^2
    const FrameGrid& g = a.grids[f];
    if (g.maxshell < 0) continue;  // all-pairs frame
    const auto sp = wol_frame_src<REC, IDX>(a, f);
    const int eps_u = a.rows ? g.row_eps_units : g.eps_units;
    for (int s0 = kb * WOL_BLOCK; s0 < a.n; s0 += KB * WOL_BLOCK) {  // uniform bounds: whole waves stay in the loop
        const int s = min(s0 + (int)threadIdx.x, a.n - 1);
        const bool live = s0 + (int)threadIdx.x < a.n;
        const int64_t w = (int64_t)f * a.n + s;
        // key, proposal and own coordinates are fetched together (one round trip), the neighbours in a second one
        // (fetching the next 256 waters' share ahead of time was measured: no change, the gathers are what costs)
        const int kd3 = live ? a.prop_key[w] : 0;
        const int4 pr_all = a.prop[w];
        const double4 mev_all = sp.by_slot(s);
        bool slow = live && kd3 < 0;
        long long acc_h = 0, acc_l = 0, acc_h2 = 0, acc_l2 = 0;  // this water's share of the frame sums
        if constexpr (LEAN) { if (live && !slow) {
            // ---- lean path (trajectory mode: q, histograms, no per-neighbour outputs): the four
            //      proposed neighbours are certified and q is formed from their min-image vectors in
            //      proposal order -- no ranking, no second re-image, shared norms ----
            const int4 pr = pr_all;
            const double4 mev = mev_all;
            const double c[3] = {mev.x, mev.y, mev.z};
#ifdef WOL_EXP_FINISH_COALESCED   // ablation build: timing only, results are wrong (neighbours = the next four slots)
            const double4 nb[4] = {sp.by_slot(min(s + 1 + (pr.x & 0), a.n - 1)), sp.by_slot(min(s + 2 + (pr.y & 0), a.n - 1)),
                                   sp.by_slot(min(s + 3 + (pr.z & 0), a.n - 1)), sp.by_slot(min(s + 4 + (pr.w & 0), a.n - 1))};
#else
            const double4 nb[4] = {sp.by_prop(pr.x), sp.by_prop(pr.y), sp.by_prop(pr.z), sp.by_prop(pr.w)};
#endif
            const double nbx[4] = {nb[0].x, nb[1].x, nb[2].x, nb[3].x}, nby[4] = {nb[0].y, nb[1].y, nb[2].y, nb[3].y},
                         nbz[4] = {nb[0].z, nb[1].z, nb[2].z, nb[3].z};
            slow = wol_lean_water(a, g, f, c, nbx, nby, nbz, kd3, eps_u, dinv, lds_qh, lds_dh, (int)__double_as_longlong(mev.w),
                                  &acc_h, &acc_l, &acc_h2, &acc_l2);
        } } else if (live && !slow) {
            const int4 pr = pr_all;
            const double4 mev = mev_all;
            const double c[3] = {mev.x, mev.y, mev.z};
            const double4 n0 = sp.by_prop(pr.x), n1 = sp.by_prop(pr.y), n2 = sp.by_prop(pr.z), n3 = sp.by_prop(pr.w);
            Top4 t;
            top4_init(t);
            double dist;
            bool in_mask, all_in = true;
            const double inf = __builtin_huge_val();
            if (wol_eval_pair(c, n0.x, n0.y, n0.z, g.box, 0.0, a.highsq, inf, &dist, &in_mask))
                top4_insert(t, dist, (int)__double_as_longlong(n0.w), 0);
            all_in = all_in && in_mask;
            if (wol_eval_pair(c, n1.x, n1.y, n1.z, g.box, 0.0, a.highsq, inf, &dist, &in_mask))
                top4_insert(t, dist, (int)__double_as_longlong(n1.w), 1);
            all_in = all_in && in_mask;
            if (wol_eval_pair(c, n2.x, n2.y, n2.z, g.box, 0.0, a.highsq, inf, &dist, &in_mask))
                top4_insert(t, dist, (int)__double_as_longlong(n2.w), 2);
            all_in = all_in && in_mask;
            if (wol_eval_pair(c, n3.x, n3.y, n3.z, g.box, 0.0, a.highsq, inf, &dist, &in_mask))
                top4_insert(t, dist, (int)__double_as_longlong(n3.w), 3);
            all_in = all_in && in_mask;
            // radius the 27-cell stencil guarantees to be complete around this water
            const double rg = wol_guaranteed_radius(c, g, a.rows);
            // the exact 4th key must also agree with the fp32 proposal (guards the error bound itself)
            const double d3s = (t.d3 * t.d3) * g.S;
            if (!all_in || t.n != 4 || !(t.d3 <= rg) || fabs(d3s - (double)kd3) > (double)eps_u) {
                slow = true;
            } else {
                // neighbours in exact rank order (t.s* = position 0..3 in the proposal)
                double N[4][3];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int which = (k == 0) ? t.s0 : (k == 1) ? t.s1 : (k == 2) ? t.s2 : t.s3;
                    const double4 m = (which == 0) ? n0 : (which == 1) ? n1 : (which == 2) ? n2 : n3;
                    N[k][0] = m.x; N[k][1] = m.y; N[k][2] = m.z;
                }
                const int64_t orow = (int64_t)f * a.n + (int)__double_as_longlong(mev.w);
                // (q straight from the cosines only where no histogram is binned from it: bins are the full chain's)
                if (a.fast_q && !lds_qh) wol_finish_water_fastq(c, t, N[0], N[1], N[2], N[3], g.box, a.out, orow, f, lds_qh, lds_dh);
                else wol_finish_water(c, t, N[0], N[1], N[2], N[3], g.box, a.out, orow, f, lds_qh, lds_dh);
            }
        }
        if (slow) {
            const int pos = atomicAdd(a.slow_count, 1);
            a.slow_list[pos] = ((unsigned long long)f << 32) | (unsigned)s;
        }
        sum_h += acc_h; sum_l += acc_l; sum_h2 += acc_h2; sum_l2 += acc_l2;
    }
    if constexpr (LEAN) {
        if (a.out.stat) {
            if (a.stat_part) {
                // frame f is done for this wave: butterfly-sum the four integers over the wave and store them in the
                // wave's own slot of the frame (k_stat_finalize adds the slots up).  No barrier, no atomic: with only
                // a few passes of the block per frame, three block barriers per frame kept the waves in lockstep.
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    sum_h += __shfl_xor(sum_h, off, 64); sum_l += __shfl_xor(sum_l, off, 64);
                    sum_h2 += __shfl_xor(sum_h2, off, 64); sum_l2 += __shfl_xor(sum_l2, off, 64);
                }
                if ((threadIdx.x & 63) == 0) {
                    const int slot = kb * (WOL_BLOCK / 64) + (int)(threadIdx.x >> 6);
                    if (slot < a.stat_slots) {
                        longlong4 v;
                        v.x = sum_h; v.y = sum_l; v.z = sum_h2; v.w = sum_l2;
                        reinterpret_cast<longlong4*>(a.stat_part)[(size_t)f * a.stat_slots + slot] = v;
                    } else {   // (cannot happen with the grids the host launches; kept exact all the same)
                        unsigned long long* st = a.out.stat + 4 * (size_t)f;
                        atomicAdd(&st[0], (unsigned long long)sum_h); atomicAdd(&st[1], (unsigned long long)sum_l);
                        atomicAdd(&st[2], (unsigned long long)sum_h2); atomicAdd(&st[3], (unsigned long long)sum_l2);
                    }
                }
                continue;
            }
            // frame f is done for this block: butterfly-sum the four integers over each wave, combine
            // the waves in LDS, one thread adds the block's share to the frame's accumulators
            if (threadIdx.x < 4) blk_stat[threadIdx.x] = 0ull;
            __syncthreads();
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                sum_h += __shfl_xor(sum_h, off, 64); sum_l += __shfl_xor(sum_l, off, 64);
                sum_h2 += __shfl_xor(sum_h2, off, 64); sum_l2 += __shfl_xor(sum_l2, off, 64);
            }
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(&blk_stat[0], (unsigned long long)sum_h); atomicAdd(&blk_stat[1], (unsigned long long)sum_l);
                atomicAdd(&blk_stat[2], (unsigned long long)sum_h2); atomicAdd(&blk_stat[3], (unsigned long long)sum_l2);
            }
            __syncthreads();
            if (threadIdx.x < 4 && blk_stat[threadIdx.x] != 0ull)
                atomicAdd(&a.out.stat[4 * (size_t)f + threadIdx.x], blk_stat[threadIdx.x]);
            __syncthreads();
        }
    }
    }
    __syncthreads();
    if (lds_qh) wol_hist_flush(lds_qh, a.out.qbins, a.out.qhist);
    if (lds_dh) wol_hist_flush(lds_dh, a.out.dbins, a.out.dhist);
}

// ---------------------------------------------------------------------------------------------
// kernel 3: exact search for the waters the filter could not certify -- HALF A WAVE PER WATER (the 27 cells of the first
// shells fit 32 lanes; at 151 VGPRs three waves per SIMD are resident, and a 100 000-water batch's ~5 000 listed waters
// then need one pass of the resident waves instead of two).
// The cells of each shell are dealt to the 32 lanes (a lane-serial walk is a chain of dependent
// L2 round trips); every lane ranks its cells' candidates with the exact arithmetic into its own
// top-4, a butterfly of sorted-list merges gives all lanes the wave-wide top-4, and the shell loop
// ends under the same guarantee as wol_search_exact_one.
// ---------------------------------------------------------------------------------------------
template <int FIRST = 32>   // 32: over the whole wave; 16: inside each half of it (two waters per wave)
__device__ __forceinline__ void wol_top4_wave_merge(Top4& t) {
#pragma unroll
    for (int off = FIRST; off >= 1; off >>= 1) {
        const double od0 = __shfl_xor(t.d0, off, 64), od1 = __shfl_xor(t.d1, off, 64);
        const double od2 = __shfl_xor(t.d2, off, 64), od3 = __shfl_xor(t.d3, off, 64);
        const int oj0 = __shfl_xor(t.j0, off, 64), oj1 = __shfl_xor(t.j1, off, 64);
        const int oj2 = __shfl_xor(t.j2, off, 64), oj3 = __shfl_xor(t.j3, off, 64);
        const int os0 = __shfl_xor(t.s0, off, 64), os1 = __shfl_xor(t.s1, off, 64);
        const int os2 = __shfl_xor(t.s2, off, 64), os3 = __shfl_xor(t.s3, off, 64);
        top4_insert(t, od0, oj0, os0);  // the two lists hold disjoint candidates at every stage
        top4_insert(t, od1, oj1, os1);
        top4_insert(t, od2, oj2, os2);
        top4_insert(t, od3, oj3, os3);
    }
}

template <class REC, bool IDX = false>   // storage form of the sorted records (wol_rec: double4, or float4 for float32 frames; IDX: none)
__global__ void __launch_bounds__(WOL_BLOCK)
k_search_exact_list(SearchArgs a, const unsigned long long* __restrict__ slow_list,
                    const int* __restrict__ slow_count) {
    extern __shared__ int lds_hist[];
    int* lds_qh = a.out.qhist ? lds_hist : nullptr;
    int* lds_dh = a.out.dhist ? lds_hist + (a.out.qhist ? a.out.qbins : 0) : nullptr;
    if (lds_qh) wol_hist_zero(lds_qh, a.out.qbins);
    if (lds_dh) wol_hist_zero(lds_dh, a.out.dbins);
    __syncthreads();
    const int count = *slow_count;
    const int lane = threadIdx.x & 31;                                  // lane inside the half wave
    const int wave = (blockIdx.x * WOL_BLOCK + threadIdx.x) >> 5;       // half-wave number
    const int nwaves = (gridDim.x * WOL_BLOCK) >> 5;
    for (int i = wave; i < count; i += nwaves) {
        const unsigned long long e = slow_list[i];
        const int f = (int)(e >> 32), s = (int)(e & 0xFFFFFFFFu);
        const FrameGrid& g = a.grids[f];
        const auto sp = wol_frame_src<REC, IDX>(a, f);
        const int* __restrict__ cs = a.cell_start + g.cell_base;
        const double4 v = sp.by_slot(s);
        const double c[3] = {v.x, v.y, v.z};
        double fr[3];
        int cc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) cc[k] = wol_cell_coord(c[k], g.box.ib[k], g.nc[k], &fr[k]);
        Top4 mine;   // this lane's candidates, accumulated over the shells
        top4_init(mine);
        Top4 all;    // wave-wide merge, identical in every lane
        top4_init(all);
        for (int shell = 0; shell <= g.maxshell; ++shell) {
            // shells 0 and 1 together are the 27-cell cube: deal them in one go
            if (shell == 0 && g.maxshell >= 1) continue;
            const int w = 2 * shell + 1;
            const float inv_w = 1.0f / (float)w;
            const int ncube = w * w * w;
            for (int idx = lane; idx < ncube; idx += 32) {
                const int iz = (int)(((float)idx + 0.5f) * inv_w * inv_w);
                const int rem = idx - iz * w * w;
                const int iy = (int)(((float)rem + 0.5f) * inv_w);
                const int ix = rem - iy * w;
                const int dx = ix - shell, dy = iy - shell, dz = iz - shell;
                const bool on_shell = (abs(dx) == shell) || (abs(dy) == shell) || (abs(dz) == shell);
                if (!(on_shell || shell == 1)) continue;  // interior cells belong to earlier shells
                if (dx < -g.lo[0] || dx > g.hi[0] || dy < -g.lo[1] || dy > g.hi[1] || dz < -g.lo[2] || dz > g.hi[2])
                    continue;                             // clipped: small periodic grids visit a cell once
                const int cx = wol_wrap_cell(cc[0] + dx, g.nc[0]);
                const int cy = wol_wrap_cell(cc[1] + dy, g.nc[1]);
                const int cz = wol_wrap_cell(cc[2] + dz, g.nc[2]);
                const int cell = (cz * g.nc[1] + cy) * g.nc[0] + cx;
                const int b = cs[cell], en = cs[cell + 1];
                for (int j = b; j < en; ++j) {
                    const double4 p = sp.by_slot(j);
                    double dist;
                    bool in_mask;
                    if (wol_eval_pair(c, p.x, p.y, p.z, g.box, a.lowsq, a.highsq, mine.bound2, &dist, &in_mask))
                        top4_insert(mine, dist, (int)__double_as_longlong(p.w), j);
                }
            }
            all = mine;
            wol_top4_wave_merge<16>(all);
            double rg = __builtin_huge_val();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool whole = (shell >= g.lo[k]) && (shell >= g.hi[k]) && (g.lo[k] + g.hi[k] + 1 >= g.nc[k]);
                if (!whole) rg = fmin(rg, (double)shell * g.cw[k] + fmin(fr[k], 1.0 - fr[k]) * g.cw[k]);
            }
            rg -= 1e-6;
            if (rg >= a.high) break;
            if (all.n == 4 && all.d3 <= rg) break;
        }
        if (lane == 0) {
            const double4 n0 = sp.by_slot(all.s0), n1 = sp.by_slot(all.s1), n2 = sp.by_slot(all.s2), n3 = sp.by_slot(all.s3);
            const double N0[3] = {n0.x, n0.y, n0.z}, N1[3] = {n1.x, n1.y, n1.z};
            const double N2[3] = {n2.x, n2.y, n2.z}, N3[3] = {n3.x, n3.y, n3.z};
            wol_finish_water(c, all, N0, N1, N2, N3, g.box, a.out, (int64_t)f * a.ns + (int)__double_as_longlong(v.w),
                             f, lds_qh, lds_dh);
        }
    }
    __syncthreads();
    if (lds_qh) wol_hist_flush(lds_qh, a.out.qbins, a.out.qhist);
    if (lds_dh) wol_hist_flush(lds_dh, a.out.dbins, a.out.dhist);
}
