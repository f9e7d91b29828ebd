// wol_search_fast.hip -- the filtered neighbour-search kernel (included by wol_api.hip).
//
// One 256-thread workgroup owns a TILE of the cell grid (sx cells along x by ry rows along y in one
// z layer, ~220 waters) and stages the tile plus its one-cell halo -- 3*(ry+2) rows of sx+2 cells,
// periodic images already shifted into the tile's local frame -- from the cell-sorted fp64 array
// into LDS as fp32 coordinates (x, y, z pre-scaled so that a squared distance comes out in integer
// key units, w = slot in the sorted array).  Each lane owns one water of the tile and sweeps the
// 27 cells around it: nine contiguous LDS spans, 13 VALU ops per candidate:
//     3 v_sub, 3 v_fma (the first adds 2^23, which turns the float into a fixed-point integer),
//     v_min (saturate), v_lshl_or (key = d2 << 11 | LDS index), and a branch-free sorted top-5
//     insert made of one v_min_u32 + four v_med3_u32 that are independent of each other.
// The fp32 keys only PROPOSE the four nearest neighbours.  They are accepted when the proposal is
// provably the exact answer -- the 5th key is more than 2*eps above the 4th (eps = bound on the
// fp32 error in key units) and, after re-evaluating the four finalists from the fp64 coordinates
// with the reference's exact arithmetic (wol_eval_pair), the 4th distance lies inside the radius
// that the 27-cell stencil guarantees to be complete.  Everything else (near-ties, sparse
// neighbourhoods, dense cells that overflow the tile buffer) is appended to a list that
// k_search_exact_list finishes with the exact shell-walking search.  So the results are those of
// the exact path by construction; the filter only decides who needs it.
#pragma once

#define WOL_FAST_CAP 1536     // staged candidates per tile
#define WOL_FAST_MAXROWS 64   // 3*(ry+2) <= 64
#define WOL_CP 13             // row stride of the cell-position table: sx + 3 <= 13
#define WOL_FAST_MAXRY 19
#define WOL_FAST_MAXSX 10

struct FastArgs {
    const FrameGrid* grids;
    const double4* sorted;  // [F][n]
    const int* cell_start;
    int n, F;
    double highsq, high;
    QOut out;
    int tiles_per_frame;    // max over the frames of the batch
    int xcd_affine;
    unsigned long long* slow_list;  // (frame << 32 | sorted slot) of waters left to the exact kernel
    int* slow_count;
    int fast_q;             // 1: q straight from the cosines (no acos/deg/cos round trip); needs out.ang == null
    int4* prop;             // [F][n] by sorted slot: sorted slots of the four proposed neighbours
    int* prop_key;          // [F][n]: 4th key (d2 in key units) or -1 = not certifiable by keys
};

__device__ __forceinline__ bool wol_fast_tile_of(const FastArgs& a, int it, int* frame, int* tile) {
    if (a.xcd_affine) {
        const int xcd = blockIdx.x & 7;
        const int k = blockIdx.x >> 3;
        const int K = gridDim.x >> 3;
        const int64_t m = (int64_t)it * K + k;
        const int f = (int)(m / a.tiles_per_frame) * 8 + xcd;
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

__device__ __forceinline__ unsigned wol_med3u(unsigned a, unsigned b, unsigned c) {
    unsigned r;  // median of three unsigned ints in one VALU op (no clang builtin for the integer form)
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// one LDS span of candidates against the lane's water; keys k0 <= ... <= k4 are the running top-5
template <bool CENTRE>
__device__ __forceinline__ void wol_fast_span(const float4* __restrict__ ent, int b, int e, float xi, float yi,
                                              float zi, int self_e, float tmax, unsigned& k0, unsigned& k1,
                                              unsigned& k2, unsigned& k3, unsigned& k4) {
    float4 nxt = ent[b];  // software prefetch: the next candidate is in flight while this one is ranked
    for (int j = b; j < e; ++j) {
        const float4 c = nxt;
        nxt = ent[j + 1];  // ent has head-room past the last staged entry
        const float dx = c.x - xi, dy = c.y - yi, dz = c.z - zi;
        float t = __builtin_fmaf(dx, dx, 8388608.0f);  // + 2^23: the mantissa becomes the integer d2
        t = __builtin_fmaf(dy, dy, t);
        t = __builtin_fmaf(dz, dz, t);
        t = fminf(t, tmax);
        unsigned key = (__float_as_uint(t) << 11) | (unsigned)j;
        if (CENTRE) key = (j == self_e) ? 0xFFFFFFFFu : key;
        const unsigned n0 = min(k0, key);
        const unsigned n1 = wol_med3u(k0, k1, key);
        const unsigned n2 = wol_med3u(k1, k2, key);
        const unsigned n3 = wol_med3u(k2, k3, key);
        const unsigned n4 = wol_med3u(k3, k4, key);
        k0 = n0; k1 = n1; k2 = n2; k3 = n3; k4 = n4;
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
                                                       const Box3& b, const QOut& o, int64_t slot,
                                                       int* __restrict__ lds_qh, int* __restrict__ lds_dh) {
    double P0[3], P1[3], P2[3], P3[3];
    wol_neighbour_point(a, N0[0], N0[1], N0[2], b, P0);
    wol_neighbour_point(a, N1[0], N1[1], N1[2], b, P1);
    wol_neighbour_point(a, N2[0], N2[1], N2[2], b, P2);
    wol_neighbour_point(a, N3[0], N3[1], N3[2], b, P3);
    const double q = wol_q_from_cosines(a, P0, P1, P2, P3);
    if (o.q) o.q[slot] = q;
    if (o.q32) o.q32[slot] = (float)q;
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
// kernel 1: proposal.  Small register footprint (no fp64), occupancy limited by LDS only.
// Writes per water (indexed by its slot in the sorted array) the sorted slots of the four proposed
// neighbours and the 4th key (or -1 when the proposal cannot be certified by keys alone).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WOL_BLOCK)
k_search_fast(FastArgs a) {
    __shared__ float4 ent[WOL_FAST_CAP + 1];
    __shared__ unsigned char ent_k[WOL_FAST_CAP];
    __shared__ int cellpos[WOL_FAST_MAXROWS * WOL_CP];
    __shared__ int rowsum[WOL_FAST_MAXROWS];
    __shared__ int qstart[WOL_FAST_MAXRY + 2];
    const int tid = threadIdx.x;

    for (int it = 0;; ++it) {
        int f, tile;
        if (!wol_fast_tile_of(a, it, &f, &tile)) break;
        const FrameGrid& g = a.grids[f];
        if (tile >= g.ntiles || g.maxshell < 0 || !g.fast_ok) continue;  // uniform over the block
        const int nx = g.nc[0], ny = g.nc[1], nz = g.nc[2];
        const int tx = tile % g.ntx, ty = (tile / g.ntx) % g.nty, cz = tile / (g.ntx * g.nty);
        const int x0 = tx * g.sx, sxe = min(g.sx, nx - x0);
        const int y0 = ty * g.ry, re = min(g.ry, ny - y0);
        const int rpl = re + 2;           // staged rows per z layer
        const int nrows = 3 * rpl, ncr = sxe + 2, ntab = nrows * ncr;
        const double4* __restrict__ sp = a.sorted + (int64_t)f * a.n;
        const int* __restrict__ cs = a.cell_start + g.cell_base;
        const int64_t frame_off = (int64_t)f * a.n;

        __syncthreads();  // the previous tile is done with the LDS tables
        // ---- phase 1: population of every staged cell (kept in registers for phase 3) ----
        int c_src[4], c_cnt[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int i = tid + m * WOL_BLOCK;
            c_src[m] = 0; c_cnt[m] = 0;
            if (i < ntab) {
                const int r = i / ncr, k = i - r * ncr;
                const int zi = r / rpl, yi = r - zi * rpl;
                const int gz = wol_wrap_cell(cz + zi - 1, nz), gy = wol_wrap_cell(y0 + yi - 1, ny);
                const int gx = wol_wrap_cell(x0 + k - 1, nx);
                const int cell = (gz * ny + gy) * nx + gx;
                c_src[m] = cs[cell];
                c_cnt[m] = cs[cell + 1] - c_src[m];
                cellpos[r * WOL_CP + k] = c_cnt[m];
            }
        }
        __syncthreads();
        // ---- phase 2: exclusive scan, row-major over (row, cell) ----
        if (tid < WOL_FAST_MAXROWS) {
            int s = 0;
            if (tid < nrows)
                for (int k = 0; k < ncr; ++k) s += cellpos[tid * WOL_CP + k];
            int inc = s;  // inclusive scan across the 64 lanes of wave 0
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(inc, off, 64);
                if (tid >= off) inc += v;
            }
            rowsum[tid] = inc - s;
        }
        __syncthreads();
        if (tid < nrows) {
            int run = rowsum[tid];
            for (int k = 0; k < ncr; ++k) {
                const int c = cellpos[tid * WOL_CP + k];
                cellpos[tid * WOL_CP + k] = run;
                run += c;
            }
            cellpos[tid * WOL_CP + ncr] = run;
        }
        __syncthreads();
        const int total = cellpos[(nrows - 1) * WOL_CP + ncr];
        const bool overflow = total > WOL_FAST_CAP;
        // ---- phase 3: stage the halo'd tile as scaled fp32, images shifted into the local frame ----
        if (!overflow) {
            const double fx = g.cw[0] * (double)g.scale, fy = g.cw[1] * (double)g.scale, fz = g.cw[2] * (double)g.scale;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int i = tid + m * WOL_BLOCK;
                if (i >= ntab) break;
                const int r = i / ncr, k = i - r * ncr;
                const int zi = r / rpl, yi = r - zi * rpl;
                const int uz = cz + zi - 1, uy = y0 + yi - 1, ux = x0 + k - 1;  // unwrapped cell coordinates
                // offsets (in cells) that move a wrapped coordinate into the tile frame
                const double ox = (double)(ux - wol_wrap_cell(ux, nx) - x0);
                const double oy = (double)(uy - wol_wrap_cell(uy, ny) - y0);
                const double oz = (double)(uz - wol_wrap_cell(uz, nz) - cz);
                const int src = c_src[m], cnt = c_cnt[m];
                const int dst = cellpos[r * WOL_CP + k];
                for (int e0 = 0; e0 < cnt; e0 += 4) {
                    double4 v[4];
#pragma unroll
                    for (int u4 = 0; u4 < 4; ++u4)  // issue the loads together, then convert
                        if (e0 + u4 < cnt) v[u4] = sp[src + e0 + u4];
#pragma unroll
                    for (int u4 = 0; u4 < 4; ++u4) {
                        if (e0 + u4 < cnt) {
                            double u;
                            float4 o;
                            u = v[u4].x * g.box.ib[0]; u -= floor(u); o.x = (float)((u * (double)nx + ox) * fx);
                            u = v[u4].y * g.box.ib[1]; u -= floor(u); o.y = (float)((u * (double)ny + oy) * fy);
                            u = v[u4].z * g.box.ib[2]; u -= floor(u); o.z = (float)((u * (double)nz + oz) * fz);
                            o.w = __int_as_float(src + e0 + u4);
                            ent[dst + e0 + u4] = o;
                            ent_k[dst + e0 + u4] = (unsigned char)k;
                        }
                    }
                }
            }
        }
        // ---- phase 4: the tile's own waters: rows 1..re of the middle layer, cells 1..sxe ----
        if (tid < 64) {
            int c = 0;
            if (tid < re) {
                const int r = rpl + tid + 1;
                c = cellpos[r * WOL_CP + sxe + 1] - cellpos[r * WOL_CP + 1];
            }
            int inc = c;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const int v = __shfl_up(inc, off, 64);
                if (tid >= off) inc += v;
            }
            if (tid <= WOL_FAST_MAXRY) qstart[tid + 1] = inc;
            if (tid == 0) qstart[0] = 0;
        }
        __syncthreads();
        const int nq = qstart[re];

        for (int qb = 0; qb < nq; qb += WOL_BLOCK) {
            const int qi = qb + tid;
            if (qi >= nq) continue;
            int ly = 0;
            while (ly + 1 < re && qstart[ly + 1] <= qi) ++ly;
            const int rq = rpl + ly + 1;
            const int self_e = cellpos[rq * WOL_CP + 1] + (qi - qstart[ly]);
            // slot of this water in the sorted array: the tile's cells of a row are contiguous there
            const int self_slot = cs[(cz * ny + (y0 + ly)) * nx + x0] + (qi - qstart[ly]);
            int4 prop;
            prop.x = prop.y = prop.z = prop.w = 0;
            int kd3 = -1;
            if (!overflow) {
                unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu, k3 = 0xFFFFFFFFu, k4 = 0xFFFFFFFFu;
                const float4 me = ent[self_e];
                const int lx = (int)ent_k[self_e] - 1;  // 0-based cell of the water inside the tile
#pragma unroll
                for (int dzi = 0; dzi < 3; ++dzi) {
#pragma unroll
                    for (int dyi = 0; dyi < 3; ++dyi) {
                        const int r = dzi * rpl + ly + dyi;
                        const int b = cellpos[r * WOL_CP + lx], e = cellpos[r * WOL_CP + lx + 3];
                        if (dzi == 1 && dyi == 1)
                            wol_fast_span<true>(ent, b, e, me.x, me.y, me.z, self_e, g.tmax, k0, k1, k2, k3, k4);
                        else
                            wol_fast_span<false>(ent, b, e, me.x, me.y, me.z, self_e, g.tmax, k0, k1, k2, k3, k4);
                    }
                }
                // the proposal stands only if it is separated from the 5th candidate by more than the
                // fp32 error and is not saturated
                const unsigned d3 = k3 >> 11, d4 = k4 >> 11;
                const unsigned sat = (1u << 21) - 1u - (unsigned)g.eps_units;
                if (k3 != 0xFFFFFFFFu && d3 < sat && d4 - d3 > 2u * (unsigned)g.eps_units) {
                    prop.x = __float_as_int(ent[k0 & 0x7FFu].w); prop.y = __float_as_int(ent[k1 & 0x7FFu].w);
                    prop.z = __float_as_int(ent[k2 & 0x7FFu].w); prop.w = __float_as_int(ent[k3 & 0x7FFu].w);
                    kd3 = (int)d3;
                }
            }
            a.prop[frame_off + self_slot] = prop;
            a.prop_key[frame_off + self_slot] = kd3;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// kernel 2: certification + epilogue, one lane per water in sorted order.  Re-evaluates the four
// proposed neighbours from the fp64 coordinates with the reference's arithmetic, accepts when the
// 4th distance is inside the radius the 27-cell stencil guarantees (and agrees with the fp32 key),
// computes angles / q / histograms; everything else goes to the exact list.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WOL_BLOCK)
k_finish_fast(FastArgs a) {
    extern __shared__ int lds_hist[];
    int* lds_qh = a.out.qhist ? lds_hist : nullptr;
    int* lds_dh = a.out.dhist ? lds_hist + (a.out.qhist ? a.out.qbins : 0) : nullptr;
    if (lds_qh) wol_hist_zero(lds_qh, a.out.qbins);
    if (lds_dh) wol_hist_zero(lds_dh, a.out.dbins);
    __syncthreads();
    const int64_t total = (int64_t)a.F * a.n;
    for (int64_t w = (int64_t)blockIdx.x * WOL_BLOCK + threadIdx.x; w < total; w += (int64_t)gridDim.x * WOL_BLOCK) {
        const int f = (int)(w / a.n);
        const int s = (int)(w - (int64_t)f * a.n);
        const FrameGrid& g = a.grids[f];
        if (g.maxshell < 0) continue;  // all-pairs frame
        const double4* __restrict__ sp = a.sorted + (int64_t)f * a.n;
        const int kd3 = a.prop_key[w];
        bool slow = kd3 < 0;
        if (!slow) {
            const int4 pr = a.prop[w];
            const double4 mev = sp[s];
            const double c[3] = {mev.x, mev.y, mev.z};
            const double4 n0 = sp[pr.x], n1 = sp[pr.y], n2 = sp[pr.z], n3 = sp[pr.w];
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
            double rg = __builtin_huge_val();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double fr;
                (void)wol_cell_coord(c[k], g.box.ib[k], g.nc[k], &fr);
                rg = fmin(rg, g.cw[k] + fmin(fr, 1.0 - fr) * g.cw[k]);
            }
            rg -= 1e-6;
            // the exact 4th key must also agree with the fp32 proposal (guards the error bound itself)
            const double d3s = (t.d3 * t.d3) * g.S;
            if (!all_in || t.n != 4 || !(t.d3 <= rg) || fabs(d3s - (double)kd3) > (double)g.eps_units) {
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
                if (a.fast_q) wol_finish_water_fastq(c, t, N[0], N[1], N[2], N[3], g.box, a.out, orow, lds_qh, lds_dh);
                else wol_finish_water(c, t, N[0], N[1], N[2], N[3], g.box, a.out, orow, lds_qh, lds_dh);
            }
        }
        if (slow) {
            const int pos = atomicAdd(a.slow_count, 1);
            a.slow_list[pos] = ((unsigned long long)f << 32) | (unsigned)s;
        }
    }
    __syncthreads();
    if (lds_qh) wol_hist_flush(lds_qh, a.out.qbins, a.out.qhist);
    if (lds_dh) wol_hist_flush(lds_dh, a.out.dbins, a.out.dhist);
}

// exact search for the waters the filter could not certify
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
    for (int i = blockIdx.x * WOL_BLOCK + threadIdx.x; i < count; i += gridDim.x * WOL_BLOCK) {
        const unsigned long long e = slow_list[i];
        const int f = (int)(e >> 32), s = (int)(e & 0xFFFFFFFFu);
        const FrameGrid& g = a.grids[f];
        const double4* __restrict__ sp = a.sorted + (int64_t)f * a.n;
        const int* __restrict__ cs = a.cell_start + g.cell_base;
        const double4 v = sp[s];
        const double c[3] = {v.x, v.y, v.z};
        Top4 t;
        wol_search_exact_one(c, g, sp, cs, a.lowsq, a.highsq, a.high, t);
        const double4 n0 = sp[t.s0], n1 = sp[t.s1], n2 = sp[t.s2], n3 = sp[t.s3];
        const double N0[3] = {n0.x, n0.y, n0.z}, N1[3] = {n1.x, n1.y, n1.z};
        const double N2[3] = {n2.x, n2.y, n2.z}, N3[3] = {n3.x, n3.y, n3.z};
        wol_finish_water(c, t, N0, N1, N2, N3, g.box, a.out, (int64_t)f * a.ns + (int)__double_as_longlong(v.w),
                         lds_qh, lds_dh);
    }
    __syncthreads();
    if (lds_qh) wol_hist_flush(lds_qh, a.out.qbins, a.out.qhist);
    if (lds_dh) wol_hist_flush(lds_dh, a.out.dbins, a.out.dhist);
}
