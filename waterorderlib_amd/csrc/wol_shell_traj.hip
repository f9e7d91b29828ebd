// wol_shell_traj.hip -- batched (trajectory) forms of the shell kernels: the frame loops of threeBodyCalc
// (orderParam_lib.py:1312-1353: wp.getCosAngs + wp.tetrahedralMetrics per frame and population) and lsiCalc
// (:1618-1640: wp.getLSI, np.mean / np.var per frame and population), plus the sub-population queries of tetOrderCalc
// (:1476-1480).  Included by wol_api.hip.
//
// All of them consume a batch that sort_batch has cell-sorted (fp64 records, cells as wide as the shell's reach) and
// reduce on the device to what the reference's callers keep: integer histograms, counts, and sums as exact fixed-point
// integers -- nothing per-water crosses the host link unless asked for, and every result is independent of the order in
// which waters and frames are processed.
//
// Results are laid out by SEGMENT: segment (t, 0) = all waters of frame t, segment (t, j) = population j of frame t
// (its query points are given by position, as the reference hands pos[subInds[t][j-1]] to the per-frame routine);
// seg = t * (npops + 1) + j.
#pragma once

struct TrajShellArgs {
    const FrameGrid* grids;      // [F] of this batch
    const double4* sorted;       // [F][n] cell-sorted (frames without a grid: one cell, all-pairs walk)
    const int* cell_start;
    int n, F;
    int f0;                      // first frame of the batch in the call (segments are numbered over the call)
    int npops;
    double lowsq, highsq;
    double reach;                // largest distance that matters to the kernel (cut-off; LSI: + 3.7 A): decides rint vs anint
    // population queries: points [pt_off[t * npops + j], pt_off[t * npops + j + 1]) belong to (frame t, population j + 1)
    const double* pts;           // [total][3]
    const long long* pt_off;     // [T * npops + 1]
    int nbins;
    double hist_max;             // LSI: upper edge of the histogram range (lower edge 0)
    unsigned long long* hist;    // cos-angs: [nseg][nbins]; LSI: [npops + 1][nbins]
    unsigned long long* stat;    // [nseg][8] integer accumulators (see the kernels)
    double* per_water;           // LSI: optional [T][n] (NaN where the reference skips the water), original water order
    unsigned long long* slow_list;   // (frame in batch << 32 | sorted slot) of waters whose shell outgrew the LDS rows
    int* slow_count;
    int slow_cap;
    int* error_flag;
};

#define WOL_TRAJ_BLOCK 256

// block -> (frames, slots): G blocks share a frame, B / G frames are in flight at a time (few blocks per frame when the
// batch has many frames, so a frame's histogram leaves LDS in few flushes)
struct TrajBlockMap { int g, G, concurrent; };
__device__ __forceinline__ TrajBlockMap wol_traj_block_map(int n, int F) {
    const int B = (int)gridDim.x, need = (n + WOL_TRAJ_BLOCK - 1) / WOL_TRAJ_BLOCK;
    int G = B / max(F, 1);
    G = max(1, min(G, need));
    TrajBlockMap m;
    m.G = G;
    m.concurrent = max(1, B / G);
    m.g = (int)blockIdx.x / G;
    return m;
}

// minimum image with anint as the reference (round half away) or, where the caller has checked that no candidate in reach
// can sit at exactly half a box length (every periodic edge longer than twice the reach), with rint: one instruction
// instead of six, the same value for every candidate that can pass the caller's cut
template <bool RINT>
__device__ __forceinline__ double wol_traj_minimg(double d, double L, double ib) {
    return RINT ? d - L * rint(d * ib) : wol_minimg1(d, L, ib);
}

__device__ __forceinline__ bool wol_traj_rint_ok(const FrameGrid& g, double reach) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (g.box.ib[k] != 0.0 && !(0.5 * g.box.L[k] > reach * 1.000001 + 1e-6)) ok = false;
    return ok;
}

// candidates of a query point c in frame grid g: f(slot, d2, record) for every water whose exact minimum-image d2 can matter
template <bool RINT, typename Fn>
__device__ __forceinline__ void wol_traj_candidates_t(const FrameGrid& g, const double4* __restrict__ sp, const int* __restrict__ cs,
                                                      int n, const double c[3], Fn f) {
    if (g.maxshell < 0) {
        for (int j = 0; j < n; ++j) {
            const double4 v = sp[j];
            const double dx = wol_minimg1(v.x - c[0], g.box.L[0], g.box.ib[0]);
            const double dy = wol_minimg1(v.y - c[1], g.box.L[1], g.box.ib[1]);
            const double dz = wol_minimg1(v.z - c[2], g.box.L[2], g.box.ib[2]);
            f(j, wol_sumsq(dx, dy, dz), v);
        }
        return;
    }
    int cc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) cc[k] = wol_cell_coord(c[k], g.box.ib[k], g.nc[k], nullptr);
    // the cells of one x run are contiguous in the sorted array: a (dz, dy) row of the walk is one span of records, or two
    // where the run wraps around the box -- a fifth of the cell-table look-ups of a cell-by-cell walk; two records in
    // flight (four cost a wave per SIMD in registers and gained nothing)
    const int nx = g.nc[0];
    const int xa = cc[0] - g.lo[0], xb = cc[0] + g.hi[0];       // first and last cell of the run, before wrapping
    const bool whole = g.lo[0] + g.hi[0] + 1 >= nx;
    for (int dz = -g.lo[2]; dz <= g.hi[2]; ++dz) {
        const int cz = wol_wrap_cell(cc[2] + dz, g.nc[2]);
        for (int dy = -g.lo[1]; dy <= g.hi[1]; ++dy) {
            const int cy = wol_wrap_cell(cc[1] + dy, g.nc[1]);
            const int* __restrict__ row = cs + (cz * g.nc[1] + cy) * nx;
            // piece 0: [p0a, p0b), piece 1: [p1a, p1b) (empty unless the run wraps)
            int p0a, p0b, p1a = 0, p1b = 0;
            if (whole) { p0a = 0; p0b = nx; }
            else if (xa < 0) { p0a = xa + nx; p0b = nx; p1a = 0; p1b = xb + 1; }
            else if (xb >= nx) { p0a = xa; p0b = nx; p1a = 0; p1b = xb - nx + 1; }
            else { p0a = xa; p0b = xb + 1; }
#pragma unroll
            for (int piece = 0; piece < 2; ++piece) {
                const int ca = piece ? p1a : p0a, cb = piece ? p1b : p0b;
                if (cb <= ca) continue;
                const int b = row[ca], e = row[cb];
                for (int j = b; j < e; j += 2) {
                    double4 v[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) v[u] = sp[min(j + u, e - 1)];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (j + u < e) {
                            const double ex = wol_traj_minimg<RINT>(v[u].x - c[0], g.box.L[0], g.box.ib[0]);
                            const double ey = wol_traj_minimg<RINT>(v[u].y - c[1], g.box.L[1], g.box.ib[1]);
                            const double ez = wol_traj_minimg<RINT>(v[u].z - c[2], g.box.L[2], g.box.ib[2]);
                            f(j + u, wol_sumsq(ex, ey, ez), v[u]);
                        }
                    }
                }
            }
        }
    }
}

template <typename Fn>
__device__ __forceinline__ void wol_traj_candidates(const FrameGrid& g, const double4* __restrict__ sp, const int* __restrict__ cs,
                                                    int n, const double c[3], bool use_rint, Fn f) {
    if (use_rint) wol_traj_candidates_t<true>(g, sp, cs, n, c, f);
    else wol_traj_candidates_t<false>(g, sp, cs, n, c, f);
}

// neighbour as tetraCosAng sees it (waterlib.f90:881-883): centre + minimum-image vector
__device__ __forceinline__ void wol_traj_reimage(const double c[3], const double4 p, const Box3& b, double out[3]) {
    out[0] = c[0] + wol_minimg1(p.x - c[0], b.L[0], b.ib[0]);
    out[1] = c[1] + wol_minimg1(p.y - c[1], b.L[1], b.ib[1]);
    out[2] = c[2] + wol_minimg1(p.z - c[2], b.L[2], b.ib[2]);
}

// what tetrahedralMetrics (water_properties.py:314-342) keeps of one angle: its np.histogram bin over [0, 180], whether it
// lies in [100, 120] and, if so, cos(angle * pi / 180) and its square as exact fixed-point integers
struct AngAcc { unsigned long long n_ang, n_tet; long long ch, cl, c2h, c2l; };
__device__ __forceinline__ void wol_traj_angle(double ang, int nbins, int* lds_hist, unsigned long long* ghist, AngAcc& acc) {
    const int bin = wol_np_hist_bin(ang, nbins, 0.0, 180.0);
    if (bin >= 0) {
        if (lds_hist) atomicAdd(&lds_hist[bin], 1);
        else atomicAdd(&ghist[bin], 1ull);
    }
    acc.n_ang += 1;
    if (ang >= 100.0 && ang <= 120.0) {
        const double cs_ = cos(ang * WOL_PI / 180.0);
        long long h, l;
        wol_fix_split_fast(cs_, &h, &l); acc.ch += h; acc.cl += l;
        wol_fix_split_fast(cs_ * cs_, &h, &l); acc.c2h += h; acc.c2l += l;
        acc.n_tet += 1;
    }
}

#define WOL_TRAJ_KMAX 32     // shell neighbours per water kept in LDS (a liquid's 3.4 A shell holds 3-7)

// ---------------------------------------------------------------------------------------------
// threeBodyCalc, all waters of every frame.  Phase 1, one lane per water: walk the cells in reach, keep the sorted slots
// of the waters inside the shell in the lane's LDS row.  Phase 2: the wave's (water, neighbour pair) items -- k (k - 1) / 2
// per water, 3 to 30 -- are dealt evenly to its 64 lanes through a prefix sum over the lanes' pair counts (a lane per
// WATER would idle most lanes behind the water with the fullest shell); every item re-images its two neighbours about
// the centre and evaluates CosAngle3 as wol_cos_angs does (same device function on the same operands, so the integer
// results are those of np.histogram over wol_cos_angs's angles).
// stat[seg] = {angles, angles in [100, 120], sum cos (hi, lo), sum cos^2 (hi, lo), -, -}
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WOL_TRAJ_BLOCK, 4)   // 4 blocks (16 waves) per CU: 128 VGPRs, 40 KB of LDS each
k_cos_angs_traj(TrajShellArgs a) {
    extern __shared__ int lds_dyn[];                         // [nbins] histogram of the frame in flight
    __shared__ int nb_slot[WOL_TRAJ_BLOCK * WOL_TRAJ_KMAX];  // [wave][k][lane]: slot of neighbour k of the lane's water
    __shared__ double cq[WOL_TRAJ_BLOCK * 3];                // [wave][comp][lane]: the lanes' centres
    __shared__ int pre[WOL_TRAJ_BLOCK + 4];                  // [wave][lane + 1]: pair items before the lane's
    __shared__ unsigned char kq[WOL_TRAJ_BLOCK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* __restrict__ my_slots = nb_slot + wave * 64 * WOL_TRAJ_KMAX;
    double* __restrict__ my_c = cq + wave * 192;
    int* __restrict__ my_pre = pre + wave * 65;
    const TrajBlockMap bm = wol_traj_block_map(a.n, a.F);
    for (int k = tid; k < a.nbins; k += WOL_TRAJ_BLOCK) lds_dyn[k] = 0;
    __syncthreads();
    for (int f = bm.g; f < a.F && bm.g < bm.concurrent; f += bm.concurrent) {
        const FrameGrid& g = a.grids[f];
        const double4* __restrict__ sp = a.sorted + (int64_t)f * a.n;
        const int* __restrict__ cs = a.cell_start + g.cell_base;
        const bool rint_ok = wol_traj_rint_ok(g, a.reach);
        AngAcc acc = {0ull, 0ull, 0ll, 0ll, 0ll, 0ll};
        for (int s0 = ((int)blockIdx.x % bm.G) * WOL_TRAJ_BLOCK; s0 < a.n; s0 += bm.G * WOL_TRAJ_BLOCK) {
            const int s = s0 + tid;
            const bool live = s < a.n;
            double c[3] = {0.0, 0.0, 0.0};
            int k = 0;
            if (live) {
                const double4 v = sp[s];
                c[0] = v.x; c[1] = v.y; c[2] = v.z;
                wol_traj_candidates(g, sp, cs, a.n, c, rint_ok, [&](int slot, double d2, const double4&) {
                    if (d2 > a.lowsq && d2 <= a.highsq) {
                        if (k < WOL_TRAJ_KMAX) my_slots[k * 64 + lane] = slot;
                        ++k;
                    }
                });
            }
            if (k > WOL_TRAJ_KMAX) {   // a shell denser than the LDS rows: the slow kernel lists it again
                const int pos = atomicAdd(a.slow_count, 1);
                if (pos < a.slow_cap) a.slow_list[pos] = ((unsigned long long)f << 32) | (unsigned)s;
                else atomicExch(a.error_flag, 1);
                k = 0;
            }
            my_c[lane] = c[0]; my_c[64 + lane] = c[1]; my_c[128 + lane] = c[2];
            kq[tid] = (unsigned char)k;
            const int np_ = k * (k - 1) / 2;
            int inc = np_;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(inc, off, 64);
                if (lane >= off) inc += t;
            }
            my_pre[lane + 1] = inc;
            if (lane == 0) my_pre[0] = 0;
            __builtin_amdgcn_wave_barrier();
            const int total = __shfl(inc, 63, 64);
            for (int it = lane; it < total; it += 64) {
                int w = 0;   // owner: largest w with pre[w] <= it
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1)
                    if (w + step < 64 && my_pre[w + step] <= it) w += step;
                const int kk = (int)kq[wave * 64 + w];
                int r = it - my_pre[w];
                // pair number r of the kk (kk - 1) / 2 pairs (u < v) in row-major order: u = row, v = column
                int u = 0, rowlen = kk - 1;
                while (r >= rowlen) { r -= rowlen; ++u; --rowlen; }
                const int v = u + 1 + r;
                const double cw[3] = {my_c[w], my_c[64 + w], my_c[128 + w]};
                const double4 pu = sp[my_slots[u * 64 + w]], pv = sp[my_slots[v * 64 + w]];
                double P1[3], P2[3];
                wol_traj_reimage(cw, pu, g.box, P1);
                wol_traj_reimage(cw, pv, g.box, P2);
                wol_traj_angle(wol_cos_angle3(P1, cw, P2), a.nbins, lds_dyn, nullptr, acc);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // the frame is done for this block: flush its histogram and its share of the sums
        __syncthreads();
        const int seg = (a.f0 + f) * (a.npops + 1);
        unsigned long long* __restrict__ gh = a.hist + (size_t)seg * a.nbins;
        for (int k = tid; k < a.nbins; k += WOL_TRAJ_BLOCK) {
            const int v = lds_dyn[k];
            if (v) { atomicAdd(&gh[k], (unsigned long long)v); lds_dyn[k] = 0; }
        }
        long long vals[6] = {(long long)acc.n_ang, (long long)acc.n_tet, acc.ch, acc.cl, acc.c2h, acc.c2l};
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) vals[j] += __shfl_xor(vals[j], off, 64);
            if (lane == 0 && vals[j] != 0) atomicAdd(&a.stat[(size_t)seg * 8 + j], (unsigned long long)vals[j]);
        }
        __syncthreads();
    }
}

// the same for query points that keep no list: the angle of neighbour pair (u, v) is found by walking the candidates
// again for every in-shell u -- O(k x candidates), used for (a) waters whose shell outgrew WOL_TRAJ_KMAX and (b) the
// population points, both few.  One lane per query; histogram and sums go straight to global memory.
__device__ __forceinline__ void wol_traj_angles_rewalk(const TrajShellArgs& a, const FrameGrid& g, const double4* __restrict__ sp,
                                                       const int* __restrict__ cs, const double c[3], int seg) {
    AngAcc acc = {0ull, 0ull, 0ll, 0ll, 0ll, 0ll};
    unsigned long long* __restrict__ gh = a.hist + (size_t)seg * a.nbins;
    int iu = 0;
    wol_traj_candidates(g, sp, cs, a.n, c, false, [&](int, double d2u, const double4& pu) {
        if (!(d2u > a.lowsq && d2u <= a.highsq)) return;
        double P1[3];
        wol_traj_reimage(c, pu, g.box, P1);
        int iv = 0;
        wol_traj_candidates(g, sp, cs, a.n, c, false, [&](int, double d2v, const double4& pv) {
            if (!(d2v > a.lowsq && d2v <= a.highsq)) return;
            if (iv > iu) {
                double P2[3];
                wol_traj_reimage(c, pv, g.box, P2);
                wol_traj_angle(wol_cos_angle3(P1, c, P2), a.nbins, nullptr, gh, acc);
            }
            ++iv;
        });
        ++iu;
    });
    unsigned long long* st = a.stat + (size_t)seg * 8;
    if (acc.n_ang) atomicAdd(&st[0], acc.n_ang);
    if (acc.n_tet) {
        atomicAdd(&st[1], acc.n_tet);
        atomicAdd(&st[2], (unsigned long long)acc.ch); atomicAdd(&st[3], (unsigned long long)acc.cl);
        atomicAdd(&st[4], (unsigned long long)acc.c2h); atomicAdd(&st[5], (unsigned long long)acc.c2l);
    }
}

__global__ void __launch_bounds__(WOL_TRAJ_BLOCK)
k_cos_angs_traj_slow(TrajShellArgs a) {
    const int count = min(*a.slow_count, a.slow_cap);
    for (int i = blockIdx.x * WOL_TRAJ_BLOCK + threadIdx.x; i < count; i += gridDim.x * WOL_TRAJ_BLOCK) {
        const unsigned long long e = a.slow_list[i];
        const int f = (int)(e >> 32), s = (int)(e & 0xFFFFFFFFu);
        const FrameGrid& g = a.grids[f];
        const double4* __restrict__ sp = a.sorted + (int64_t)f * a.n;
        const double4 v = sp[s];
        const double c[3] = {v.x, v.y, v.z};
        wol_traj_angles_rewalk(a, g, sp, a.cell_start + g.cell_base, c, (a.f0 + f) * (a.npops + 1));
    }
}

// segment of population point p of this batch: binary search in the offsets of the batch's (frame, population) pairs
__device__ __forceinline__ int wol_traj_point_segment(const TrajShellArgs& a, long long p) {
    int lo = a.f0 * a.npops, hi = (a.f0 + a.F) * a.npops;   // pt_off[lo] <= p < pt_off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.pt_off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(WOL_TRAJ_BLOCK)
k_cos_angs_traj_points(TrajShellArgs a) {
    const long long p0 = a.pt_off[a.f0 * a.npops], p1 = a.pt_off[(a.f0 + a.F) * a.npops];
    for (long long p = p0 + (long long)blockIdx.x * WOL_TRAJ_BLOCK + threadIdx.x; p < p1; p += (long long)gridDim.x * WOL_TRAJ_BLOCK) {
        const int sp_ = wol_traj_point_segment(a, p);
        const int t = sp_ / a.npops, j = sp_ - t * a.npops, f = t - a.f0;
        const FrameGrid& g = a.grids[f];
        const double c[3] = {a.pts[3 * p], a.pts[3 * p + 1], a.pts[3 * p + 2]};
        wol_traj_angles_rewalk(a, g, a.sorted + (int64_t)f * a.n, a.cell_start + g.cell_base, c, t * (a.npops + 1) + 1 + j);
    }
}

// ---------------------------------------------------------------------------------------------
// lsiCalc.  One lane per query: the minimum-image distances of the shell (lowCut, highCut] and of the "next" neighbour in
// (highCut, highCut + 3.7] -- chosen by its RAW distance, first minimum in ascending original index, the reference's
// quirk (water_properties.py:289-291) -- are sorted in the lane's LDS row; LSI = np.var of the gaps with numpy's pairwise
// sums (wol_np_sum_small), i.e. the value wol_lsi_frame returns, bit for bit.  Waters the reference skips (fewer than two
// shell neighbours, or no next neighbour) contribute nothing.
// stat[seg] = {len(lsiVals), sum LSI (hi, lo), sum LSI^2 (hi, lo), -, -, -}; hist[population][nbins] over [0, hist_max]
// ---------------------------------------------------------------------------------------------
#define WOL_TRAJ_LSI_ROW 16    // least doubles per lane in LDS: shell + next neighbour (a liquid's 3.7 A shell holds 4-9); the host
#define WOL_TRAJ_LSI_ROWMAX 72 // sizes the rows for twice the shell the mean density predicts, up to this
#define WOL_TRAJ_LSI_SLOWROW 1024

struct LsiAcc { unsigned long long n; long long h, l, h2, l2; };

__device__ __forceinline__ double wol_traj_lsi_of_row(double* __restrict__ dist, int stride, int m) {
    // dist[0 .. m) (stride apart): np.sort, successive gaps, np.var (ddof = 0)
    for (int i = 1; i < m; ++i) {
        const double x = dist[i * stride];
        int p = i;
        while (p > 0 && dist[(p - 1) * stride] > x) { dist[p * stride] = dist[(p - 1) * stride]; --p; }
        dist[p * stride] = x;
    }
    const int nd = m - 1;
    double tmp[8];
    // numpy's pairwise sum for fewer than 128 terms (wol_np_sum_small) over a strided row
    auto row_sum = [&](bool squares, double mean) -> double {
        auto term = [&](int i) -> double {
            const double gap = dist[(i + 1) * stride] - dist[i * stride];
            if (!squares) return gap;
            const double x = gap - mean;
            return x * x;
        };
        if (nd < 8) {
            double r = -0.0;
            for (int i = 0; i < nd; ++i) r += term(i);
            return r;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) tmp[k] = term(k);
        int i;
        for (i = 8; i < nd - (nd % 8); i += 8)
#pragma unroll
            for (int k = 0; k < 8; ++k) tmp[k] += term(i + k);
        double res = ((tmp[0] + tmp[1]) + (tmp[2] + tmp[3])) + ((tmp[4] + tmp[5]) + (tmp[6] + tmp[7]));
        for (; i < nd; ++i) res += term(i);
        return res;
    };
    const double mean = row_sum(false, 0.0) / (double)nd;
    return row_sum(true, mean) / (double)nd;
}

// shell walk of one query: the shell's minimum-image distances into row[] (stride apart, at most cap - 1 of them), returns k
// (may exceed cap - 1) and the next neighbour's minimum-image d2 (have_next).  Inside the walk only squared distances are
// handled: the square roots are taken once per kept distance afterwards, and the next neighbour -- the reference's argmin of
// the RAW distances sqrt(sum (p - c)^2), first minimum in ascending original index -- is tracked by raw squared distance;
// only where two squares are within a few ulp of each other (their roots could round to the same double, which the
// reference would see as a tie) are the roots themselves compared.
__device__ __forceinline__ int wol_traj_lsi_walk(const TrajShellArgs& a, const FrameGrid& g, const double4* __restrict__ sp,
                                                 const int* __restrict__ cs, const double c[3], double high2sq,
                                                 double* __restrict__ row, int stride, int cap, bool* have_next, double* next_d2) {
    int k = 0;
    bool hn = false;
    double next_raw2 = 0.0, nd2 = 0.0;
    int next_idx = 0x7fffffff;
    wol_traj_candidates(g, sp, cs, a.n, c, wol_traj_rint_ok(g, a.reach), [&](int, double d2, const double4& p) {
        if (d2 > a.lowsq && d2 <= a.highsq) {
            if (k < cap - 1) row[k * stride] = d2;
            ++k;
        }
        // (the raw squared distance is never below the minimum-image one: a candidate whose d2 already lies above the best
        //  raw square so far -- by more than the tie band below -- cannot take its place, and most cannot)
        if (d2 > a.highsq && d2 <= high2sq && (!hn || d2 <= next_raw2 * (1.0 + 4e-15))) {
            const double rx = p.x - c[0], ry = p.y - c[1], rz = p.z - c[2];
            const double raw2 = wol_sumsq(rx, ry, rz);
            const int idx = (int)__double_as_longlong(p.w);
            bool better;
            if (!hn) better = true;
            else if (fabs(raw2 - next_raw2) > 1e-15 * next_raw2) better = raw2 < next_raw2;
            else {
                const double ra = sqrt(raw2), rb = sqrt(next_raw2);
                better = ra < rb || (ra == rb && idx < next_idx);
            }
            if (better) { hn = true; next_raw2 = raw2; next_idx = idx; nd2 = d2; }
        }
    });
    const int kk = min(k, cap - 1);
    for (int i = 0; i < kk; ++i) row[i * stride] = sqrt(row[i * stride]);   // lsiDists, waterlib.f90:916
    *have_next = hn;
    *next_d2 = nd2;
    return k;
}

__device__ __forceinline__ void wol_traj_lsi_record(const TrajShellArgs& a, double lsi, int pop, int* lds_hist, LsiAcc& acc) {
    long long h, l;
    wol_fix_split_fast(lsi, &h, &l); acc.h += h; acc.l += l;
    wol_fix_split_fast(lsi * lsi, &h, &l); acc.h2 += h; acc.l2 += l;
    acc.n += 1;
    const int bin = wol_np_hist_bin(lsi, a.nbins, 0.0, a.hist_max);
    if (bin >= 0) {
        if (lds_hist) atomicAdd(&lds_hist[bin], 1);
        else atomicAdd(&a.hist[(size_t)pop * a.nbins + bin], 1ull);
    }
}

__device__ __forceinline__ void wol_traj_lsi_flush(const TrajShellArgs& a, int seg, const LsiAcc& acc) {
    if (!acc.n) return;
    unsigned long long* st = a.stat + (size_t)seg * 8;
    atomicAdd(&st[0], acc.n);
    atomicAdd(&st[1], (unsigned long long)acc.h); atomicAdd(&st[2], (unsigned long long)acc.l);
    atomicAdd(&st[3], (unsigned long long)acc.h2); atomicAdd(&st[4], (unsigned long long)acc.l2);
}

__global__ void __launch_bounds__(WOL_TRAJ_BLOCK)
k_lsi_traj(TrajShellArgs a, double high2sq, int row_len) {
    extern __shared__ double lds_rows[];                               // [row_len][thread] distance rows, then the histogram
    double* __restrict__ rows = lds_rows;
    int* __restrict__ lds_dyn = reinterpret_cast<int*>(lds_rows + (size_t)row_len * WOL_TRAJ_BLOCK);   // [nbins] (population 0, all frames)
    const int tid = threadIdx.x, lane = tid & 63;
    const TrajBlockMap bm = wol_traj_block_map(a.n, a.F);
    for (int k = tid; k < a.nbins; k += WOL_TRAJ_BLOCK) lds_dyn[k] = 0;
    __syncthreads();
    for (int f = bm.g; f < a.F && bm.g < bm.concurrent; f += bm.concurrent) {
        const FrameGrid& g = a.grids[f];
        const double4* __restrict__ sp = a.sorted + (int64_t)f * a.n;
        const int* __restrict__ cs = a.cell_start + g.cell_base;
        LsiAcc acc = {0ull, 0ll, 0ll, 0ll, 0ll};
        for (int s0 = ((int)blockIdx.x % bm.G) * WOL_TRAJ_BLOCK; s0 < a.n; s0 += bm.G * WOL_TRAJ_BLOCK) {
            const int s = s0 + tid;
            if (s >= a.n) continue;
            const double4 v = sp[s];
            const double c[3] = {v.x, v.y, v.z};
            bool have_next;
            double next_d2;
            const int k = wol_traj_lsi_walk(a, g, sp, cs, c, high2sq, rows + tid, WOL_TRAJ_BLOCK, row_len, &have_next, &next_d2);
            double res = __builtin_nan("");
            if (k >= row_len - 1) {
                const int pos = atomicAdd(a.slow_count, 1);
                if (pos < a.slow_cap) a.slow_list[pos] = ((unsigned long long)f << 32) | (unsigned)s;
                else atomicExch(a.error_flag, 1);
                continue;   // (its per-water value is written by the slow kernel)
            }
            if (k > 1 && have_next) {
                rows[k * WOL_TRAJ_BLOCK + tid] = sqrt(next_d2);
                res = wol_traj_lsi_of_row(rows + tid, WOL_TRAJ_BLOCK, k + 1);
                wol_traj_lsi_record(a, res, 0, lds_dyn, acc);
            }
            if (a.per_water) a.per_water[(int64_t)(a.f0 + f) * a.n + (int)__double_as_longlong(v.w)] = res;
        }
        long long vals[5] = {(long long)acc.n, acc.h, acc.l, acc.h2, acc.l2};
        const int seg = (a.f0 + f) * (a.npops + 1);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) vals[j] += __shfl_xor(vals[j], off, 64);
            if (lane == 0 && vals[j] != 0) atomicAdd(&a.stat[(size_t)seg * 8 + j], (unsigned long long)vals[j]);
        }
    }
    __syncthreads();
    for (int k = tid; k < a.nbins; k += WOL_TRAJ_BLOCK) {
        const int v = lds_dyn[k];
        if (v) atomicAdd(&a.hist[k], (unsigned long long)v);
    }
}

// population points: one lane per point, the same LDS rows; histogram and sums go straight to global memory (few points)
__global__ void __launch_bounds__(WOL_TRAJ_BLOCK)
k_lsi_traj_points(TrajShellArgs a, double high2sq, int row_len, double* __restrict__ pt_out) {
    extern __shared__ double lds_rows[];                               // [row_len][thread]
    double* __restrict__ rows = lds_rows;
    const int tid = threadIdx.x;
    const long long p0 = a.pt_off[a.f0 * a.npops], p1 = a.pt_off[(a.f0 + a.F) * a.npops];
    for (long long p = p0 + (long long)blockIdx.x * WOL_TRAJ_BLOCK + tid; p < p1; p += (long long)gridDim.x * WOL_TRAJ_BLOCK) {
        const int sp_ = wol_traj_point_segment(a, p);
        const int t = sp_ / a.npops, j = sp_ - t * a.npops, f = t - a.f0;
        const FrameGrid& g = a.grids[f];
        const double c[3] = {a.pts[3 * p], a.pts[3 * p + 1], a.pts[3 * p + 2]};
        bool have_next;
        double next_d2;
        const int k = wol_traj_lsi_walk(a, g, a.sorted + (int64_t)f * a.n, a.cell_start + g.cell_base, c, high2sq, rows + tid,
                                        WOL_TRAJ_BLOCK, row_len, &have_next, &next_d2);
        if (k >= row_len - 1) {
            const int pos = atomicAdd(a.slow_count, 1);
            if (pos < a.slow_cap) a.slow_list[pos] = (1ull << 63) | (unsigned long long)p;
            else atomicExch(a.error_flag, 1);
            continue;
        }
        double res = __builtin_nan("");
        if (k > 1 && have_next) {
            rows[k * WOL_TRAJ_BLOCK + tid] = sqrt(next_d2);
            res = wol_traj_lsi_of_row(rows + tid, WOL_TRAJ_BLOCK, k + 1);
            LsiAcc acc = {0ull, 0ll, 0ll, 0ll, 0ll};
            wol_traj_lsi_record(a, res, 1 + j, nullptr, acc);
            wol_traj_lsi_flush(a, t * (a.npops + 1) + 1 + j, acc);
        }
        if (pt_out) pt_out[p] = res;
    }
}

// queries whose shell outgrew the LDS row (waters: frame << 32 | slot; population points: bit 63 | point): one lane per
// query, each lane of the grid owning one row of WOL_TRAJ_LSI_SLOWROW doubles in global scratch (so the list may hold every
// water of the batch: a droplet far denser than the box's mean puts thousands there)
__global__ void __launch_bounds__(WOL_TRAJ_BLOCK)
k_lsi_traj_slow(TrajShellArgs a, double high2sq, double* __restrict__ scratch, double* __restrict__ pt_out) {
    const int count = min(*a.slow_count, a.slow_cap);
    for (int i = blockIdx.x * WOL_TRAJ_BLOCK + threadIdx.x; i < count; i += gridDim.x * WOL_TRAJ_BLOCK) {
        const unsigned long long e = a.slow_list[i];
        const bool is_point = (e >> 63) != 0;
        int f, pop = 0, seg;
        double c[3];
        long long p = 0;
        int orig = 0;
        if (is_point) {
            p = (long long)(e & 0x7FFFFFFFFFFFFFFFull);
            const int sp_ = wol_traj_point_segment(a, p);
            const int t = sp_ / a.npops, j = sp_ - t * a.npops;
            f = t - a.f0; pop = 1 + j; seg = t * (a.npops + 1) + 1 + j;
            c[0] = a.pts[3 * p]; c[1] = a.pts[3 * p + 1]; c[2] = a.pts[3 * p + 2];
        } else {
            f = (int)(e >> 32);
            const double4 v = a.sorted[(int64_t)f * a.n + (int)(e & 0xFFFFFFFFu)];
            c[0] = v.x; c[1] = v.y; c[2] = v.z;
            orig = (int)__double_as_longlong(v.w);
            seg = (a.f0 + f) * (a.npops + 1);
        }
        const FrameGrid& g = a.grids[f];
        double* __restrict__ row = scratch + (size_t)(blockIdx.x * WOL_TRAJ_BLOCK + threadIdx.x) * WOL_TRAJ_LSI_SLOWROW;
        bool have_next;
        double next_d2;
        const int k = wol_traj_lsi_walk(a, g, a.sorted + (int64_t)f * a.n, a.cell_start + g.cell_base, c, high2sq, row, 1,
                                        WOL_TRAJ_LSI_SLOWROW, &have_next, &next_d2);
        double res = __builtin_nan("");
        if (k >= WOL_TRAJ_LSI_SLOWROW - 1) {
            atomicExch(a.error_flag, 2);
        } else if (k > 1 && have_next) {
            row[k] = sqrt(next_d2);
            res = wol_traj_lsi_of_row(row, 1, k + 1);
            LsiAcc acc = {0ull, 0ll, 0ll, 0ll, 0ll};
            wol_traj_lsi_record(a, res, pop, nullptr, acc);
            wol_traj_lsi_flush(a, seg, acc);
        }
        if (is_point) { if (pt_out) pt_out[p] = res; }
        else if (a.per_water) a.per_water[(int64_t)(a.f0 + f) * a.n + orig] = res;
    }
}

// ---------------------------------------------------------------------------------------------
// tetOrderCalc's populations (orderParam_lib.py:1476-1480): q of every population point against the frame's waters --
// getOrderParamq's subPos != Pos branch (water_properties.py:365-366).  One WAVE per point, the exact search of
// k_search_exact_list (cells of each shell dealt to the lanes, butterfly merge, same termination guarantee) and the
// reference's chain for the angles and q (wol_finish_water); a point that coincides with a water does not see itself
// (d2 > lowCut^2 = 0), so a population made of water oxygens gets exactly the q those waters have in the frame.
// stat[seg] = {points, sum q (hi, lo), sum q^2 (hi, lo)}; hist[population][qbins]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WOL_BLOCK)
k_q_traj_points(TrajShellArgs a, double high, int qbins, double* __restrict__ pt_q) {
    const long long p0 = a.pt_off[a.f0 * a.npops], p1 = a.pt_off[(a.f0 + a.F) * a.npops];
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * WOL_BLOCK + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * WOL_BLOCK) >> 6;
    for (long long p = p0 + wave; p < p1; p += nwaves) {
        const int sp_ = wol_traj_point_segment(a, p);
        const int t = sp_ / a.npops, j = sp_ - t * a.npops, f = t - a.f0;
        const FrameGrid& g = a.grids[f];
        const double4* __restrict__ sp = a.sorted + (int64_t)f * a.n;
        const int* __restrict__ cs = a.cell_start + g.cell_base;
        const double c[3] = {a.pts[3 * p], a.pts[3 * p + 1], a.pts[3 * p + 2]};
        Top4 mine, all;
        top4_init(mine);
        top4_init(all);
        if (g.maxshell < 0) {   // no grid: every water, dealt to the lanes
            for (int jj = lane; jj < a.n; jj += 64) {
                const double4 q = sp[jj];
                double dist;
                bool in_mask;
                if (wol_eval_pair(c, q.x, q.y, q.z, g.box, a.lowsq, a.highsq, mine.bound2, &dist, &in_mask))
                    top4_insert(mine, dist, (int)__double_as_longlong(q.w), jj);
            }
            all = mine;
            wol_top4_wave_merge(all);
        } else {
            double fr[3];
            int cc[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) cc[k] = wol_cell_coord(c[k], g.box.ib[k], g.nc[k], &fr[k]);
            for (int shell = 0; shell <= g.maxshell; ++shell) {
                if (shell == 0 && g.maxshell >= 1) continue;   // shells 0 and 1 together are the 27-cell cube
                const int w = 2 * shell + 1;
                const float inv_w = 1.0f / (float)w;
                const int ncube = w * w * w;
                for (int idx = lane; idx < ncube; idx += 64) {
                    const int iz = (int)(((float)idx + 0.5f) * inv_w * inv_w);
                    const int rem = idx - iz * w * w;
                    const int iy = (int)(((float)rem + 0.5f) * inv_w);
                    const int ix = rem - iy * w;
                    const int dx = ix - shell, dy = iy - shell, dz = iz - shell;
                    const bool on_shell = (abs(dx) == shell) || (abs(dy) == shell) || (abs(dz) == shell);
                    if (!(on_shell || shell == 1)) continue;
                    if (dx < -g.lo[0] || dx > g.hi[0] || dy < -g.lo[1] || dy > g.hi[1] || dz < -g.lo[2] || dz > g.hi[2]) continue;
                    const int cell = (wol_wrap_cell(cc[2] + dz, g.nc[2]) * g.nc[1] + wol_wrap_cell(cc[1] + dy, g.nc[1])) * g.nc[0] +
                                     wol_wrap_cell(cc[0] + dx, g.nc[0]);
                    const int b = cs[cell], en = cs[cell + 1];
                    for (int jj = b; jj < en; ++jj) {
                        const double4 q = sp[jj];
                        double dist;
                        bool in_mask;
                        if (wol_eval_pair(c, q.x, q.y, q.z, g.box, a.lowsq, a.highsq, mine.bound2, &dist, &in_mask))
                            top4_insert(mine, dist, (int)__double_as_longlong(q.w), jj);
                    }
                }
                all = mine;
                wol_top4_wave_merge(all);
                double rg = __builtin_huge_val();
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const bool whole = (shell >= g.lo[k]) && (shell >= g.hi[k]) && (g.lo[k] + g.hi[k] + 1 >= g.nc[k]);
                    if (!whole) rg = fmin(rg, (double)shell * g.cw[k] + fmin(fr[k], 1.0 - fr[k]) * g.cw[k]);
                }
                rg -= 1e-6;
                if (rg >= high) break;
                if (all.n == 4 && all.d3 <= rg) break;
            }
        }
        if (lane == 0) {
            const double4 n0 = sp[all.s0], n1 = sp[all.s1], n2 = sp[all.s2], n3 = sp[all.s3];
            const double N0[3] = {n0.x, n0.y, n0.z}, N1[3] = {n1.x, n1.y, n1.z};
            const double N2[3] = {n2.x, n2.y, n2.z}, N3[3] = {n3.x, n3.y, n3.z};
            QOut o;
            o.q = nullptr; o.q32 = nullptr; o.nn_idx = nullptr; o.nn_dist = nullptr; o.ang = nullptr;
            o.qhist = nullptr; o.dhist = nullptr; o.qbins = 0; o.dbins = 0; o.dbinwidth = 0.0; o.stat = nullptr;
            double ang[6];
            double P0[3] = {0, 0, 0}, P1[3] = {0, 0, 0}, P2[3] = {0, 0, 0}, P3[3] = {0, 0, 0};
            const int nb = all.n;
            if (nb >= 1) wol_neighbour_point(c, N0[0], N0[1], N0[2], g.box, P0);
            if (nb >= 2) wol_neighbour_point(c, N1[0], N1[1], N1[2], g.box, P1);
            if (nb >= 3) wol_neighbour_point(c, N2[0], N2[1], N2[2], g.box, P2);
            if (nb >= 4) wol_neighbour_point(c, N3[0], N3[1], N3[2], g.box, P3);
            const double q = wol_q_from_points(c, P0, P1, P2, P3, nb, ang);
            (void)o;
            if (pt_q) pt_q[p] = q;
            const int seg = t * (a.npops + 1) + 1 + j;
            long long h, l, h2, l2;
            wol_fix_split(q, &h, &l);
            wol_fix_split(q * q, &h2, &l2);
            unsigned long long* st = a.stat + (size_t)seg * 8;
            atomicAdd(&st[0], 1ull);
            atomicAdd(&st[1], (unsigned long long)h); atomicAdd(&st[2], (unsigned long long)l);
            atomicAdd(&st[3], (unsigned long long)h2); atomicAdd(&st[4], (unsigned long long)l2);
            const int bin = wol_np_hist_bin(q, qbins, 0.0, 1.0);
            if (bin >= 0) atomicAdd(&a.hist[(size_t)j * qbins + bin], 1ull);
        }
    }
}

// per-segment results from the integer accumulators.  kind 0 (cos-angs): counts[seg] = {angles, tetrahedral angles},
// sums[seg] = {sum cos, sum cos^2}.  kind 1 (LSI, q): counts[seg] = {values}, mean / var (np.mean / np.var; NaN for an
// empty segment, as numpy).  first / step select the segments (step = npops + 1 skips population 0).
__global__ void __launch_bounds__(256)
k_traj_finalize(const unsigned long long* __restrict__ stat, int nseg, int first, int npops1, int kind, int skip_pop0,
                long long* __restrict__ counts, double* __restrict__ out_a, double* __restrict__ out_b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nseg) return;
    const int seg = first + i;
    if (skip_pop0 && seg % npops1 == 0) return;
    const long long* s = reinterpret_cast<const long long*>(stat) + (size_t)seg * 8;
    const double k30 = 9.313225746154785e-10, k70 = 8.470329472543003e-22;   // 2^-30, 2^-70
    if (kind == 0) {
        counts[2 * (size_t)seg] = s[0]; counts[2 * (size_t)seg + 1] = s[1];
        out_a[2 * (size_t)seg] = (double)s[2] * k30 + (double)s[3] * k70;
        out_a[2 * (size_t)seg + 1] = (double)s[4] * k30 + (double)s[5] * k70;
    } else {
        const double cnt = (double)s[0];
        const double s1 = (double)s[1] * k30 + (double)s[2] * k70, s2 = (double)s[3] * k30 + (double)s[4] * k70;
        // out index: population-major series are laid out [T][npops + 1] like the segments, or, for the q populations,
        // [T][npops] (skip_pop0): seg -> t * npops + (j - 1)
        const size_t o = skip_pop0 ? (size_t)(seg / npops1) * (npops1 - 1) + (seg % npops1 - 1) : (size_t)seg;
        if (counts) counts[o] = s[0];
        const double m = s1 / cnt;   // 0 / 0 = NaN for an empty segment, as np.mean
        if (out_a) out_a[o] = m;
        if (out_b) out_b[o] = s2 / cnt - m * m;
    }
}
