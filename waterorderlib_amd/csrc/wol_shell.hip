// wol_shell.hip -- shell-neighbour kernels behind wol_cos_angs (water_properties.py:210-250) and
// wol_lsi_frame (water_properties.py:252-311); included by wol_api.hip.
//
// Both need EVERY neighbour inside a cutoff shell (not just the nearest four), in ascending
// original index (the order of `Pos[nearNeighbs[i]]` in the reference).  One lane per query walks
// all cells the cutoff can reach (grid built with that reach, so no early exit and no list is
// missed) or, for boxes with a non-periodic axis, all positions.  Variable-length results use a
// count pass, a device scan and a fill pass; per-query neighbour lists live in a global scratch
// segment that the lane sorts in place (shells hold ~4-10 waters).
#pragma once

struct ShellArgs {
    const FrameGrid* grid;     // single frame
    const double4* sorted;     // cell-sorted positions (x, y, z, original index)
    const int* cell_start;
    int n;                     // candidates
    const double* sub;         // queries, row stride 3; null = the candidates themselves (sorted order)
    int ns;
    int use_cells;             // 0: all-pairs over `sorted` (any box)
    double lowsq, highsq;
};

// calls f(slot, d2, dx, dy, dz) for every candidate whose exact minimum-image d2 can matter
template <typename F>
__device__ __forceinline__ void wol_for_each_candidate(const ShellArgs& a, const double c[3], F f) {
    const FrameGrid& g = *a.grid;
    const double4* __restrict__ sp = a.sorted;
    if (!a.use_cells) {
        for (int j = 0; j < a.n; ++j) {
            const double4 v = sp[j];
            const double dx = wol_minimg1(v.x - c[0], g.box.L[0], g.box.ib[0]);
            const double dy = wol_minimg1(v.y - c[1], g.box.L[1], g.box.ib[1]);
            const double dz = wol_minimg1(v.z - c[2], g.box.L[2], g.box.ib[2]);
            f(j, wol_sumsq(dx, dy, dz), dx, dy, dz);
        }
        return;
    }
    const int* __restrict__ cs = a.cell_start + g.cell_base;
    int cc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) cc[k] = wol_cell_coord(c[k], g.box.ib[k], g.nc[k], nullptr);
    for (int dz = -g.lo[2]; dz <= g.hi[2]; ++dz) {
        const int cz = wol_wrap_cell(cc[2] + dz, g.nc[2]);
        for (int dy = -g.lo[1]; dy <= g.hi[1]; ++dy) {
            const int cy = wol_wrap_cell(cc[1] + dy, g.nc[1]);
            for (int dx = -g.lo[0]; dx <= g.hi[0]; ++dx) {
                const int cx = wol_wrap_cell(cc[0] + dx, g.nc[0]);
                const int cell = (cz * g.nc[1] + cy) * g.nc[0] + cx;
                const int b = cs[cell], e = cs[cell + 1];
                for (int j = b; j < e; ++j) {
                    const double4 v = sp[j];
                    const double ex = wol_minimg1(v.x - c[0], g.box.L[0], g.box.ib[0]);
                    const double ey = wol_minimg1(v.y - c[1], g.box.L[1], g.box.ib[1]);
                    const double ez = wol_minimg1(v.z - c[2], g.box.L[2], g.box.ib[2]);
                    f(j, wol_sumsq(ex, ey, ez), ex, ey, ez);
                }
            }
        }
    }
}

__device__ __forceinline__ void wol_shell_query(const ShellArgs& a, int q, double c[3], int* out_row) {
    if (a.sub) {
        c[0] = a.sub[3 * (int64_t)q]; c[1] = a.sub[3 * (int64_t)q + 1]; c[2] = a.sub[3 * (int64_t)q + 2];
        *out_row = q;
    } else {
        const double4 v = a.sorted[q];
        c[0] = v.x; c[1] = v.y; c[2] = v.z;
        *out_row = (int)__double_as_longlong(v.w);
    }
}

// ---- pass 1: neighbours in the shell per query (by output row) ------------------------------
__global__ void __launch_bounds__(WOL_BLOCK)
k_shell_count(ShellArgs a, int* __restrict__ count) {
    const int q = blockIdx.x * WOL_BLOCK + threadIdx.x;
    if (q >= a.ns) return;
    double c[3];
    int row;
    wol_shell_query(a, q, c, &row);
    int k = 0;
    wol_for_each_candidate(a, c, [&](int, double d2, double, double, double) {
        k += (d2 > a.lowsq && d2 <= a.highsq) ? 1 : 0;
    });
    count[row] = k;
}

// exclusive scans of k and of k(k-1)/2 over the rows: one block, tiles of 4096 rows (four consecutive per thread),
// wave scans by shuffle, running carries
__global__ void __launch_bounds__(1024)
k_shell_offsets(const int* __restrict__ count, int ns, long long* __restrict__ off_nbr,
                long long* __restrict__ off_ang, long long* __restrict__ totals) {
    __shared__ long long wa[16], wb[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long ca = 0, cb = 0;
    for (int t0 = 0; t0 < ns; t0 += 4096) {
        const int i = t0 + 4 * tid;
        long long k[4], sa = 0, sb = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { k[u] = (i + u < ns) ? count[i + u] : 0; sa += k[u]; sb += k[u] * (k[u] - 1) / 2; }
        long long ia = sa, ib = sb;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long xa = __shfl_up(ia, off, 64), xb = __shfl_up(ib, off, 64);
            if (lane >= off) { ia += xa; ib += xb; }
        }
        if (lane == 63) { wa[wave] = ia; wb[wave] = ib; }
        __syncthreads();
        long long ba = 0, bb = 0, ta = 0, tb = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { ta += wa[w]; tb += wb[w]; if (w < wave) { ba += wa[w]; bb += wb[w]; } }
        long long ra = ca + ba + ia - sa, rb = cb + bb + ib - sb;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u < ns) { off_nbr[i + u] = ra; off_ang[i + u] = rb; }
            ra += k[u]; rb += k[u] * (k[u] - 1) / 2;
        }
        ca += ta; cb += tb;
        __syncthreads();
    }
    if (tid == 0) { totals[0] = ca; totals[1] = cb; }
}

// ---- getCosAngs pass 2: list the shell (ascending original index), all pair angles ----------
__global__ void __launch_bounds__(WOL_BLOCK)
k_cos_angs_fill(ShellArgs a, const long long* __restrict__ off_nbr, const long long* __restrict__ off_ang,
                unsigned long long* __restrict__ nbr, double* __restrict__ ang) {
    const int q = blockIdx.x * WOL_BLOCK + threadIdx.x;
    if (q >= a.ns) return;
    double c[3];
    int row;
    wol_shell_query(a, q, c, &row);
    unsigned long long* __restrict__ lst = nbr + off_nbr[row];
    int k = 0;
    wol_for_each_candidate(a, c, [&](int slot, double d2, double, double, double) {
        if (d2 > a.lowsq && d2 <= a.highsq) {
            const unsigned long long key =
                ((unsigned long long)(unsigned)__double_as_longlong(a.sorted[slot].w) << 32) | (unsigned)slot;
            int p = k++;  // insertion sort by original index
            while (p > 0 && lst[p - 1] > key) { lst[p] = lst[p - 1]; --p; }
            lst[p] = key;
        }
    });
    const FrameGrid& g = *a.grid;
    double* __restrict__ out = ang + off_ang[row];
    long long o = 0;
    for (int u = 0; u < k; ++u) {
        const double4 pu = a.sorted[(int)(lst[u] & 0xFFFFFFFFu)];
        double P1[3];  // tetraCosAng re-images every neighbour about the centre (waterlib.f90:881-883)
        P1[0] = c[0] + wol_minimg1(pu.x - c[0], g.box.L[0], g.box.ib[0]);
        P1[1] = c[1] + wol_minimg1(pu.y - c[1], g.box.L[1], g.box.ib[1]);
        P1[2] = c[2] + wol_minimg1(pu.z - c[2], g.box.L[2], g.box.ib[2]);
        for (int v = u + 1; v < k; ++v) {
            const double4 pv = a.sorted[(int)(lst[v] & 0xFFFFFFFFu)];
            double P2[3];
            P2[0] = c[0] + wol_minimg1(pv.x - c[0], g.box.L[0], g.box.ib[0]);
            P2[1] = c[1] + wol_minimg1(pv.y - c[1], g.box.L[1], g.box.ib[1]);
            P2[2] = c[2] + wol_minimg1(pv.z - c[2], g.box.L[2], g.box.ib[2]);
            out[o++] = wol_cos_angle3(P1, c, P2);
        }
    }
}

// ---- getLSI --------------------------------------------------------------------------------
// numpy's pairwise sum for n < 128 (numpy/_core/src/umath/loops_utils.h.src), needed so that
// np.var of the gap list is reproduced to the bit
__device__ __forceinline__ double wol_np_sum_small(const double* v, int n) {
    if (n < 8) {
        double r = -0.0;
        for (int i = 0; i < n; ++i) r += v[i];
        return r;
    }
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = v[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] += v[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += v[i];
    return res;
}

#define WOL_LSI_CAP 96  // shell + 1 distances per water held in scratch at the first attempt (the 3.7 A shell has ~5-8);
                        // a denser shell reports its size and the host launches again with room for it

__global__ void __launch_bounds__(WOL_BLOCK)
k_lsi(ShellArgs a, double high2sq, double* __restrict__ scratch, int cap, double* __restrict__ lsi_full,
      double* __restrict__ num_lsi, int* __restrict__ overflow) {
    const int q = blockIdx.x * WOL_BLOCK + threadIdx.x;
    if (q >= a.ns) return;
    double c[3];
    int row;
    wol_shell_query(a, q, c, &row);
    double* __restrict__ dist = scratch + (int64_t)q * cap;
    const FrameGrid& g = *a.grid;
    int k = 0;
    bool have_next = false;
    double next_raw = 0.0, next_d2 = 0.0;
    int next_idx = 0x7fffffff;
    wol_for_each_candidate(a, c, [&](int slot, double d2, double, double, double) {
        if (d2 > a.lowsq && d2 <= a.highsq) {
            if (k < cap - 1) dist[k] = sqrt(d2);  // lsiDists, waterlib.f90:916
            ++k;
        }
        if (d2 > a.highsq && d2 <= high2sq) {
            // wp:289-291: the "next" neighbour is picked by its RAW (not minimum-image) distance;
            // np.argmin keeps the first minimum in ascending original index
            const double4 p = a.sorted[slot];
            const double rx = p.x - c[0], ry = p.y - c[1], rz = p.z - c[2];
            const double raw = sqrt(wol_sumsq(rx, ry, rz));
            const int idx = (int)__double_as_longlong(p.w);
            if (!have_next || raw < next_raw || (raw == next_raw && idx < next_idx)) {
                have_next = true; next_raw = raw; next_idx = idx; next_d2 = d2;
            }
        }
    });
    (void)g;
    double res = __builtin_nan("");
    double cnt = 0.0;
    if (k >= cap - 1) {
        atomicMax(overflow, k + 2);   // the capacity this water needs
    } else if (k > 1 && have_next) {
        dist[k] = sqrt(next_d2);
        const int m = k + 1;
        for (int i = 1; i < m; ++i) {  // np.sort
            const double x = dist[i];
            int p = i;
            while (p > 0 && dist[p - 1] > x) { dist[p] = dist[p - 1]; --p; }
            dist[p] = x;
        }
        for (int i = 0; i < m - 1; ++i) dist[i] = dist[i + 1] - dist[i];  // successive gaps
        const int nd = m - 1;
        const double mean = wol_np_sum_small(dist, nd) / (double)nd;  // np.var (ddof = 0)
        for (int i = 0; i < nd; ++i) { const double x = dist[i] - mean; dist[i] = x * x; }
        res = wol_np_sum_small(dist, nd) / (double)nd;
        cnt = (double)nd;
    }
    lsi_full[row] = res;
    num_lsi[row] = cnt;
}

// ---------------------------------------------------------------------------------------------
// Willard-Chandler density field (waterlib.f90:1286-1341 WillardDensityField,
// :1351-1398 WillardDensityPoints; surface_library.densityGrid :170-210 is its caller).
// One lane per grid / query point gathers the waters within 3*smoothlen through the cell list
// (the reference loops over all N waters per point: O(G*N)); truncated-shifted Gaussian density and
// its normalised gradient in the reference's operation order.  The cells are first put in
// original-index order (k_cell_order) so the floating-point sums are reproducible run to run.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WOL_BLOCK)
k_cell_order(const FrameGrid* __restrict__ grid, const int* __restrict__ cell_start, double4* __restrict__ sorted) {
    const int c = blockIdx.x * WOL_BLOCK + threadIdx.x;
    if (c >= grid->ncell) return;
    const int b = cell_start[grid->cell_base + c], e = cell_start[grid->cell_base + c + 1];
    for (int i = b + 1; i < e; ++i) {  // insertion sort by original index; cells hold a handful of waters
        const double4 v = sorted[i];
        const long long key = __double_as_longlong(v.w);
        int p = i;
        while (p > b && __double_as_longlong(sorted[p - 1].w) > key) { sorted[p] = sorted[p - 1]; --p; }
        sorted[p] = v;
    }
}

struct WillardArgs {
    const double* gx; const double* gy; const double* gz;  // grid mode (device); null = points mode (a.sub)
    int nx, ny, nz;
    double smoothlen;
    double* densvals;   // grid: Fortran order (nx,ny,nz); points: [npts]
    double* densnorms;  // grid: Fortran order (nx,ny,nz,3); points: row-major [npts][3]
};

__global__ void __launch_bounds__(WOL_BLOCK)
k_willard(ShellArgs a, WillardArgs w) {
    const int64_t id = (int64_t)blockIdx.x * WOL_BLOCK + threadIdx.x;
    const int64_t npts = w.gx ? (int64_t)w.nx * w.ny * w.nz : (int64_t)a.ns;
    if (id >= npts) return;
    double c[3];
    if (w.gx) {  // id runs over (i, j, k) in Fortran order: i fastest
        const int i = (int)(id % w.nx), j = (int)((id / w.nx) % w.ny), k = (int)(id / ((int64_t)w.nx * w.ny));
        c[0] = w.gx[i]; c[1] = w.gy[j]; c[2] = w.gz[k];
    } else {
        c[0] = a.sub[3 * id]; c[1] = a.sub[3 * id + 1]; c[2] = a.sub[3 * id + 2];
    }
    const double pi = WOL_PI;
    const double s2 = w.smoothlen * w.smoothlen;
    const double pref = pow(2.0 * pi * s2, 1.5);
    const double shiftterm = exp(-9.0 / 2.0) / pref;
    const double cut = 9.0 * (w.smoothlen * w.smoothlen);
    double densval = 0.0, n0 = 0.0, n1 = 0.0, n2 = 0.0;
    // the reference forms thisvec = apos - watpos; the shared walker hands back watpos - apos after the
    // same minimum-image step, and negation is exact (anint is odd), so v = -d bit for bit
    wol_for_each_candidate(a, c, [&](int, double r2, double dx, double dy, double dz) {
        if (r2 < cut) {
            const double expterm = -r2 / (2.0 * s2);
            const double densfunc = exp(expterm) / pref - shiftterm;
            densval += densfunc;
            const double g = densfunc + shiftterm;  // -thisvec * (densfunc + shiftterm) / s2, left to right
            n0 += (dx * g) / s2; n1 += (dy * g) / s2; n2 += (dz * g) / s2;
        }
    });
    const double nn = sqrt((n0 * n0 + n1 * n1) + n2 * n2);
    w.densvals[id] = densval;
    if (w.gx) {
        w.densnorms[id] = n0 / nn; w.densnorms[npts + id] = n1 / nn; w.densnorms[2 * npts + id] = n2 / nn;
    } else {
        w.densnorms[3 * id] = n0 / nn; w.densnorms[3 * id + 1] = n1 / nn; w.densnorms[3 * id + 2] = n2 / nn;
    }
}

// ---------------------------------------------------------------------------------------------
// pair-distance histograms behind RadialDistSame / RadialDist / PairDistanceHistogram
// (waterlib.f90:316-357, 193-231, 363-389; caller orderParam_lib.rdfCalc :575-727).  One lane per query
// walks the cells within totbins*binwidth (the reference loops over all pairs); dist = sqrt((x^2+y^2)+z^2)
// of the minimum-image vector and nbin = ceiling(dist/binwidth) as the reference; integer counts in an
// LDS histogram, flushed with 64-bit atomics.  `same`: the queries are the candidates themselves and each
// unordered pair is counted once (slot order instead of the reference's index order: the same pair set).
// Zero distances are skipped (PairDistanceHistogram does; in the other two the reference writes counts(0)).
// ---------------------------------------------------------------------------------------------
// nbin = ceiling(sqrt(d2) / binwidth) (waterlib.f90:219-222) of an exact squared distance: guessed in fp32 and accepted
// when d2 lies strictly inside the squared edges of that bin with a 1e-12 margin (which covers the roundings of the
// reference's sqrt and division, so its ceiling is this bin); on or near an edge the reference's own expression decides
__device__ __forceinline__ int wol_pair_bin_exact(double d2, double binwidth, float inv_wf) {
    int k = (int)ceilf(__builtin_sqrtf((float)d2) * inv_wf);
    const double lo = (double)(k - 1) * binwidth, hi = (double)k * binwidth;
    if (!(d2 > lo * lo * (1.0 + 1e-12) && d2 < hi * hi * (1.0 - 1e-12))) k = (int)ceil(sqrt(d2) / binwidth);
    return k;
}

// ---------------------------------------------------------------------------------------------
// The same counts for ONE set (RadialDistSame: pairs once), LDS-tiled.  A 128-thread block owns the waters of a run of
// cells along x (one lane each) and sweeps the cell rows in reach, staged into LDS as float32 coordinates relative to the
// run's centre (the image nearest the centre, taken in fp64; 16 bytes per candidate with its slot): every lane looks at
// the same candidate in the same iteration -- a broadcast LDS read and uniform trip counts, where the lane-per-water
// walk pays the longest cell of 64 different walks in every wave and a chain of dependent L2 round trips.
// The fp32 squared distance only PROPOSES the bin: it is taken when it lies inside the bin's squared edges by more than a
// bound on its own error (coordinates rounded at <= 64 A: 4e-6 A each; three products), else -- about one pair in a
// thousand -- the exact minimum-image distance of the fp64 records decides, as in k_pair_hist.  So the integer counts are
// those of the exact kernel.
// ---------------------------------------------------------------------------------------------
#define WOL_PT_BLOCK 64
#define WOL_PT_CAP 512       // staged candidates per sweep (8 KiB): one row of the walk is ~150
struct PairTileArgs {
    const FrameGrid* grid;
    const double4* sorted;
    const int* cell_start;
    int n;
    int cxrun, ntx;          // cells per run along x, runs per cell row
    int need_minimg;         // the box is too small for "nearest image to the run's centre" to be every lane's nearest image
    double binwidth;
    int totbins;
    float cut2f;             // (totbins * binwidth)^2 + margin: beyond it a pair cannot fall into a bin
    unsigned long long* hist;
};

template <bool MINIMG>
__global__ void __launch_bounds__(WOL_PT_BLOCK)
k_pair_hist_tiled(PairTileArgs a) {
    extern __shared__ int lds_hist[];                 // [totbins]
    __shared__ float4 cand[WOL_PT_CAP];               // (x, y, z relative to the centre, slot)
    const FrameGrid& g = *a.grid;
    const int tid = threadIdx.x;
    const int nx = g.nc[0], ny = g.nc[1], nz = g.nc[2];
    wol_hist_zero(lds_hist, a.totbins);
    const double4* __restrict__ sp = a.sorted;
    const int* __restrict__ cs = a.cell_start + g.cell_base;
    const float inv_wf = (float)(1.0 / a.binwidth), wf = (float)a.binwidth;
    const float Lf[3] = {(float)g.box.L[0], (float)g.box.L[1], (float)g.box.L[2]};
    const float ibf[3] = {(float)g.box.ib[0], (float)g.box.ib[1], (float)g.box.ib[2]};
    // work unit = (run of cells, one (dz, dy) row of the walk): units are dealt round-robin to a grid of resident blocks, so
    // the launch ends evenly whatever the number of runs (whole tiles per block left the last round of blocks half empty)
    const int wy = g.lo[1] + g.hi[1] + 1, wz = g.lo[2] + g.hi[2] + 1, nwalk = wy * wz;
    const int64_t nunits = (int64_t)nz * ny * a.ntx * nwalk;
    for (int64_t u = blockIdx.x; u < nunits; u += gridDim.x) {
        const int tile = (int)(u / nwalk), wr = (int)(u - (int64_t)tile * nwalk);
        const int row = tile / a.ntx, tx = tile - row * a.ntx;
        const int cz = row / ny, cy = row - cz * ny;
        const int cx0 = tx * a.cxrun, cxe = min(a.cxrun, nx - cx0);
        if (cxe <= 0) continue;
        const int dz = wr / wy - g.lo[2], dy = wr - (wr / wy) * wy - g.lo[1];
        const int qb = cs[row * nx + cx0], qe = cs[row * nx + cx0 + cxe];
        if (qe <= qb) continue;
        const double ctr[3] = {((double)cx0 + 0.5 * (double)cxe) * g.cw[0], ((double)cy + 0.5) * g.cw[1], ((double)cz + 0.5) * g.cw[2]};
        // candidate cells along x: the union of the lanes' walks, [cx0 - lo, cx0 + cxe - 1 + hi], or the whole row
        const int xa = cx0 - g.lo[0], xb = cx0 + cxe - 1 + g.hi[0];
        const bool whole = xb - xa + 1 >= nx;
        // every unordered pair is counted once, from the water with the lower slot (slots run row by row): a candidate row
        // below the lanes' own holds no partner that counts here -- half of the walk's rows are skipped whole
        const int crow = wol_wrap_cell(cz + dz, nz) * ny + wol_wrap_cell(cy + dy, ny);
        if (crow < row) continue;
        const int* __restrict__ rw = cs + crow * nx;
        int p0a, p0b, p1a = 0, p1b = 0;
        if (whole) { p0a = 0; p0b = nx; }
        else if (xa < 0) { p0a = xa + nx; p0b = nx; p1a = 0; p1b = xb + 1; }
        else if (xb >= nx) { p0a = xa; p0b = nx; p1a = 0; p1b = xb - nx + 1; }
        else { p0a = xa; p0b = xb + 1; }
        for (int q0 = qb; q0 < qe; q0 += WOL_PT_BLOCK) {
            const int qs = q0 + tid;
            const bool live = qs < qe;
            double c[3] = {0.0, 0.0, 0.0};
            float qx = 0.f, qy = 0.f, qz = 0.f;
            if (live) {
                const double4 v = sp[qs];
                c[0] = v.x; c[1] = v.y; c[2] = v.z;
                qx = (float)wol_minimg1(v.x - ctr[0], g.box.L[0], g.box.ib[0]);
                qy = (float)wol_minimg1(v.y - ctr[1], g.box.L[1], g.box.ib[1]);
                qz = (float)wol_minimg1(v.z - ctr[2], g.box.L[2], g.box.ib[2]);
            }
            int fill = 0;                                       // candidates staged and not yet swept (block-uniform)
            auto sweep = [&]() {
                __syncthreads();
                if (live) {
                    for (int i = 0; i < fill; ++i) {
                        const float4 p = cand[i];
                        const int slot = __float_as_int(p.w);
                        float dx = p.x - qx, dy_ = p.y - qy, dz_ = p.z - qz;
                        if (MINIMG) {
                            dx -= Lf[0] * rintf(dx * ibf[0]); dy_ -= Lf[1] * rintf(dy_ * ibf[1]); dz_ -= Lf[2] * rintf(dz_ * ibf[2]);
                        }
                        const float d2f = __builtin_fmaf(dz_, dz_, __builtin_fmaf(dy_, dy_, dx * dx));
                        if (d2f >= a.cut2f || slot <= qs) continue;     // out of range, or the pair belongs to the other water
                        const float kf = ceilf(__builtin_sqrtf(d2f) * inv_wf);
                        const float lo = (kf - 1.0f) * wf, hi = kf * wf;
                        const float m = 4.0e-4f + 2.0e-6f * d2f;        // bound on |d2f - d2| (+ the edges' own rounding)
                        int k = (int)kf;
                        if (!(d2f > __builtin_fmaf(lo, lo, m) && d2f < __builtin_fmaf(hi, hi, -m))) {
                            const double4 w = sp[slot];
                            const double ex = wol_minimg1(w.x - c[0], g.box.L[0], g.box.ib[0]);
                            const double ey = wol_minimg1(w.y - c[1], g.box.L[1], g.box.ib[1]);
                            const double ez = wol_minimg1(w.z - c[2], g.box.L[2], g.box.ib[2]);
                            const double d2 = wol_sumsq(ex, ey, ez);
                            if (!(d2 > 0.0)) continue;
                            k = wol_pair_bin_exact(d2, a.binwidth, inv_wf);
                        }
                        if (k >= 1 && k <= a.totbins) atomicAdd(&lds_hist[k - 1], 1);
                    }
                }
                __syncthreads();
                fill = 0;
            };
#pragma unroll
            for (int piece = 0; piece < 2; ++piece) {
                const int ca = piece ? p1a : p0a, cb = piece ? p1b : p0b;
                if (cb <= ca) continue;
                int b = rw[ca];
                const int e = rw[cb];
                while (b < e) {
                    const int take = min(e - b, WOL_PT_CAP - fill);
                    for (int j = tid; j < take; j += WOL_PT_BLOCK) {
                        const double4 v = sp[b + j];
                        float4 r;
                        r.x = (float)wol_minimg1(v.x - ctr[0], g.box.L[0], g.box.ib[0]);
                        r.y = (float)wol_minimg1(v.y - ctr[1], g.box.L[1], g.box.ib[1]);
                        r.z = (float)wol_minimg1(v.z - ctr[2], g.box.L[2], g.box.ib[2]);
                        r.w = __int_as_float(b + j);
                        cand[fill + j] = r;
                    }
                    fill += take;
                    b += take;
                    if (fill == WOL_PT_CAP) sweep();
                }
            }
            if (fill > 0) sweep();
        }
    }
    __syncthreads();
    wol_hist_flush(lds_hist, a.totbins, a.hist);
}

__global__ void __launch_bounds__(WOL_BLOCK)
k_pair_hist(ShellArgs a, int same, double binwidth, int totbins, unsigned long long* __restrict__ hist) {
    extern __shared__ int lds_hist[];
    wol_hist_zero(lds_hist, totbins);
    __syncthreads();
    const float inv_wf = (float)(1.0 / binwidth);
    for (int q = blockIdx.x * WOL_BLOCK + threadIdx.x; q < a.ns; q += gridDim.x * WOL_BLOCK) {
        double c[3];
        int row;
        wol_shell_query(a, q, c, &row);
        wol_for_each_candidate(a, c, [&](int slot, double d2, double, double, double) {
            if (same && slot <= q) return;
            if (!(d2 > 0.0)) return;
            const int k = wol_pair_bin_exact(d2, binwidth, inv_wf);
            if (k >= 1 && k <= totbins) atomicAdd(&lds_hist[k - 1], 1);
        });
    }
    __syncthreads();
    wol_hist_flush(lds_hist, totbins, hist);
}

// ---------------------------------------------------------------------------------------------
// hydrogen bonds (waterlib.f90:1156-1210 generalHbonds; caller orderParam_lib.getBoundWrap
// :419-572).  The pair test in the reference's arithmetic; two drivers: a dense (nacc, ndon) Fortran-order int32
// mask for f2py compatibility (one thread per element, like the reference's O(nacc*ndon) loop) and a sparse
// list through the cell list on the donors (reach distCut), for frames where the mask would not fit.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool wol_hbond_pair(const double a[3], const double d[3], const double* __restrict__ h,
                                               const Box3& b, double distcutsq, const AngCut& angcut) {
    double av[3], dv[3];
    const double v0 = wol_minimg1(d[0] - a[0], b.L[0], b.ib[0]), v1 = wol_minimg1(d[1] - a[1], b.L[1], b.ib[1]);
    const double v2 = wol_minimg1(d[2] - a[2], b.L[2], b.ib[2]);
    const double dsq = wol_sumsq(v0, v1, v2);
    if (dsq > distcutsq || dsq <= (double)1.0E-2f) return false;   // single-precision literal, as the Fortran
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        av[k] = wol_minimg1(a[k] - h[k], b.L[k], b.ib[k]);
        dv[k] = wol_minimg1(d[k] - h[k], b.L[k], b.ib[k]);
    }
    const double na = sqrt(wol_sumsq(av[0], av[1], av[2])), nd = sqrt(wol_sumsq(dv[0], dv[1], dv[2]));
#pragma unroll
    for (int k = 0; k < 3; ++k) { av[k] = av[k] / na; dv[k] = dv[k] / nd; }
    return !wol_ang_below_cut(av, dv, angcut);   // waterlib.f90:1200: `if (ang < angCut) cycle`
}

__global__ void __launch_bounds__(WOL_BLOCK)
k_hbonds_dense(const double* __restrict__ acc, int nacc, const double* __restrict__ don, const double* __restrict__ donh,
               int ndon, Box3 b, double distcutsq, AngCut angcut, int32_t* __restrict__ bond) {
    const int64_t e = (int64_t)blockIdx.x * WOL_BLOCK + threadIdx.x;   // element of the (nacc, ndon) Fortran-order mask
    if (e >= (int64_t)nacc * ndon) return;
    const int j = (int)(e / nacc), i = (int)(e - (int64_t)j * nacc);
    const double a[3] = {acc[3 * (int64_t)i], acc[3 * (int64_t)i + 1], acc[3 * (int64_t)i + 2]};
    const double d[3] = {don[3 * (int64_t)j], don[3 * (int64_t)j + 1], don[3 * (int64_t)j + 2]};
    bond[e] = wol_hbond_pair(a, d, donh + 3 * (int64_t)j, b, distcutsq, angcut) ? 1 : 0;
}

// sparse form: a.sorted = cell-sorted donors, a.sub = acceptors; pairs appended as (acceptor << 32 | donor)
__global__ void __launch_bounds__(WOL_BLOCK)
k_hbonds_list(ShellArgs a, const double* __restrict__ donh, double distcutsq, AngCut angcut,
              unsigned long long* __restrict__ pairs, unsigned long long capacity, unsigned long long* __restrict__ count) {
    const int q = blockIdx.x * WOL_BLOCK + threadIdx.x;
    if (q >= a.ns) return;
    double c[3];
    int row;
    wol_shell_query(a, q, c, &row);
    const Box3& b = a.grid->box;
    wol_for_each_candidate(a, c, [&](int slot, double d2, double, double, double) {
        if (d2 > distcutsq || d2 <= (double)1.0E-2f) return;   // the test inside wol_hbond_pair again: cheap pre-filter
        const double4 dn = a.sorted[slot];
        const int j = (int)__double_as_longlong(dn.w);
        const double d[3] = {dn.x, dn.y, dn.z};
        if (wol_hbond_pair(c, d, donh + 3 * (int64_t)j, b, distcutsq, angcut)) {
            const unsigned long long pos = atomicAdd(count, 1ull);
            if (pos < capacity) pairs[pos] = ((unsigned long long)(unsigned)row << 32) | (unsigned)j;
        }
    });
}
