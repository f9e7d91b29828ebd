// wol_device.h -- device-side arithmetic shared by the kernels of libwol.so (gfx950).
//
// Everything parity-critical is written in the exact operation order of the reference
// (fortran/waterlib.f90, structureLibs/water_properties.py) and the translation unit is built
// with -ffp-contract=off so no multiply-add is fused behind our back; where an FMA is wanted
// (the fp32 pre-filter) it is spelled out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WOL_PI 3.1415926535897931
#define WOL_NLEV 12

struct Box3 {
    double L[3];   // BoxL
    double ib[3];  // merge(1/BoxL, 0, BoxL >= 0)   (waterlib.f90:41)
};

// per-frame uniform grid over the wrapped coordinates (built on the host, read by the kernels)
struct FrameGrid {
    Box3 box;
    double cw[3];      // cell width per axis = L/nc
    int nc[3];         // cells per axis
    int lo[3], hi[3];  // furthest cell offsets ever needed per axis (dedupes small periodic grids)
    int ncell;
    int maxshell;       // < 0: frame is handled by the all-pairs kernel
    int64_t cell_base;  // offset of this frame's cell_start[] in the batch array
    // ---- tiling of the filtered (fp32 + exact finish) search kernel ----
    int fast_ok;        // grid has >= 3 cells per axis and the cutoff covers the guaranteed radius
    // tile = sx cells along x by ry rows along y by sz layers along z.  The shape is chosen on the
    // device, after the sort, from a short list the host prepared for increasing local densities
    // (waters per cell): level l fits tiles as long as the densest cell row holds <= lev_wpc[l] per cell.
    int nlev;
    float lev_wpc[WOL_NLEV];
    short lev_sx[WOL_NLEV], lev_ry[WOL_NLEV], lev_sz[WOL_NLEV];
    short lev_ntx[WOL_NLEV], lev_nty[WOL_NLEV], lev_ntz[WOL_NLEV];
    int lev_ntiles[WOL_NLEV];
    int ntiles;         // max over the levels (launch bound)
    // tiles of the row-merged proposal kernel (wol_search_rows.hip): sx cells by ry rows in one z layer
    short row_sx[WOL_NLEV], row_ry[WOL_NLEV], row_ntx[WOL_NLEV], row_nty[WOL_NLEV];
    int row_ntiles[WOL_NLEV];
    int row_tiles_max;
    int row_block;      // threads of the k_search_rows instantiation these tiles were sized for (512 or 768)
    int row_eps_units;  // its bound on |fp32 key - exact d^2 * S| (longer tiles: larger coordinates)
    float row_eps_a, row_eps_b;  // the same bound at distance d: a * sqrt(d^2 in key units) + b
    float scale;        // sqrt(S): staged coordinates are multiplied by it so d^2 comes out in key units
    float tmax;         // 2^23 + 2^21 - 1: saturation of the biased squared distance
    double S;           // key units per A^2 (2^21 / r2max)
    int eps_units;      // bound on |fp32 key - exact d^2 * S|
};

// inclusive prefix sum over the 64 lanes of a wave in six DPP additions (shifts inside the rows of 16, then the row totals
// broadcast to the rows behind) where a __shfl_up step is a ds_bpermute, i.e. one more instruction for the LDS unit the
// sweeps of the other blocks are reading their candidates through
__device__ __forceinline__ int wol_wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// part[tid] <- inclusive prefix of s over the threads of the block (whole waves): the waves scan in DPP, their totals are
// combined through the first entries of part[] -- four barriers where a Hillis-Steele scan in LDS takes two per doubling
__device__ __forceinline__ void wol_block_scan_incl(int s, int* part) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int inc = wol_wave_scan_incl(s);
    __syncthreads();
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += part[w];
    __syncthreads();
    part[tid] = before + inc;
    __syncthreads();
}

// A water in the cell-sorted arrays: (x, y, z, original index).  Two storage forms: double4 (index as the bits of .w) and,
// for frames stored as float32, float4 -- single-precision values are carried as they are (16 bytes instead of 32 through
// both sort levels and in the certification kernel's gathers) and promoted, exactly, where the fp64 arithmetic starts.
template <class REC> struct wol_rec;
template <> struct wol_rec<double4> {
    static __device__ __forceinline__ double4 make(double x, double y, double z, int orig) {
        double4 v;
        v.x = x; v.y = y; v.z = z; v.w = __longlong_as_double((long long)orig);
        return v;
    }
    static __device__ __forceinline__ double4 wide(const double4& r) { return r; }
};
template <> struct wol_rec<float4> {
    static __device__ __forceinline__ float4 make(double x, double y, double z, int orig) {   // (x, y, z came from floats)
        float4 v;
        v.x = (float)x; v.y = (float)y; v.z = (float)z; v.w = __int_as_float(orig);
        return v;
    }
    static __device__ __forceinline__ double4 wide(const float4& r) {
        double4 v;
        v.x = (double)r.x; v.y = (double)r.y; v.z = (double)r.z; v.w = __longlong_as_double((long long)__float_as_int(r.w));
        return v;
    }
};
template <class REC> __device__ __forceinline__ double4 wol_rec_load(const REC* __restrict__ p, int64_t i) { return wol_rec<REC>::wide(p[i]); }

__device__ __forceinline__ double wol_minimg1(double d, double L, double ib) {
    // distvec - BoxL*anint(distvec*iBoxL)   (waterlib.f90:44); round() is half-away like anint
    return d - L * round(d * ib);
}

__device__ __forceinline__ double wol_sumsq(double x, double y, double z) { return (x * x + y * y) + z * z; }

// cell coordinate of x along an axis with nc cells; frac in [0,1) is the position inside the cell
__device__ __forceinline__ int wol_cell_coord(double x, double ib, int nc, double* frac) {
    double u = x * ib;
    u = u - floor(u);
    double cu = u * (double)nc;
    int c = (int)cu;
    if (c >= nc) c = nc - 1;
    if (c < 0) c = 0;
    if (frac) {
        double f = cu - (double)c;
        *frac = f < 0.0 ? 0.0 : (f > 1.0 ? 1.0 : f);
    }
    return c;
}

// ---- sorted top-4 of (dist, idx), ascending, ties to the lower ORIGINAL index; s* carry the
// position of the neighbour in whatever array the kernel will fetch its coordinates from --------
struct Top4 {
    double d0, d1, d2, d3;
    int j0, j1, j2, j3;
    int s0, s1, s2, s3;
    int n;          // entries held (<= 4)
    double bound2;  // squared-key bound above which a candidate cannot enter (inf until full)
};

__device__ __forceinline__ void top4_init(Top4& t) {
    t.d0 = t.d1 = t.d2 = t.d3 = __builtin_huge_val();
    t.j0 = t.j1 = t.j2 = t.j3 = 0x7fffffff;
    t.s0 = t.s1 = t.s2 = t.s3 = 0;
    t.n = 0;
    t.bound2 = __builtin_huge_val();
}

__device__ __forceinline__ bool wol_less(double da, int ja, double db, int jb) {
    return (da < db) || (da == db && ja < jb);
}

__device__ __forceinline__ void top4_insert(Top4& t, double d, int j, int s) {
    if (!wol_less(d, j, t.d3, t.j3)) return;
    if (wol_less(d, j, t.d2, t.j2)) {
        t.d3 = t.d2; t.j3 = t.j2; t.s3 = t.s2;
        if (wol_less(d, j, t.d1, t.j1)) {
            t.d2 = t.d1; t.j2 = t.j1; t.s2 = t.s1;
            if (wol_less(d, j, t.d0, t.j0)) {
                t.d1 = t.d0; t.j1 = t.j0; t.s1 = t.s0;
                t.d0 = d; t.j0 = j; t.s0 = s;
            } else { t.d1 = d; t.j1 = j; t.s1 = s; }
        } else { t.d2 = d; t.j2 = j; t.s2 = s; }
    } else { t.d3 = d; t.j3 = j; t.s3 = s; }
    if (t.n < 4) ++t.n;
    // candidates whose squared key exceeds the 4th best by more than a few ulps can neither beat
    // it nor tie with it after the square root
    if (t.n == 4) t.bound2 = (t.d3 * t.d3) * 1.000000000000002;
}

// One (centre a, candidate p) pair evaluated as the reference does (waterlib.f90:733-737 for the
// mask, water_properties.py:372-373 for the ranking key).  Returns true if the pair is in the mask
// and could enter the top-4; *dist receives the key.
__device__ __forceinline__ bool wol_eval_pair(const double a[3], double px, double py, double pz,
                                              const Box3& b, double lowsq, double highsq,
                                              double bound2, double* dist, bool* in_mask) {
    double dx = wol_minimg1(px - a[0], b.L[0], b.ib[0]);
    double dy = wol_minimg1(py - a[1], b.L[1], b.ib[1]);
    double dz = wol_minimg1(pz - a[2], b.L[2], b.ib[2]);
    double d2 = wol_sumsq(dx, dy, dz);
    bool m = (d2 > lowsq) && (d2 <= highsq);
    *in_mask = m;
    if (!m) return false;
    double ex = (a[0] + dx) - a[0];  // reimage (waterlib.f90:45) then thisPos - apos (wp:373)
    double ey = (a[1] + dy) - a[1];
    double ez = (a[2] + dz) - a[2];
    double k2 = wol_sumsq(ex, ey, ez);
    if (k2 > bound2) return false;
    *dist = sqrt(k2);
    return true;
}

// waterlib.f90:683-703, degrees
__device__ __forceinline__ double wol_cos_angle3(const double p1[3], const double p2[3], const double p3[3]) {
    const double pi = WOL_PI;
    const double DegPerRad = 180.0 / pi;
    if ((p1[0] == p2[0] && p1[1] == p2[1] && p1[2] == p2[2]) ||
        (p2[0] == p3[0] && p2[1] == p3[1] && p2[2] == p3[2]))
        return 0.0;
    double v21[3], v23[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { v21[k] = p1[k] - p2[k]; v23[k] = p3[k] - p2[k]; }
    double nrm = sqrt(wol_sumsq(v21[0], v21[1], v21[2]) * wol_sumsq(v23[0], v23[1], v23[2]));
    double dot = (v21[0] * v23[0] + v21[1] * v23[1]) + v21[2] * v23[2];
    double phi = dot / nrm;
    phi = fmin(1.0, fmax(-1.0, phi));
    phi = acos(phi);
    double r = fmod(phi + pi, pi * 2.0) - pi;
    if (r < -pi) r = r + pi * 2.0;
    return r * DegPerRad;
}

// neighbour position as tetraCosAng sees it: the reimaged neighbour (wp:372) is re-imaged about
// the centre once more (waterlib.f90:881-883) -- a no-op except at half-box distances
__device__ __forceinline__ void wol_neighbour_point(const double a[3], double px, double py, double pz,
                                                    const Box3& b, double out[3]) {
    const double p[3] = {px, py, pz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double d = wol_minimg1(p[k] - a[k], b.L[k], b.ib[k]);
        double t = a[k] + d;                               // thisPos (reimage)
        double d1 = wol_minimg1(t - a[k], b.L[k], b.ib[k]);  // tetraCosAng distvec1
        out[k] = a[k] + d1;
    }
}

// water_properties.py:376-388 from the (<=4) selected neighbour points; writes the six angles in
// the order of np.triu_indices (01,02,03,12,13,23 for four neighbours), 180-padded as wp:379-384.
// Fully unrolled (no runtime-indexed arrays -> no scratch).
__device__ __forceinline__ double wol_q_from_points(const double a[3], const double P0[3],
                                                    const double P1[3], const double P2[3],
                                                    const double P3[3], int nb, double ang[6]) {
    if (nb == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) ang[k] = __builtin_nan("");
        return 0.0;  // wp:358
    }
    double a01 = 180.0, a02 = 180.0, a03 = 180.0, a12 = 180.0, a13 = 180.0, a23 = 180.0;
    if (nb >= 2) a01 = wol_cos_angle3(P0, a, P1);
    if (nb >= 3) { a02 = wol_cos_angle3(P0, a, P2); a12 = wol_cos_angle3(P1, a, P2); }
    if (nb >= 4) { a03 = wol_cos_angle3(P0, a, P3); a13 = wol_cos_angle3(P1, a, P3); a23 = wol_cos_angle3(P2, a, P3); }
    if (nb >= 4) { ang[0] = a01; ang[1] = a02; ang[2] = a03; ang[3] = a12; ang[4] = a13; ang[5] = a23; }
    else if (nb == 3) { ang[0] = a01; ang[1] = a02; ang[2] = a12; ang[3] = 180.0; ang[4] = 180.0; ang[5] = 180.0; }
    else { ang[0] = a01; ang[1] = 180.0; ang[2] = 180.0; ang[3] = 180.0; ang[4] = 180.0; ang[5] = 180.0; }
    double s = -0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double c = cos(ang[k] * WOL_PI / 180.0) + (1.0 / 3.0);
        s += c * c;
    }
    return 1.0 - (3.0 / 8.0) * s;  // wp:388
}

// np.histogram(x, bins=nbins, range=[lo,hi]) bin of x, or -1 if x is outside the range
// (numpy/lib/_histograms_impl.py uniform-bin path; edges are linspace(lo,hi,nbins+1))
__device__ __forceinline__ int wol_np_hist_bin(double x, int nbins, double lo, double hi) {
    if (!(x >= lo) || !(x <= hi)) return -1;
    const double step = (hi - lo) / (double)nbins;
    double f = ((x - lo) / (hi - lo)) * (double)nbins;
    int idx = (int)f;
    if (idx == nbins) idx -= 1;
    double e_lo = (double)idx * step + lo;
    if (x < e_lo) idx -= 1;
    double e_hi = (idx + 1 == nbins) ? hi : ((double)(idx + 1) * step + lo);
    if (x >= e_hi && idx != nbins - 1) idx += 1;
    return idx;
}

// pair-distance histogram rule of waterlib.f90:219-222: nbin = ceiling(d/binwidth), 1-based
__device__ __forceinline__ int wol_dist_bin(double d, double binwidth, int nbins) {
    if (!(d == d)) return -1;
    double c = ceil(d / binwidth);
    if (c >= 1.0 && c <= (double)nbins) return (int)c - 1;
    return -1;
}
