/*
 * wol_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU restatement of the reference hot path).
 *
 * Plain-C restatement of WaterOrderLib's per-frame water order-parameter path:
 *   fortran/waterlib.f90  : reimage (:32-47), CosAngle3 (:683-703), nearNeighbors (:710-743),
 *                           allNearNeighbors (:830-862), tetraCosAng (:867-895), lsiDists (:900-918)
 *   structureLibs/water_properties.py : getCosAngs (:210-250), getLSI (:252-311),
 *                           tetrahedralMetrics (:314-342, histogram part), getOrderParamq (:344-391)
 *   structureLibs/orderParam_lib.py   : tetOrderCalc per-frame statistics (:1471-1473) and the
 *                           np.histogram(q, 500, [0,1]) of :1493-1495
 * plus the numpy pieces those Python lines lean on (linalg.norm, argsort, sum/mean/var pairwise
 * summation, histogram's uniform-bin index rule, linspace edges), restated from numpy 2.2.
 *
 * It is the CHECKER for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (waterorderlib_amd/) never imports, links or
 * falls back to it.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks every function here against
 * fixtures under tests/golden/ that were produced by the reference itself (its Fortran compiled
 * with flang via oracle/build_ref.sh, its water_properties.py imported from where it lies; see
 * tests/golden/make_golden.py), and tests/test_oracle_vs_reference.py re-checks live when the
 * reference is present.
 *
 * Arithmetic notes (all fp64, no FMA: build with -ffp-contract=off):
 *   iBoxL = BoxL >= 0 ? 1/BoxL : 0                      (waterlib.f90:41)
 *   d     = d - BoxL * anint(d * iBoxL)                 (waterlib.f90:44; anint = round half away)
 *   d2    = (dx*dx + dy*dy) + dz*dz                     (sum(distvec**2))
 *   mask  = lowCut^2 < d2 <= highCut^2                  (waterlib.f90:737)
 * Neighbour ranking follows water_properties.py:372-374: key = ||(apos + dmin) - apos||, ascending;
 * numpy's argsort is unstable on exact ties, the restatement (and the HIP path) break ties by the
 * lower global index.
 *
 * All position arrays here are C-order [n][3] (what the Python callers hold); masks are written
 * in the layout f2py returns them (Fortran order, int32, element (i,j) at i + j*ns).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define WOLO_PI 3.1415926535897931

/* thread count of the OpenMP cell-list form (bench.py's cpu_baseline states it) */
int wolo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

static inline void ibox_of(const double box[3], double ib[3]) {
    for (int k = 0; k < 3; ++k) ib[k] = (box[k] >= 0.0) ? 1.0 / box[k] : 0.0;
}

/* distvec = p - ref; distvec -= BoxL*anint(distvec*iBoxL)   (waterlib.f90:43-44) */
static inline void minimg(const double p[3], const double ref[3], const double box[3],
                          const double ib[3], double d[3]) {
    for (int k = 0; k < 3; ++k) {
        double t = p[k] - ref[k];
        t = t - box[k] * round(t * ib[k]);
        d[k] = t;
    }
}

static inline double sumsq3(const double d[3]) { return (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]; }

/* ---- waterlib.f90:32-47 ---- */
void wolo_reimage(const double* pos, int n, const double ref[3], const double box[3], double* out) {
    double ib[3], d[3];
    ibox_of(box, ib);
    for (int i = 0; i < n; ++i) {
        minimg(pos + 3 * (size_t)i, ref, box, ib, d);
        for (int k = 0; k < 3; ++k) out[3 * (size_t)i + k] = ref[k] + d[k];
    }
}

/* ---- waterlib.f90:683-703 (returns degrees) ---- */
double wolo_cos_angle3(const double p1[3], const double p2[3], const double p3[3]) {
    const double pi = WOLO_PI;
    const double DegPerRad = 180.0 / pi;
    if ((p1[0] == p2[0] && p1[1] == p2[1] && p1[2] == p2[2]) ||
        (p2[0] == p3[0] && p2[1] == p3[1] && p2[2] == p3[2]))
        return 0.0;
    double v21[3], v23[3];
    for (int k = 0; k < 3; ++k) { v21[k] = p1[k] - p2[k]; v23[k] = p3[k] - p2[k]; }
    double n = sqrt(sumsq3(v21) * sumsq3(v23));
    double dot = (v21[0] * v23[0] + v21[1] * v23[1]) + v21[2] * v23[2];
    double phi = dot / n;
    phi = fmin(1.0, fmax(-1.0, phi));
    phi = acos(phi);
    double r = fmod(phi + pi, pi * 2.0) - pi;
    if (r < -pi) r = r + pi * 2.0;
    return r * DegPerRad;
}

/* ---- waterlib.f90:710-743; mask is (ns,n) Fortran order int32 ---- */
void wolo_near_neighbors(const double* sub, int ns, const double* pos, int n, const double box[3],
                         double low, double high, int32_t* mask) {
    double ib[3], d[3];
    ibox_of(box, ib);
    const double lowsq = low * low, highsq = high * high;
    for (int i = 0; i < ns; ++i)
        for (int j = 0; j < n; ++j) {
            minimg(pos + 3 * (size_t)j, sub + 3 * (size_t)i, box, ib, d);
            double d2 = sumsq3(d);
            mask[(size_t)i + (size_t)j * ns] = (d2 > lowsq && d2 <= highsq) ? 1 : 0;
        }
}

/* ---- waterlib.f90:830-862 ---- */
void wolo_all_near_neighbors(const double* pos, int n, const double box[3], double low, double high,
                             int32_t* mask) {
    double ib[3], d[3];
    ibox_of(box, ib);
    const double lowsq = low * low, highsq = high * high;
    memset(mask, 0, sizeof(int32_t) * (size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            minimg(pos + 3 * (size_t)j, pos + 3 * (size_t)i, box, ib, d);
            double d2 = sumsq3(d);
            if (d2 > lowsq && d2 <= highsq) {
                mask[(size_t)i + (size_t)j * n] = 1;
                mask[(size_t)j + (size_t)i * n] = 1;
            }
        }
}

/* ---- waterlib.f90:867-895; allangs is n*n, diagonal set to NaN (reference leaves it
 *      uninitialised); symmetric so layout is moot ---- */
void wolo_tetra_cos_ang(const double ref[3], const double* neigh, int n, const double box[3],
                        double* allangs) {
    double ib[3];
    ibox_of(box, ib);
    for (int i = 0; i < n; ++i) allangs[(size_t)i * n + i] = NAN;
    for (int i = 0; i < n; ++i) {
        double d1[3], p1[3];
        minimg(neigh + 3 * (size_t)i, ref, box, ib, d1);
        for (int k = 0; k < 3; ++k) p1[k] = ref[k] + d1[k];
        for (int j = i + 1; j < n; ++j) {
            double d2[3], p2[3];
            minimg(neigh + 3 * (size_t)j, ref, box, ib, d2);
            for (int k = 0; k < 3; ++k) p2[k] = ref[k] + d2[k];
            double a = wolo_cos_angle3(p1, ref, p2);
            allangs[(size_t)i * n + j] = a;
            allangs[(size_t)j * n + i] = a;
        }
    }
}

/* ---- waterlib.f90:900-918 ---- */
void wolo_lsi_dists(const double ref[3], const double* neigh, int n, const double box[3], double* dist) {
    double ib[3], d[3];
    ibox_of(box, ib);
    for (int i = 0; i < n; ++i) {
        minimg(neigh + 3 * (size_t)i, ref, box, ib, d);
        dist[i] = sqrt(sumsq3(d));
    }
}

/* ------------------------------------------------------------------------------------------
 * numpy pieces
 * ------------------------------------------------------------------------------------------ */

/* numpy/_core/src/umath/loops_utils.h.src pairwise_sum, contiguous doubles */
static double np_pairwise_sum(const double* a, long n) {
    if (n < 8) {
        double res = -0.0;
        for (long i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        long i;
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

double wolo_np_sum(const double* a, long n) { return n > 0 ? np_pairwise_sum(a, n) : 0.0; }
double wolo_np_mean(const double* a, long n) { return wolo_np_sum(a, n) / (double)n; }

/* np.var (ddof=0): mean, then sum((a-mean)^2)/n (numpy/_core/_methods.py:_var) */
double wolo_np_var(const double* a, long n) {
    double m = wolo_np_mean(a, n);
    double* t = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (long i = 0; i < n; ++i) { double x = a[i] - m; t[i] = x * x; }
    double v = wolo_np_sum(t, n) / (double)n;
    free(t);
    return v;
}

/* np.histogram(a, bins=nbins, range=[lo,hi]) uniform-bin path (numpy/lib/_histograms_impl.py
 * :826-867) with bin_edges = linspace(lo,hi,nbins+1) (edge[i] = i*((hi-lo)/nbins) + lo, last = hi).
 * Accumulates into counts[nbins] (int64). */
void wolo_np_histogram(const double* a, long n, int nbins, double lo, double hi, int64_t* counts) {
    const double step = (hi - lo) / (double)nbins;
    const double denom = hi - lo;
    for (long i = 0; i < n; ++i) {
        double x = a[i];
        if (!(x >= lo) || !(x <= hi)) continue;
        double f = ((x - lo) / denom) * (double)nbins;
        long idx = (long)f;
        if (idx == nbins) idx -= 1;
        double e_lo = (idx == nbins) ? hi : ((double)idx * step + lo);
        if (x < e_lo) idx -= 1;
        double e_hi = (idx + 1 == nbins) ? hi : ((double)(idx + 1) * step + lo);
        if (x >= e_hi && idx != nbins - 1) idx += 1;
        counts[idx] += 1;
    }
}

/* pair-distance histogram rule of waterlib.f90:219-222 (RadialDist): nbin = ceiling(d/binwidth),
 * 1-based bin k covers ((k-1)w, k w]; d with nbin outside [1,nbins] are dropped.  Applied by the
 * trajectory driver to the sorted 4-NN O-O distances. */
void wolo_dist_histogram(const double* d, long n, double binwidth, int nbins, int64_t* counts) {
    for (long i = 0; i < n; ++i) {
        double x = d[i];
        if (!(x == x)) continue; /* NaN padding */
        double c = ceil(x / binwidth);
        if (c >= 1.0 && c <= (double)nbins) counts[(long)c - 1] += 1;
    }
}

/* ------------------------------------------------------------------------------------------
 * getOrderParamq  (water_properties.py:344-391)
 * ------------------------------------------------------------------------------------------ */
typedef struct { double dist; int idx; double p[3]; } cand_t;

static int cand_less(const cand_t* a, const cand_t* b) {
    if (a->dist < b->dist) return 1;
    if (a->dist > b->dist) return 0;
    return a->idx < b->idx;
}

/* keep the 4 smallest (dist, idx) in best[0..*nb-1], sorted ascending */
static inline void top4_insert(cand_t* best, int* nb, const cand_t* c) {
    int pos = *nb;
    if (pos == 4) {
        if (!cand_less(c, &best[3])) return;
        pos = 3;
    } else {
        *nb = pos + 1;
    }
    while (pos > 0 && cand_less(c, &best[pos - 1])) { best[pos] = best[pos - 1]; --pos; }
    best[pos] = *c;
}

/* Evaluate one (centre a, candidate j) pair exactly as the reference does; returns 1 if in mask. */
static inline int eval_pair(const double a[3], const double pj[3], const double box[3],
                            const double ib[3], double lowsq, double highsq, cand_t* c) {
    double d[3];
    minimg(pj, a, box, ib, d);                       /* waterlib.f90:733-735 */
    double d2 = sumsq3(d);
    if (!(d2 > lowsq && d2 <= highsq)) return 0;     /* :737 */
    double e[3];
    for (int k = 0; k < 3; ++k) {
        c->p[k] = a[k] + d[k];                        /* reimage, waterlib.f90:45 (wp:372) */
        e[k] = c->p[k] - a[k];                        /* thisPos - apos (wp:373) */
    }
    c->dist = sqrt(sumsq3(e));                       /* np.linalg.norm(axis=1) */
    return 1;
}

/* angles + q from the (<=4) selected neighbours; wp:376-388 */
static void finish_water(const double a[3], const cand_t* best, int nb, long nmask,
                         const double box[3], double* q, int32_t* nn_idx, double* nn_dist,
                         double* ang_deg) {
    double ang[6];
    int na = 0;
    if (nmask > 0) {
        double neigh[12] = {0}, allangs[16];
        for (int k = 0; k < nb; ++k) memcpy(neigh + 3 * k, best[k].p, sizeof(double) * 3);
        wolo_tetra_cos_ang(a, neigh, nb, box, allangs);
        for (int i = 0; i < nb; ++i)
            for (int j = i + 1; j < nb; ++j) ang[na++] = allangs[i * nb + j];
        if (nmask == 1) { na = 0; }                  /* wp:379-380: all six 180 */
        while (na < 6) ang[na++] = 180.0;            /* wp:381-384 */
        double s = -0.0;
        for (int k = 0; k < 6; ++k) {
            double c = cos(ang[k] * WOLO_PI / 180.0) + (1.0 / 3.0);
            s += c * c;
        }
        *q = 1.0 - (3.0 / 8.0) * s;                  /* wp:388 */
    } else {
        *q = 0.0;                                    /* wp:358 (zeros) */
        for (int k = 0; k < 6; ++k) ang[k] = NAN;
    }
    if (nn_idx) for (int k = 0; k < 4; ++k) nn_idx[k] = (k < nb) ? best[k].idx : -1;
    if (nn_dist) for (int k = 0; k < 4; ++k) nn_dist[k] = (k < nb) ? best[k].dist : NAN;
    if (ang_deg) for (int k = 0; k < 6; ++k) ang_deg[k] = ang[k];
}

/* literal O(ns*n) form -- same pair work as the reference's Fortran mask + numpy selection.
 * Outputs (each nullable except q): q[ns]; nn_idx[ns][4] (global index into pos, -1 padded);
 * nn_dist[ns][4] (the ranking key, NaN padded); ang_deg[ns][6] (the six angles fed to q, 180
 * padded as wp:379-384; NaN if the water has no neighbour). */
int wolo_order_param_q(const double* sub, int ns, const double* pos, int n, const double box[3],
                       double low, double high, double* q, int32_t* nn_idx, double* nn_dist,
                       double* ang_deg) {
    double ib[3];
    ibox_of(box, ib);
    const double lowsq = low * low, highsq = high * high;
    for (int i = 0; i < ns; ++i) {
        const double* a = sub + 3 * (size_t)i;
        cand_t best[4], c;
        int nb = 0;
        long nmask = 0;
        for (int j = 0; j < n; ++j) {
            if (!eval_pair(a, pos + 3 * (size_t)j, box, ib, lowsq, highsq, &c)) continue;
            c.idx = j;
            ++nmask;
            top4_insert(best, &nb, &c);
        }
        finish_water(a, best, nb, nmask, box, q + i, nn_idx ? nn_idx + 4 * (size_t)i : 0,
                     nn_dist ? nn_dist + 4 * (size_t)i : 0, ang_deg ? ang_deg + 6 * (size_t)i : 0);
    }
    return 0;
}

/* ---- cell-list form: identical per-pair arithmetic and ranking, candidates enumerated through
 * a uniform grid over the wrapped coordinates.  Used for CPU parity at sizes where the literal
 * form is too slow, and as the "good CPU code" yardstick.  Requires box[k] > 0 on every axis;
 * returns -1 otherwise (callers then use the literal form). ---- */
typedef struct {
    int nc[3];
    double cw[3];
    int* start; /* ncell+1 */
    int* order; /* n: sorted -> original */
} grid_t;

static inline int cell_coord(double x, double L, int nc) {
    double f = x / L;
    f = f - floor(f);
    int c = (int)(f * nc);
    if (c >= nc) c = nc - 1;
    if (c < 0) c = 0;
    return c;
}

static int grid_build(grid_t* g, const double* pos, int n, const double box[3], double target) {
    for (int k = 0; k < 3; ++k) {
        if (!(box[k] > 0.0)) return -1;
        int nc = (int)floor(box[k] / target);
        if (nc < 1) nc = 1;
        if (nc > 1024) nc = 1024;
        g->nc[k] = nc;
        g->cw[k] = box[k] / nc;
    }
    long ncell = (long)g->nc[0] * g->nc[1] * g->nc[2];
    g->start = (int*)calloc((size_t)ncell + 1, sizeof(int));
    g->order = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int* cid = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        int cx = cell_coord(pos[3 * (size_t)i], box[0], g->nc[0]);
        int cy = cell_coord(pos[3 * (size_t)i + 1], box[1], g->nc[1]);
        int cz = cell_coord(pos[3 * (size_t)i + 2], box[2], g->nc[2]);
        cid[i] = (cz * g->nc[1] + cy) * g->nc[0] + cx;
        g->start[cid[i] + 1]++;
    }
    for (long c = 0; c < ncell; ++c) g->start[c + 1] += g->start[c];
    int* fill = (int*)malloc(sizeof(int) * (size_t)ncell);
    memcpy(fill, g->start, sizeof(int) * (size_t)ncell);
    for (int i = 0; i < n; ++i) g->order[fill[cid[i]]++] = i;
    free(fill);
    free(cid);
    return 0;
}

static void grid_free(grid_t* g) { free(g->start); free(g->order); }

int wolo_order_param_q_cells(const double* sub, int ns, const double* pos, int n,
                             const double box[3], double low, double high, double* q,
                             int32_t* nn_idx, double* nn_dist, double* ang_deg) {
    grid_t g;
    double target = high / 2.0;
    if (target < 2.5) target = 2.5;
    if (grid_build(&g, pos, n, box, target) != 0) return -1;
    double ib[3];
    ibox_of(box, ib);
    const double lowsq = low * low, highsq = high * high;
    int m[3], lo[3], cnt[3];
    for (int k = 0; k < 3; ++k) {
        m[k] = (int)ceil(high / g.cw[k]) + 1; /* +1: margin for wrap/rounding at cell walls */
        if (2 * m[k] + 1 >= g.nc[k]) { lo[k] = 0; cnt[k] = g.nc[k]; m[k] = -1; }
        else { lo[k] = -m[k]; cnt[k] = 2 * m[k] + 1; }
    }
    #pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < ns; ++i) {
        const double* a = sub + 3 * (size_t)i;
        int cc[3];
        for (int k = 0; k < 3; ++k) cc[k] = cell_coord(a[k], box[k], g.nc[k]);
        cand_t best[4], c;
        int nb = 0;
        long nmask = 0;
        for (int iz = 0; iz < cnt[2]; ++iz) {
            int cz = (m[2] < 0) ? iz : ((cc[2] + lo[2] + iz) % g.nc[2] + g.nc[2]) % g.nc[2];
            for (int iy = 0; iy < cnt[1]; ++iy) {
                int cy = (m[1] < 0) ? iy : ((cc[1] + lo[1] + iy) % g.nc[1] + g.nc[1]) % g.nc[1];
                for (int ix = 0; ix < cnt[0]; ++ix) {
                    int cx = (m[0] < 0) ? ix : ((cc[0] + lo[0] + ix) % g.nc[0] + g.nc[0]) % g.nc[0];
                    long cell = ((long)cz * g.nc[1] + cy) * g.nc[0] + cx;
                    for (int s = g.start[cell]; s < g.start[cell + 1]; ++s) {
                        int j = g.order[s];
                        if (!eval_pair(a, pos + 3 * (size_t)j, box, ib, lowsq, highsq, &c)) continue;
                        c.idx = j;
                        ++nmask;
                        top4_insert(best, &nb, &c);
                    }
                }
            }
        }
        finish_water(a, best, nb, nmask, box, q + i, nn_idx ? nn_idx + 4 * (size_t)i : 0,
                     nn_dist ? nn_dist + 4 * (size_t)i : 0, ang_deg ? ang_deg + 6 * (size_t)i : 0);
    }
    grid_free(&g);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * getCosAngs (water_properties.py:210-250).  Two-pass: call with ang == NULL to get the total
 * number of angles (return value) and num_angs[ns] (= neighbours in shell, wp:248), then with a
 * buffer of that size.  Angles are emitted water by water, row-major upper triangle of the
 * tetraCosAng matrix over the shell neighbours in ascending global index (wp:245-247).
 * ------------------------------------------------------------------------------------------ */
long wolo_cos_angs(const double* sub, int ns, const double* pos, int n, const double box[3],
                   double low, double high, double* ang, double* num_angs) {
    double ib[3], d[3];
    ibox_of(box, ib);
    const double lowsq = low * low, highsq = high * high;
    long total = 0;
    int cap = 64;
    double* neigh = (double*)malloc(sizeof(double) * 3 * (size_t)cap);
    for (int i = 0; i < ns; ++i) {
        const double* a = sub + 3 * (size_t)i;
        int k = 0;
        for (int j = 0; j < n; ++j) {
            minimg(pos + 3 * (size_t)j, a, box, ib, d);
            double d2 = sumsq3(d);
            if (d2 > lowsq && d2 <= highsq) {
                if (k == cap) { cap *= 2; neigh = (double*)realloc(neigh, sizeof(double) * 3 * (size_t)cap); }
                memcpy(neigh + 3 * (size_t)k, pos + 3 * (size_t)j, sizeof(double) * 3);
                ++k;
            }
        }
        if (num_angs) num_angs[i] = (double)k;
        if (k > 0 && ang) {
            /* tetraCosAng re-images every neighbour about the centre itself (waterlib.f90:882-890) */
            double* P = (double*)malloc(sizeof(double) * 3 * (size_t)k);
            for (int u = 0; u < k; ++u) {
                minimg(neigh + 3 * (size_t)u, a, box, ib, d);
                for (int c = 0; c < 3; ++c) P[3 * (size_t)u + c] = a[c] + d[c];
            }
            for (int u = 0; u < k; ++u)
                for (int v = u + 1; v < k; ++v)
                    ang[total++] = wolo_cos_angle3(P + 3 * (size_t)u, a, P + 3 * (size_t)v);
            free(P);
        } else {
            total += (long)k * (k - 1) / 2;
        }
    }
    free(neigh);
    return total;
}

/* ------------------------------------------------------------------------------------------
 * getLSI (water_properties.py:252-311).  lsi_full[ns] receives the LSI of water i or NaN when
 * the reference skips it (wp:283-284,295-296,308-309); num_lsi[ns] as wp:306 (0 when skipped).
 * Returns the number of non-skipped waters (= len(lsiVals)).
 * ------------------------------------------------------------------------------------------ */
static int cmp_double(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

long wolo_lsi(const double* sub, int ns, const double* pos, int n, const double box[3], double low,
              double high, double* lsi_full, double* num_lsi) {
    double ib[3], d[3];
    ibox_of(box, ib);
    const double lowsq = low * low, highsq = high * high;
    const double high2 = high + 3.7;                 /* wp:276,281 */
    const double high2sq = high2 * high2;
    long nvalid = 0;
    int cap = 64;
    double* dist = (double*)malloc(sizeof(double) * (size_t)cap);
    for (int i = 0; i < ns; ++i) {
        const double* a = sub + 3 * (size_t)i;
        int k = 0;
        int have_next = 0;
        double next_raw = 0.0;
        int next_j = -1;
        for (int j = 0; j < n; ++j) {
            const double* pj = pos + 3 * (size_t)j;
            minimg(pj, a, box, ib, d);
            double d2 = sumsq3(d);
            if (d2 > lowsq && d2 <= highsq) {
                if (k + 2 > cap) { cap *= 2; dist = (double*)realloc(dist, sizeof(double) * (size_t)cap); }
                dist[k++] = sqrt(d2);                 /* lsiDists, waterlib.f90:916 */
            }
            if (d2 > highsq && d2 <= high2sq) {
                /* wp:289-291: RAW (not minimum-image) distance picks the "next" neighbour;
                 * np.argmin keeps the first minimum */
                double r[3] = { pj[0] - a[0], pj[1] - a[1], pj[2] - a[2] };
                double raw = sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
                if (!have_next || raw < next_raw) { have_next = 1; next_raw = raw; next_j = j; }
            }
        }
        lsi_full[i] = NAN;
        if (num_lsi) num_lsi[i] = 0.0;
        if (k > 1 && have_next) {
            minimg(pos + 3 * (size_t)next_j, a, box, ib, d);
            dist[k] = sqrt(sumsq3(d));
            int m = k + 1;
            qsort(dist, (size_t)m, sizeof(double), cmp_double);   /* np.sort */
            double* delta = (double*)malloc(sizeof(double) * (size_t)(m - 1));
            for (int u = 0; u < m - 1; ++u) delta[u] = dist[u + 1] - dist[u];
            lsi_full[i] = wolo_np_var(delta, m - 1);
            if (num_lsi) num_lsi[i] = (double)(m - 1);
            free(delta);
            ++nvalid;
        }
    }
    free(dist);
    return nvalid;
}

/* ------------------------------------------------------------------------------------------
 * tetOrderCalc per-trajectory reduction (orderParam_lib.py:1458-1495) over T frames that all
 * have n waters: per-frame mean/var of q, the 500-bin q histogram and (build-side addition,
 * SURVEY 8d config 3) the 4-NN O-O distance histogram.  use_cells != 0 selects the cell-list
 * enumeration (falls back to literal if a box axis is non-periodic).
 * ------------------------------------------------------------------------------------------ */
int wolo_tet_order_traj(const double* frames, const double* boxes, int T, int n, double low,
                        double high, int use_cells, double* mean, double* var, int64_t* qhist,
                        int qbins, int64_t* dhist, int dbins, double dbinwidth) {
    double* q = (double*)malloc(sizeof(double) * (size_t)n);
    double* nd = (double*)malloc(sizeof(double) * 4 * (size_t)n);
    for (int t = 0; t < T; ++t) {
        const double* p = frames + (size_t)t * n * 3;
        const double* b = boxes + (size_t)t * 3;
        int rc = -1;
        if (use_cells) rc = wolo_order_param_q_cells(p, n, p, n, b, low, high, q, 0, nd, 0);
        if (rc != 0) wolo_order_param_q(p, n, p, n, b, low, high, q, 0, nd, 0);
        if (mean) mean[t] = wolo_np_mean(q, n);
        if (var) var[t] = wolo_np_var(q, n);
        if (qhist) wolo_np_histogram(q, n, qbins, 0.0, 1.0, qhist);
        if (dhist) wolo_dist_histogram(nd, 4L * n, dbinwidth, dbins, dhist);
    }
    free(q);
    free(nd);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Willard-Chandler density field.
 * waterlib.f90:1351-1398 WillardDensityPoints (and :1286-1341 WillardDensityField, the same
 * per-point arithmetic on a grid): truncated-shifted Gaussian density and its normalised gradient.
 * The sum runs over ALL waters in index order, as in the reference.  densnorms is [npts][3]
 * (row-major); points without any water inside 3*smoothlen get NaN normals (0/0), as the reference.
 * ------------------------------------------------------------------------------------------ */
void wolo_willard_density_points(const double* pos, int n, const double* pts, int npts, const double box[3],
                                 double smoothlen, double* densvals, double* densnorms) {
    const double pi = WOLO_PI;
    double ib[3];
    ibox_of(box, ib);
    const double s2 = smoothlen * smoothlen;
    const double pref = pow(2.0 * pi * s2, 1.5);
    const double shiftterm = exp(-9.0 / 2.0) / pref;
    const double cut = 9.0 * (smoothlen * smoothlen);
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < npts; ++i) {
        const double* a = pts + 3 * (size_t)i;
        double densval = 0.0, nv[3] = {0.0, 0.0, 0.0};
        for (int l = 0; l < n; ++l) {
            double v[3];
            for (int k = 0; k < 3; ++k) {
                double t = a[k] - pos[3 * (size_t)l + k];        /* thisvec = apos - watpos */
                v[k] = t - box[k] * round(t * ib[k]);
            }
            const double r2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
            if (r2 >= cut) continue;                              /* contributes exactly 0 */
            const double expterm = -r2 / (2.0 * s2);
            const double densfunc = exp(expterm) / pref - shiftterm;
            densval += densfunc;
            for (int k = 0; k < 3; ++k) nv[k] += -v[k] * (densfunc + shiftterm) / s2;
        }
        densvals[i] = densval;
        const double nn = sqrt((nv[0] * nv[0] + nv[1] * nv[1]) + nv[2] * nv[2]);
        for (int k = 0; k < 3; ++k) densnorms[3 * (size_t)i + k] = nv[k] / nn;
    }
}

/* ------------------------------------------------------------------------------------------
 * radial distribution functions / pair-distance histograms.
 * waterlib.f90:316-357 RadialDistSame (pairs i<j of one set), :193-231 RadialDist (every Pos2 x Pos1
 * pair), :363-389 PairDistanceHistogram (Dim = 3 here; zero distances skipped).  All share
 *   distVec = jPos - iPos;  distVec -= BoxL*anint(distVec*iBoxL);  dist = sqrt(sum(distVec**2));
 *   nbin = ceiling(dist/binwidth);  counted when nbin <= totbins
 * and the two g(r) routines normalise with single-precision (4./3.) and pi = 3.141592653589D0:
 *   rdf(k) = counts(k) / (NPos * BulkDens * (4./3.) * pi * binwidth**3 * (k**3 - (k-1)**3)).
 * A zero distance gives nbin = 0, which the reference writes out of bounds (counts(0)); it is skipped here.
 * ------------------------------------------------------------------------------------------ */
static inline long rdf_bin(const double a[3], const double b[3], const double box[3], const double ib[3], double binwidth) {
    double v[3];
    for (int k = 0; k < 3; ++k) {
        const double t = b[k] - a[k];
        v[k] = t - box[k] * round(t * ib[k]);
    }
    const double dist = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    return (long)ceil(dist / binwidth);
}

static void rdf_normalise(const int64_t* counts, int totbins, int npos, double bulkdens, double binwidth, double* rdf) {
    const double pi = 3.141592653589;
    const double fourthirds = (double)(4.f / 3.f);
    for (int k = 1; k <= totbins; ++k) {
        const int shell = k * k * k - (k - 1) * (k - 1) * (k - 1);   /* integer arithmetic, as the Fortran */
        const double den = (double)npos * bulkdens * fourthirds * pi * (binwidth * binwidth * binwidth) * (double)shell;
        rdf[k - 1] = (double)counts[k - 1] / den;
    }
}

void wolo_pair_counts(const double* pos1, int n1, const double* pos2, int n2, int same, double binwidth, int totbins,
                      const double box[3], int64_t* counts) {
    double ib[3];
    ibox_of(box, ib);
    for (int k = 0; k < totbins; ++k) counts[k] = 0;
    #pragma omp parallel
    {
        int64_t* loc = (int64_t*)calloc((size_t)totbins, sizeof(int64_t));
        #pragma omp for schedule(dynamic, 16)
        for (int i = 0; i < n2; ++i)
            for (int j = same ? i + 1 : 0; j < n1; ++j) {
                const long nb = rdf_bin(pos2 + 3 * (size_t)i, pos1 + 3 * (size_t)j, box, ib, binwidth);
                if (nb >= 1 && nb <= totbins) loc[nb - 1] += 1;
            }
        #pragma omp critical
        for (int k = 0; k < totbins; ++k) counts[k] += loc[k];
        free(loc);
    }
}

void wolo_radial_dist_same(const double* pos, int n, double binwidth, int totbins, double bulkdens, const double box[3],
                           double* rdf) {
    int64_t* c = (int64_t*)malloc(sizeof(int64_t) * (size_t)totbins);
    wolo_pair_counts(pos, n, pos, n, 1, binwidth, totbins, box, c);
    rdf_normalise(c, totbins, n, bulkdens, binwidth, rdf);
    free(c);
}

void wolo_radial_dist(const double* pos1, int n1, const double* pos2, int n2, double binwidth, int totbins,
                      double bulkdens, const double box[3], double* rdf) {
    int64_t* c = (int64_t*)malloc(sizeof(int64_t) * (size_t)totbins);
    wolo_pair_counts(pos1, n1, pos2, n2, 0, binwidth, totbins, box, c);
    rdf_normalise(c, totbins, n1, bulkdens, binwidth, rdf);
    free(c);
}

void wolo_pair_distance_histogram(const double* pos1, int n1, const double* pos2, int n2, double binwidth, int totbins,
                                  const double box[3], double* hist) {
    int64_t* c = (int64_t*)malloc(sizeof(int64_t) * (size_t)totbins);
    wolo_pair_counts(pos2, n2, pos1, n1, 0, binwidth, totbins, box, c);   /* counts are symmetric in the two sets */
    for (int k = 0; k < totbins; ++k) hist[k] = (double)c[k];
    free(c);
}

/* ------------------------------------------------------------------------------------------
 * water orientation.  waterlib.f90:954-965 AngBetween, :973-1011 watOrient (angle of the
 * dipole = sum of the two imaged O-H vectors, and of the molecular-plane normal = OH1 x OH2, to a reference
 * direction, degrees), :1019-1044 watOHvec (unit dipole).  hpos holds the two hydrogens of water i in rows
 * 2i, 2i+1.
 * ------------------------------------------------------------------------------------------ */
static double ang_between(const double a[3], const double b[3]) {
    const double pi = WOLO_PI;
    const double DegPerRad = 180.0 / pi;
    double phi = fmin(1.0, fmax(-1.0, (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]));
    phi = acos(phi);
    double r = fmod(phi + pi, pi * 2.0) - pi;
    if (r < -pi) r = r + pi * 2.0;
    return r * DegPerRad;
}

static void oh_vectors(const double* o, const double* h1, const double* h2, const double box[3], const double ib[3],
                       double v1[3], double v2[3], double dip[3]) {
    for (int k = 0; k < 3; ++k) {
        double t = h1[k] - o[k];
        v1[k] = t - box[k] * round(t * ib[k]);
        t = h2[k] - o[k];
        v2[k] = t - box[k] * round(t * ib[k]);
        t = v1[k] + v2[k];
        dip[k] = t - box[k] * round(t * ib[k]);
    }
}

void wolo_wat_orient(const double* opos, const double* hpos, int n, const double refvec[3], const double box[3],
                     double* angdip, double* angplane) {
    double ib[3], ref[3];
    ibox_of(box, ib);
    const double rn = sqrt((refvec[0] * refvec[0] + refvec[1] * refvec[1]) + refvec[2] * refvec[2]);
    for (int k = 0; k < 3; ++k) ref[k] = refvec[k] / rn;
    for (int i = 0; i < n; ++i) {
        double v1[3], v2[3], dip[3], pl[3], u[3];
        oh_vectors(opos + 3 * (size_t)i, hpos + 6 * (size_t)i, hpos + 6 * (size_t)i + 3, box, ib, v1, v2, dip);
        const double nd = sqrt(sumsq3(dip));
        for (int k = 0; k < 3; ++k) u[k] = dip[k] / nd;
        angdip[i] = ang_between(u, ref);
        pl[0] = v1[1] * v2[2] - v1[2] * v2[1];
        pl[1] = v1[2] * v2[0] - v1[0] * v2[2];
        pl[2] = v1[0] * v2[1] - v1[1] * v2[0];
        const double np_ = sqrt(sumsq3(pl));
        for (int k = 0; k < 3; ++k) u[k] = pl[k] / np_;
        angplane[i] = ang_between(u, ref);
    }
}

void wolo_wat_oh_vec(const double* opos, const double* hpos, int n, const double box[3], double* normdip) {
    double ib[3];
    ibox_of(box, ib);
    for (int i = 0; i < n; ++i) {
        double v1[3], v2[3], dip[3];
        oh_vectors(opos + 3 * (size_t)i, hpos + 6 * (size_t)i, hpos + 6 * (size_t)i + 3, box, ib, v1, v2, dip);
        const double nd = sqrt(sumsq3(dip));
        for (int k = 0; k < 3; ++k) normdip[3 * (size_t)i + k] = dip[k] / nd;
    }
}

/* ------------------------------------------------------------------------------------------
 * hydrogen bonds.  waterlib.f90:1156-1210 generalHbonds: acceptor i and donor j are
 * bonded when 1.0E-2 < |D - A|^2 <= distCut^2 (minimum image; the lower bound is a single-precision literal) and
 * the angle at the donor hydrogen between (A - H) and (D - H), both imaged and normalised, is >= angCut degrees.
 * bond is (nacc, ndon) in Fortran order, int32 as f2py returns logicals.
 * ------------------------------------------------------------------------------------------ */
int wolo_hbond_pair(const double* a, const double* d, const double* h, const double box[3], const double ib[3],
                    double distcutsq, double angcut) {
    double v[3], av[3], dv[3];
    for (int k = 0; k < 3; ++k) {
        const double t = d[k] - a[k];
        v[k] = t - box[k] * round(t * ib[k]);
    }
    const double dsq = sumsq3(v);
    if (dsq > distcutsq || dsq <= (double)1.0E-2f) return 0;
    for (int k = 0; k < 3; ++k) {
        double t = a[k] - h[k];
        av[k] = t - box[k] * round(t * ib[k]);
        t = d[k] - h[k];
        dv[k] = t - box[k] * round(t * ib[k]);
    }
    const double na = sqrt(sumsq3(av)), nd = sqrt(sumsq3(dv));
    for (int k = 0; k < 3; ++k) { av[k] = av[k] / na; dv[k] = dv[k] / nd; }
    return ang_between(av, dv) < angcut ? 0 : 1;
}

void wolo_general_hbonds(const double* acc, int nacc, const double* don, const double* donh, int ndon,
                         const double box[3], double distcut, double angcut, int32_t* bond) {
    double ib[3];
    ibox_of(box, ib);
    const double dcs = distcut * distcut;
    #pragma omp parallel for schedule(static)
    for (int j = 0; j < ndon; ++j)
        for (int i = 0; i < nacc; ++i)
            bond[(size_t)j * nacc + i] = wolo_hbond_pair(acc + 3 * (size_t)i, don + 3 * (size_t)j, donh + 3 * (size_t)j,
                                                         box, ib, dcs, angcut);
}
