/* TEST INFRASTRUCTURE ONLY: runs the CPU oracle's entry points (oracle/wol_oracle.c) under AddressSanitizer and
 * UndefinedBehaviorSanitizer (sanitizer build of the host-side code; GPU sanitizers are not available on
 * this pool).  tests/test_oracle_sanitized.py builds it with
 *     gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fopenmp oracle/wol_oracle.c oracle/asan_driver.c -lm
 * feeds it a golden frame and compares what it writes with the regular oracle build, bit for bit.
 *     asan_driver <in.bin> <out.bin>
 * in.bin : int32 n, int32 ns | box[3] | pos[n][3] | sub[ns][3]          (doubles)
 * out.bin: q_lit[ns] q_cells[ns] | idx[ns][4] as doubles | dist[ns][4] | ang[ns][6] | n_ang, angles... | lsi_full[ns] |
 *          mask sums | rdf[60] | hbonds sum | traj mean[2] var[2] qhist[500] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int wolo_set_threads(int n);
int wolo_order_param_q(const double*, int, const double*, int, const double*, double, double, double*, int32_t*, double*, double*);
int wolo_order_param_q_cells(const double*, int, const double*, int, const double*, double, double, double*, int32_t*, double*, double*);
long wolo_cos_angs(const double*, int, const double*, int, const double*, double, double, double*, double*);
long wolo_lsi(const double*, int, const double*, int, const double*, double, double, double*, double*);
void wolo_near_neighbors(const double*, int, const double*, int, const double*, double, double, int32_t*);
void wolo_all_near_neighbors(const double*, int, const double*, double, double, int32_t*);
void wolo_radial_dist_same(const double*, int, double, int, double, const double*, double*);
void wolo_general_hbonds(const double*, int, const double*, const double*, int, const double*, double, double, int32_t*);
int wolo_tet_order_traj(const double*, const double*, int, int, double, double, int, double*, double*, int64_t*, int, int64_t*, int, double);
void wolo_willard_density_points(const double*, int, const double*, int, const double*, double, double*, double*);

static void put(FILE* f, const double* a, size_t n) { if (fwrite(a, sizeof(double), n, f) != n) exit(3); }

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    FILE* fi = fopen(argv[1], "rb");
    if (!fi) return 2;
    int32_t hdr[2];
    if (fread(hdr, sizeof(int32_t), 2, fi) != 2) return 2;
    const int n = hdr[0], ns = hdr[1];
    double box[3];
    double* pos = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    double* sub = (double*)malloc(sizeof(double) * 3 * (size_t)ns);
    if (fread(box, sizeof(double), 3, fi) != 3 || fread(pos, sizeof(double), 3 * (size_t)n, fi) != 3 * (size_t)n ||
        fread(sub, sizeof(double), 3 * (size_t)ns, fi) != 3 * (size_t)ns) return 2;
    fclose(fi);
    FILE* fo = fopen(argv[2], "wb");
    if (!fo) return 2;
    wolo_set_threads(2);
    double* q = (double*)malloc(sizeof(double) * (size_t)ns);
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)ns);
    double* dist = (double*)malloc(sizeof(double) * 4 * (size_t)ns);
    double* ang = (double*)malloc(sizeof(double) * 6 * (size_t)ns);
    wolo_order_param_q(sub, ns, pos, n, box, 0.0, 10.0, q, idx, dist, ang);
    put(fo, q, (size_t)ns);
    wolo_order_param_q_cells(sub, ns, pos, n, box, 0.0, 10.0, q, idx, dist, ang);
    put(fo, q, (size_t)ns);
    double* idxd = (double*)malloc(sizeof(double) * 4 * (size_t)ns);
    for (size_t k = 0; k < 4 * (size_t)ns; ++k) idxd[k] = (double)idx[k];
    put(fo, idxd, 4 * (size_t)ns); put(fo, dist, 4 * (size_t)ns); put(fo, ang, 6 * (size_t)ns);
    double* numa = (double*)malloc(sizeof(double) * (size_t)ns);
    const long tot = wolo_cos_angs(sub, ns, pos, n, box, 0.0, 3.413, NULL, numa);
    double* angs = (double*)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
    wolo_cos_angs(sub, ns, pos, n, box, 0.0, 3.413, angs, numa);
    double dtot = (double)tot;
    put(fo, &dtot, 1); put(fo, angs, (size_t)tot);
    double* lsi = (double*)malloc(sizeof(double) * (size_t)ns);
    wolo_lsi(sub, ns, pos, n, box, 0.0, 3.7, lsi, numa);
    put(fo, lsi, (size_t)ns);
    int32_t* mask = (int32_t*)malloc(sizeof(int32_t) * (size_t)n * (size_t)(ns > n ? ns : n));
    wolo_near_neighbors(sub, ns, pos, n, box, 0.0, 3.5, mask);
    double s1 = 0.0, s2 = 0.0;
    for (size_t k = 0; k < (size_t)n * ns; ++k) s1 += mask[k];
    wolo_all_near_neighbors(pos, n, box, 0.0, 3.5, mask);
    for (size_t k = 0; k < (size_t)n * n; ++k) s2 += mask[k];
    put(fo, &s1, 1); put(fo, &s2, 1);
    double rdf[60];
    wolo_radial_dist_same(pos, n, 0.1, 60, 1.0, box, rdf);
    put(fo, rdf, 60);
    /* hydrogen bonds: the first ns waters accept, every water donates through a made-up hydrogen 0.96 A along x */
    double* donh = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    for (int i = 0; i < n; ++i) { donh[3 * i] = pos[3 * i] + 0.96; donh[3 * i + 1] = pos[3 * i + 1]; donh[3 * i + 2] = pos[3 * i + 2]; }
    wolo_general_hbonds(sub, ns, pos, donh, n, box, 3.5, 120.0, mask);
    double s3 = 0.0;
    for (size_t k = 0; k < (size_t)n * ns; ++k) s3 += mask[k];
    put(fo, &s3, 1);
    /* two-frame trajectory (the frame and its copy) */
    double* fr = (double*)malloc(sizeof(double) * 6 * (size_t)n);
    for (size_t k = 0; k < 3 * (size_t)n; ++k) { fr[k] = pos[k]; fr[3 * (size_t)n + k] = pos[k]; }
    double boxes[6] = {box[0], box[1], box[2], box[0], box[1], box[2]}, mean[2], var[2], hq[500];
    int64_t qh[500] = {0}, dh[1000] = {0};
    wolo_tet_order_traj(fr, boxes, 2, n, 0.0, 10.0, 1, mean, var, qh, 500, dh, 1000, 0.01);
    for (int k = 0; k < 500; ++k) hq[k] = (double)qh[k];
    put(fo, mean, 2); put(fo, var, 2); put(fo, hq, 500);
    double dens[64], norms[192];
    wolo_willard_density_points(pos, n, sub, ns < 64 ? ns : 64, box, 2.4, dens, norms);
    put(fo, dens, ns < 64 ? ns : 64);
    fclose(fo);
    free(pos); free(sub); free(q); free(idx); free(dist); free(ang); free(idxd); free(numa); free(angs); free(lsi); free(mask);
    free(donh); free(fr);
    return 0;
}
