// pack_probe.hip -- how fast can host threads pack the O rows of TIP4P fp64 frames (24 of every 96 bytes) into a staging
// buffer, by source memory (malloc / hipHostMalloc), destination (malloc / hipHostMalloc) and thread count?  Design input
// for the host-frame feeder (option host_pack).  build: hipcc --offload-arch=gfx950 -O3 tools/pack_probe.hip -o tools/pack_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static void pack(const double* src, double* dst, long rows, long stride) {
    for (long r = 0; r < rows; ++r) { dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst += 3; src += stride; }
}

int main() {
    const long n = 100000, F = 64, stride = 12, rows = n * F;
    const size_t sb = sizeof(double) * rows * stride, db = sizeof(double) * rows * 3;
    double* src_m = (double*)aligned_alloc(2 << 20, ((sb + (2 << 20) - 1) / (2 << 20)) * (2 << 20));
    double* dst_m = (double*)aligned_alloc(2 << 20, ((db + (2 << 20) - 1) / (2 << 20)) * (2 << 20));
    double *src_p = nullptr, *dst_p = nullptr;
    if (hipHostMalloc(&src_p, sb, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&dst_p, db, hipHostMallocDefault) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
    for (size_t i = 0; i < sb / 8; ++i) src_m[i] = (double)i;
    memcpy(src_p, src_m, sb);
    memset(dst_m, 0, db); memset(dst_p, 0, db);
    struct Case { const char* name; double* s; double* d; } cases[] = {{"malloc -> malloc", src_m, dst_m}, {"malloc -> pinned", src_m, dst_p},
                                                                       {"pinned -> pinned", src_p, dst_p}, {"pinned -> malloc", src_p, dst_m}};
    for (auto& cs : cases)
        for (int nt : {1, 2, 4, 8, 16, 32}) {
            double best = 1e30;
            for (int rep = 0; rep < 4; ++rep) {
                auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> th;
                for (int i = 0; i < nt; ++i) {
                    const long r0 = rows * i / nt, r1 = rows * (i + 1) / nt;
                    th.emplace_back(pack, cs.s + r0 * stride, cs.d + r0 * 3, r1 - r0, stride);
                }
                for (auto& t : th) t.join();
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (dt < best) best = dt;
            }
            printf("%-18s %2d threads: %7.0f frames/s  (%.1f GB/s of source lines)\n", cs.name, nt, F / best, sb / best / 1e9);
        }
    return 0;
}
