// gather_probe.hip -- what do the certification kernel's neighbour gathers cost, and which access shape is cheapest?
// (design input for k_finish_fast; build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe)
// F frames of n 32-byte records (x, y, z, index) in "sorted" order; every water reads its own record and four
// neighbours' whose slots lie near its own (same / adjacent cell rows: +-300 slots; adjacent layers: +-3200), as the
// proposals of k_search_rows do.  Blocks b, b + 8, ... (one XCD) walk the same frames, as in the library.
//   A  one lane per water, a neighbour = one 32-byte record = two global_load_dwordx4 (today's k_finish_fast)
//   B  two lanes per water, each lane loads one 16-byte half of every record: adjacent lanes share a line
//   C  one lane per water, a neighbour = 24 bytes of an [n][3] array of doubles (reading the source frame instead)
//   D  one lane per water, 16-byte records (what a neighbour would cost if it fitted a single dwordx4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k_gather(const double4* __restrict__ rec, const double* __restrict__ rows3, const float4* __restrict__ rec16,
                                                const int4* __restrict__ prop, int n, int F, double* __restrict__ out) {
    extern __shared__ int pad[];
    (void)pad;
    const int xcd = blockIdx.x & 7, kb = blockIdx.x >> 3, KB = gridDim.x >> 3;
    double acc = 0.0;
    for (int f = xcd; f < F; f += 8) {
        const double4* __restrict__ sp = rec + (int64_t)f * n;
        const double* __restrict__ r3 = rows3 + (int64_t)f * n * 3;
        const float4* __restrict__ s16 = rec16 + (int64_t)f * n;
        const int4* __restrict__ pp = prop + (int64_t)f * n;
        if (KIND == 1) {
            const int h = threadIdx.x & 1;
            for (int s0 = kb * 128; s0 < n; s0 += KB * 128) {
                const int s = min(s0 + (int)(threadIdx.x >> 1), n - 1);
                const int4 pr = pp[s];
                const double2* __restrict__ hp = reinterpret_cast<const double2*>(sp) + h;
                const double2 me = hp[2 * (int64_t)s];
                const double2 a = hp[2 * (int64_t)pr.x], b = hp[2 * (int64_t)pr.y], c = hp[2 * (int64_t)pr.z], d = hp[2 * (int64_t)pr.w];
                acc += (me.x + me.y) + ((a.x + a.y) + (b.x + b.y)) + ((c.x + c.y) + (d.x + d.y));
            }
        } else {
            for (int s0 = kb * 256; s0 < n; s0 += KB * 256) {
                const int s = min(s0 + (int)threadIdx.x, n - 1);
                const int4 pr = pp[s];
                if (KIND == 0) {
                    const double4 me = sp[s];
                    const double4 a = sp[pr.x], b = sp[pr.y], c = sp[pr.z], d = sp[pr.w];
                    acc += (me.x + me.y + me.z) + ((a.x + a.y + a.z) + (b.x + b.y + b.z)) + ((c.x + c.y + c.z) + (d.x + d.y + d.z));
                } else if (KIND == 2) {
                    const double* m = r3 + 3 * (int64_t)s;
                    const double* a = r3 + 3 * (int64_t)pr.x; const double* b = r3 + 3 * (int64_t)pr.y;
                    const double* c = r3 + 3 * (int64_t)pr.z; const double* d = r3 + 3 * (int64_t)pr.w;
                    acc += (m[0] + m[1] + m[2]) + ((a[0] + a[1] + a[2]) + (b[0] + b[1] + b[2])) + ((c[0] + c[1] + c[2]) + (d[0] + d[1] + d[2]));
                } else {
                    const float4 me = s16[s];
                    const float4 a = s16[pr.x], b = s16[pr.y], c = s16[pr.z], d = s16[pr.w];
                    acc += (double)((me.x + me.y + me.z) + ((a.x + a.y + a.z) + (b.x + b.y + b.z)) + ((c.x + c.y + c.z) + (d.x + d.y + d.z)));
                }
            }
        }
    }
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int n = 100000, F = argc > 1 ? atoi(argv[1]) : 256;
    std::vector<int4> hp((size_t)n);
    unsigned long long st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(st >> 33); };
    for (int s = 0; s < n; ++s) {
        int v[4];
        for (int k = 0; k < 4; ++k) {
            const unsigned r = rnd();
            int off = (int)(r % 601) - 300;
            if ((r >> 12) % 3 == 0) off += ((r >> 20) & 1) ? 3200 : -3200;   // a neighbour in the layer above / below
            v[k] = ((s + off) % n + n) % n;
        }
        hp[(size_t)s] = make_int4(v[0], v[1], v[2], v[3]);
    }
    double4* rec; double* rows3; float4* rec16; int4* prop; double* out;
    CK(hipMalloc(&rec, sizeof(double4) * (size_t)F * n));
    CK(hipMalloc(&rows3, sizeof(double) * 3 * (size_t)F * n));
    CK(hipMalloc(&rec16, sizeof(float4) * (size_t)F * n));
    CK(hipMalloc(&prop, sizeof(int4) * (size_t)F * n));
    CK(hipMalloc(&out, sizeof(double) * 256 * 8192));
    CK(hipMemset(rec, 0, sizeof(double4) * (size_t)F * n));
    CK(hipMemset(rows3, 0, sizeof(double) * 3 * (size_t)F * n));
    CK(hipMemset(rec16, 0, sizeof(float4) * (size_t)F * n));
    for (int f = 0; f < F; ++f) CK(hipMemcpy(prop + (size_t)f * n, hp.data(), sizeof(int4) * (size_t)n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"A lane per water, 32-B records (2 x dwordx4)", "B two lanes per water, 16-B halves", "C lane per water, 24-B rows of [n][3]",
                            "D lane per water, 16-B records"};
    for (int wps = 4; wps <= 8; wps += 4) {
        const int per_cu = wps;                       // 256-thread blocks per CU = waves per SIMD
        const size_t lds = (160 * 1024 / per_cu) - 1024;
        const int grid = per_cu * 256;
        for (int kind = 0; kind < 4; ++kind) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (kind == 0) hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(256), lds, 0, rec, rows3, rec16, prop, n, F, out);
                if (kind == 1) hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), lds, 0, rec, rows3, rec16, prop, n, F, out);
                if (kind == 2) hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(256), lds, 0, rec, rows3, rec16, prop, n, F, out);
                if (kind == 3) hipLaunchKernelGGL(k_gather<3>, dim3(grid), dim3(256), lds, 0, rec, rows3, rec16, prop, n, F, out);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("%d waves/SIMD  %-48s %8.1f us per %d frames x %d waters\n", wps, names[kind], best * 1e3f, F, n);
        }
    }
    return 0;
}
