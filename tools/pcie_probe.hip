// pcie_probe.hip -- how fast can the O rows of interleaved host frames reach HBM?  (design input for the host-trajectory
// path of wol_q_traj; build: hipcc --offload-arch=gfx950 -O3 tools/pcie_probe.hip -o tools/pcie_probe)
//   (1) one linear hipMemcpyAsync of the whole frames (pinned / pageable source)
//   (2) hipMemcpy2DAsync: width 24 B (one O row), source pitch 96 B (TIP4P water), destination packed
//   (3) a gather kernel reading the O rows straight out of pinned (device-mapped) host memory
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_gather_rows(const double* __restrict__ src, int64_t stride, int64_t rows, double* __restrict__ dst) {
    // three lanes per row: lane k of a row moves component k (24 contiguous bytes per row, rows `stride` doubles apart)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = rows * 3;
    for (int64_t e = i; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / 3, k = e - 3 * r;
        dst[e] = src[r * stride + k];
    }
}

// whole 64-byte lines: every lane moves 16 B of the interleaved frame (the O rows are picked out later, on the device)
__global__ void k_copy_lines(const double2* __restrict__ src, int64_t n16, double2* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t e = i; e < n16; e += (int64_t)gridDim.x * blockDim.x) dst[e] = src[e];
}

int main(int argc, char** argv) {
    const int n = 100000, lw = argc > 1 ? atoi(argv[1]) : 4, F = argc > 2 ? atoi(argv[2]) : 64;
    const int64_t rows = (int64_t)n * F, stride = 3 * lw;
    const size_t full = sizeof(double) * rows * stride, packed = sizeof(double) * rows * 3;
    double *pin = nullptr, *dev_full = nullptr, *dev_packed = nullptr;
    CK(hipHostMalloc((void**)&pin, full, hipHostMallocDefault));
    std::vector<double> pageable(rows * stride);
    for (int64_t i = 0; i < rows * stride; ++i) { pin[i] = (double)(i % 977); pageable[i] = pin[i]; }
    CK(hipMalloc((void**)&dev_full, full));
    CK(hipMalloc((void**)&dev_packed, packed));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto timeit = [&](const char* name, size_t bytes_moved, auto fn) -> int {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            if (fn() != 0) return 1;
            CK(hipStreamSynchronize(st));
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2)
                printf("%-62s %8.2f ms  %7.1f GB/s of payload  %8.0f frames/s\n", name, dt * 1e3, bytes_moved / dt / 1e9, F / dt);
        }
        return 0;
    };
    printf("%d frames of %d waters, %d atoms per water: %.1f MB interleaved, %.1f MB of O rows\n", F, n, lw, full / 1e6, packed / 1e6);
    timeit("linear hipMemcpyAsync, pinned, whole frames", full, [&] { CK(hipMemcpyAsync(dev_full, pin, full, hipMemcpyHostToDevice, st)); return 0; });
    timeit("linear hipMemcpyAsync, pageable, whole frames", full, [&] { CK(hipMemcpyAsync(dev_full, pageable.data(), full, hipMemcpyHostToDevice, st)); return 0; });
    timeit("hipMemcpy2DAsync 24-B rows, pinned", packed, [&] {
        CK(hipMemcpy2DAsync(dev_packed, 24, pin, sizeof(double) * stride, 24, rows, hipMemcpyHostToDevice, st)); return 0; });
    timeit("hipMemcpy2DAsync 24-B rows, pageable", packed, [&] {
        CK(hipMemcpy2DAsync(dev_packed, 24, pageable.data(), sizeof(double) * stride, 24, rows, hipMemcpyHostToDevice, st)); return 0; });
    for (int blocks : {256, 1024, 4096}) {
        char name[128];
        snprintf(name, sizeof name, "gather kernel from pinned host memory, %d blocks x 256", blocks);
        timeit(name, packed, [&] { hipLaunchKernelGGL(k_gather_rows, dim3(blocks), dim3(256), 0, st, pin, stride, rows, dev_packed); return 0; });
        snprintf(name, sizeof name, "line-copy kernel from pinned host memory, %d blocks x 256", blocks);
        timeit(name, full, [&] { hipLaunchKernelGGL(k_copy_lines, dim3(blocks), dim3(256), 0, st, (const double2*)pin, (int64_t)(full / 16), (double2*)dev_full); return 0; });
    }
    // host-side packing of the O rows (what a CPU gather in front of a linear copy would cost), 1 thread
    {
        double* pin2 = nullptr;
        CK(hipHostMalloc((void**)&pin2, packed, hipHostMallocDefault));
        auto t0 = std::chrono::steady_clock::now();
        for (int64_t r = 0; r < rows; ++r) memcpy(pin2 + 3 * r, pageable.data() + r * stride, 24);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%-62s %8.2f ms  %7.1f GB/s of payload  %8.0f frames/s\n", "CPU pack of O rows into pinned memory, 1 thread", dt * 1e3, packed / dt / 1e9, F / dt);
        (void)hipHostFree(pin2);
    }
    return 0;
}
