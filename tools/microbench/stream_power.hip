// Board power of an HBM-bound row kernel by how many waves wait on the memory: y[r, :] = 2 x[r, :] over [113 440, 1 024] bf16 (the RMSNorm forward's traffic, 232 MB in + 232 MB out),
// one wave per row, grid-stride, GRID workgroups of 256 threads.  Does a smaller grid hold the same rate at less power (the step runs at the board's cap: watts saved here are clock
// for the GEMMs around it)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_power tools/microbench/stream_power.hip && /tmp/stream_power GRID SECONDS [ROWS_IN_FLIGHT 1|2|4] [ROWS]
// Prints `WINDOW t0 t1 us_per_launch` for tools/power_trace.py (workload name exe=/tmp/stream_power,GRID,SECONDS,VECS).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int ROWS>  // rows per wave in flight
__global__ __launch_bounds__(256) void k(long rows, const u32x4* __restrict__ x, u32x4* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    for (long r = w0 * ROWS; r < rows; r += nw * ROWS) {
        u32x4 v[ROWS][2];
#pragma unroll
        for (int i = 0; i < ROWS; ++i)
            if (r + i < rows) { v[i][0] = x[(r + i) * 128 + lane]; v[i][1] = x[(r + i) * 128 + 64 + lane]; }
#pragma unroll
        for (int i = 0; i < ROWS; ++i)
            if (r + i < rows) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[i][0][e] ^= 0x00800080u; v[i][1][e] ^= 0x00800080u; }
                y[(r + i) * 128 + lane] = v[i][0];
                y[(r + i) * 128 + 64 + lane] = v[i][1];
            }
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 2048;
    const double sec = argc > 2 ? atof(argv[2]) : 4.0;
    const int rows_in_flight = argc > 3 ? atoi(argv[3]) : 1;
    const long rows = argc > 4 ? atol(argv[4]) : 113440;  // 16 384 rows = 32 MB in + 32 MB out: misses the L2s (4 MB each), lives in the Infinity Cache (256 MB)
    u32x4 *x, *y;
    hipMalloc(&x, rows * 2048); hipMalloc(&y, rows * 2048);
    hipMemset(x, 0x3c, rows * 2048);
    auto launch = [&]() {
        if (rows_in_flight == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, rows, x, y);
        else if (rows_in_flight == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, rows, x, y);
        else hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, rows, x, y);
    };
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int i = 0; i < 50; ++i) launch(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int n = (int)(sec * 1e3 / (ms / 50)) + 1;
    const double t0 = now();
    hipEventRecord(e0); for (int i = 0; i < n; ++i) launch(); hipEventRecord(e1); hipDeviceSynchronize();
    const double t1 = now();
    hipEventElapsedTime(&ms, e0, e1);
    printf("grid %d rows-in-flight %d rows %ld: %.1f us per launch, %.2f TB/s\n", grid, rows_in_flight, rows, ms / n * 1e3, 2.0 * rows * 2048 / (ms / n * 1e-3) / 1e12);
    printf("WINDOW %.6f %.6f %.2f\n", t0, t1, ms / n * 1e3);
    return 0;
}
