// What the stream operations between two short launches cost the MAIN stream: N steps of two ~10 us kernels A, B
// on one stream, timed wall-clock around a stream sync, in six forms:
//   plain     A, B back to back
//   record    hipEventRecord (disable-timing event) on the main stream between A and B
//   forkjoin  A; record -> a second stream waits, runs a ~5 us kernel S, records; B on the main stream meanwhile; the main stream waits
//             for S before the next step (the route the Gram inverse of block pivoting took until round 6: S hidden behind B)
//   rider     S's work done by one more workgroup of B's launch (what replaced it)
//   memcpy    a 64-byte device -> pinned-host hipMemcpyAsync between A and B (the old progress check's read-back)
//   serial    A, S, B on the one stream (the inversion in stream order)
// usage: mb_event_hop [steps, default 2000]   (profiles/r06_event_hop_cost.txt)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

// ~`spins` dependent multiply-adds per thread; workgroup `extra_wg` (if >= 0) does `extra_spins` instead (the rider)
__global__ void busy_kernel(double* out, int spins, int extra_wg, int extra_spins)
{
    const int n = ((int)blockIdx.x == extra_wg) ? extra_spins : spins;
    double a = threadIdx.x * 1e-9, b = 1.0000001;
    for (int i = 0; i < n; ++i) a = a * b + 1e-12;
    if (a == 12345.678) out[0] = a;      // never true: keeps the loop
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 2000;
    double* d = nullptr;
    double* pin = nullptr;
    CK(hipMalloc(&d, 4096));
    CK(hipHostMalloc((void**)&pin, 4096));
    hipStream_t st, side;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t ev, ev2;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    const int main_spins = 600, side_spins = 300, grid = 256;        // ~10 us and ~5 us
    auto chain = [&](int form) -> double {
        for (int rep = 0; rep < 2; ++rep) {                          // the first pass warms up
            (void)hipStreamSynchronize(st);
            (void)hipStreamSynchronize(side);
            const double t0 = now_us();
            for (int i = 0; i < N; ++i) {
                busy_kernel<<<grid, 256, 0, st>>>(d, main_spins, -1, 0);                                  // A
                if (form == 1) (void)hipEventRecord(ev, st);
                if (form == 2) {
                    (void)hipEventRecord(ev, st);
                    (void)hipStreamWaitEvent(side, ev, 0);
                    busy_kernel<<<1, 256, 0, side>>>(d + 8, side_spins, -1, 0);                           // S beside B
                    (void)hipEventRecord(ev2, side);
                }
                if (form == 4) (void)hipMemcpyAsync(pin, d, 64, hipMemcpyDeviceToHost, st);
                if (form == 5) busy_kernel<<<1, 256, 0, st>>>(d + 8, side_spins, -1, 0);                  // S in stream order
                if (form == 3) busy_kernel<<<grid + 1, 256, 0, st>>>(d, main_spins, grid, side_spins);    // B with S as a rider
                else busy_kernel<<<grid, 256, 0, st>>>(d, main_spins, -1, 0);                             // B
                if (form == 2) (void)hipStreamWaitEvent(st, ev2, 0);
            }
            (void)hipStreamSynchronize(st);
            (void)hipStreamSynchronize(side);
            const double t1 = now_us();
            if (rep == 1) return (t1 - t0) / N;
        }
        return 0.0;
    };
    const char* name[6] = {"plain", "record", "forkjoin", "rider", "memcpy", "serial"};
    double t[6];
    for (int f = 0; f < 6; ++f) t[f] = chain(f);
    for (int f = 0; f < 6; ++f)
        printf("%-9s %7.2f us per step of two kernels  (%+6.2f against plain)\n", name[f], t[f], t[f] - t[0]);
    return 0;
}
