// Which HIP streams get in each other's way? (round 4, behind the stream layout of swsem_create)
//   hipcc --offload-arch=gfx950 -O3 -o profiles/queue_pipes profiles/queue_pipes.hip && GPU_MAX_HW_QUEUES=8 profiles/queue_pipes
// A kernel whose grid is larger than what the device holds at once keeps being dispatched for as long as it runs; a kernel
// launched meanwhile on another stream starts at once if its stream's hardware queue is served by another dispatch pipe, and
// only behind the first kernel's last workgroup if the two queues share one. The matrix says which: N streams are made and
// first used in order (plus classes: n normal, h high, l low priority), then for every ordered pair (A, B) a long-dispatch
// "hog" goes to A, a one-workgroup kernel to B, and B's completion time is compared with the hog's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_hog(unsigned long long ticks, unsigned long long *sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks == 0x7fffffffffffull) *sink = t0;
}
__global__ void k_tiny(unsigned long long *sink) { if (threadIdx.x == 1000) *sink = 1; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const char *classes = argc > 1 ? argv[1] : "nnnnnnnnhhll";
    const int N = (int) strlen(classes);
    int least = 0, greatest = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    unsigned long long *sink;
    CHECK(hipMalloc(&sink, 8));
    std::vector<hipStream_t> st(N);
    for (int i = 0; i < N; i++) {
        const int pr = classes[i] == 'h' ? greatest : (classes[i] == 'l' ? least : (least + greatest) / 2);
        CHECK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, pr));
        k_tiny<<<1, 64, 0, st[i]>>>(sink);                                    // first use: the stream gets its queue now
        CHECK(hipStreamSynchronize(st[i]));
    }
    // the hog: 4 generations of workgroups of 25 us each
    hipDeviceProp_t pr;
    CHECK(hipGetDeviceProperties(&pr, 0));
    const int wgs = pr.multiProcessorCount * 8 * 4;
    printf("{\"classes\": \"%s\", \"priority_range\": [%d, %d], \"hog_workgroups\": %d, \"rows_hog_on\": [\n", classes, least, greatest, wgs);
    for (int a = 0; a < N; a++) {
        printf("  [");
        for (int b = 0; b < N; b++) {
            if (a == b) { printf("null%s", b + 1 < N ? ", " : ""); continue; }
            double best = 1e30, hogBest = 0;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipDeviceSynchronize());
                const double t0 = now_us();
                k_hog<<<wgs, 256, 0, st[a]>>>(2500, sink);
                k_tiny<<<1, 64, 0, st[b]>>>(sink);
                CHECK(hipStreamSynchronize(st[b]));
                const double t1 = now_us();
                CHECK(hipStreamSynchronize(st[a]));
                const double t2 = now_us();
                if (t1 - t0 < best) { best = t1 - t0; hogBest = t2 - t0; }
            }
            printf("%.0f%s", best, b + 1 < N ? ", " : "");
            (void) hogBest;
        }
        printf("]%s\n", a + 1 < N ? "," : "");
    }
    CHECK(hipDeviceSynchronize());
    double t0 = now_us();
    k_hog<<<wgs, 256, 0, st[0]>>>(2500, sink);
    CHECK(hipStreamSynchronize(st[0]));
    printf("], \"hog_alone_us\": %.0f}\n", now_us() - t0);
    return 0;
}
