// What page-locking a round's worth of staging memory costs (the C++ host pins two buffers of a round's files, 160 MB each on
// 28 x 4.6 MB genomes, before the first round): hipHostMalloc, and hipHostRegister on memory the process already owns (touched
// 4 KB pages; 2 MB pages where the system gives them). hipcc --offload-arch=gfx950 -O2 -o profiles/pin_bench profiles/pin_bench.hip
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t n = (argc > 1 ? atol(argv[1]) : 160) << 20;
    hipFree(0);
    void *d = nullptr;
    hipMalloc(&d, n);
    printf("{\"bytes\": %zu", n);
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        void *p = nullptr;
        if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) return 1;
        double t1 = now();
        memset(p, 1, n);
        double t2 = now();
        hipMemcpy(d, p, n, hipMemcpyHostToDevice);
        double t3 = now();
        hipHostFree(p);
        printf(", \"hipHostMalloc_%d\": {\"alloc_ms\": %.1f, \"first_touch_ms\": %.1f, \"h2d_ms\": %.1f, \"free_ms\": %.1f}", rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (now() - t3) * 1e3);
    }
    for (int huge = 0; huge < 2; huge++) {
        double t0 = now();
        void *p = mmap(nullptr, n + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        char *a = (char *) (((uintptr_t) p + (2 << 20) - 1) & ~(uintptr_t) ((2 << 20) - 1));
        if (huge) madvise(a, n, MADV_HUGEPAGE);
        memset(a, 1, n);
        double t1 = now();
        hipError_t e = hipHostRegister(a, n, hipHostRegisterDefault);
        double t2 = now();
        hipMemcpy(d, a, n, hipMemcpyHostToDevice);
        double t3 = now();
        if (e == hipSuccess) hipHostUnregister(a);
        printf(", \"register_%s\": {\"touch_ms\": %.1f, \"register_ms\": %.1f, \"ok\": %d, \"h2d_ms\": %.1f, \"unregister_ms\": %.1f}", huge ? "thp" : "4k", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (int) (e == hipSuccess), (t3 - t2) * 1e3, (now() - t3) * 1e3);
        munmap(p, n + (2 << 20));
    }
    {   // pageable memory straight into hipMemcpy
        char *a = (char *) malloc(n);
        memset(a, 1, n);
        double t0 = now();
        hipMemcpy(d, a, n, hipMemcpyHostToDevice);
        printf(", \"pageable_h2d_ms\": %.1f", (now() - t0) * 1e3);
        free(a);
    }
    printf("}\n");
    return 0;
}
