// host_register.hip — what does zero-copy ingest cost?  (krep.c:2630-2726 mmaps the file with
// MAP_POPULATE; the backend memcpy's every byte into a pinned ring before the DMA.)  Measures, for a malloc'd buffer and for a
// read-only private mapping of a /dev/shm file: hipHostRegister time, H2D rate from the registered range, unregister time —
// against the pinned-ring rate (one memcpy thread) and a plain pageable hipMemcpy.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hr tools/ubench/host_register.hip && /tmp/hr [GiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); } } while (0)
int main(int argc, char **argv)
{
    const size_t n = (size_t)((argc > 1 ? atof(argv[1]) : 2.0) * (1 << 30));
    void *d;
    CK(hipMalloc(&d, n));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    // (a) malloc'd, touched
    char *h = (char *)aligned_alloc(4096, n);
    memset(h, 1, n);
    double t0 = now();
    CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
    printf("pageable hipMemcpy            : %7.1f ms  %6.1f GB/s\n", (now() - t0) * 1e3, n / (now() - t0) / 1e9);
    for (int rep = 0; rep < 2; ++rep)
    {
        t0 = now();
        hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault);
        double t1 = now();
        printf("hipHostRegister(malloc)  rep %d: %7.1f ms (%s)  %6.1f GB/s\n", rep, (t1 - t0) * 1e3, hipGetErrorString(e), n / (t1 - t0) / 1e9);
        if (e == hipSuccess)
        {
            t0 = now();
            CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            t1 = now();
            printf("  H2D from registered         : %7.1f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
            t0 = now();
            CK(hipHostUnregister(h));
            printf("  hipHostUnregister           : %7.1f ms\n", (now() - t0) * 1e3);
        }
        else
            (void)hipGetLastError();
    }
    // (b) a read-only private file mapping, as krep's mmap
    const char *path = "/dev/shm/krep_hr.bin";
    int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd >= 0 && ftruncate(fd, (off_t)n) == 0)
    {
        for (size_t o = 0; o < n; o += 1 << 30)
            if (pwrite(fd, h + o, (n - o) < ((size_t)1 << 30) ? n - o : (size_t)1 << 30, (off_t)o) < 0) break;
        t0 = now();
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        printf("mmap(PROT_READ, MAP_POPULATE)  : %7.1f ms\n", (now() - t0) * 1e3);
        if (m != MAP_FAILED)
        {
            for (unsigned flags : {(unsigned)hipHostRegisterDefault, (unsigned)hipHostRegisterReadOnly})
            {
                t0 = now();
                hipError_t e = hipHostRegister(m, n, flags);
                double t1 = now();
                printf("hipHostRegister(mmap, flags %u) : %7.1f ms (%s)\n", flags, (t1 - t0) * 1e3, hipGetErrorString(e));
                if (e == hipSuccess)
                {
                    t0 = now();
                    CK(hipMemcpyAsync(d, m, n, hipMemcpyHostToDevice, st));
                    CK(hipStreamSynchronize(st));
                    t1 = now();
                    printf("  H2D from registered mapping : %7.1f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
                    CK(hipHostUnregister(m));
                    break;
                }
                (void)hipGetLastError();
            }
            t0 = now();
            CK(hipMemcpy(d, m, n, hipMemcpyHostToDevice));
            printf("pageable hipMemcpy from mapping: %7.1f ms  %6.1f GB/s\n", (now() - t0) * 1e3, n / (now() - t0) / 1e9);
            munmap(m, n);
        }
        close(fd);
        unlink(path);
    }
    // (c) the memcpy side of the ring: one thread, pageable -> pinned
    void *pin;
    CK(hipHostMalloc(&pin, 64 << 20));
    t0 = now();
    for (size_t o = 0; o + (64 << 20) <= n; o += 64 << 20)
        memcpy(pin, h + o, 64 << 20);
    printf("memcpy pageable -> pinned, 1 thread: %6.1f GB/s\n", (n / (64 << 20)) * (double)(64 << 20) / (now() - t0) / 1e9);
    return 0;
}
