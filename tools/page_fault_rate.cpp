#include <sys/mman.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>
int main() {
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const size_t N = (size_t)1 << 30;
    for (int mode = 0; mode < 4; mode++) {
        char *p = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | (mode == 2 ? MAP_POPULATE : 0), -1, 0);
        if (mode == 1) madvise(p, N, MADV_HUGEPAGE);
        double t0 = now();
        const int T = mode == 3 ? 8 : 1;
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([=]() { for (size_t i = N / T * t; i < N / T * (t + 1); i += 4096) p[i] = 1; });
        for (auto &x : th) x.join();
        printf("mode %d (%s): touch 1 GB in %.3f s\n", mode, mode == 0 ? "plain" : mode == 1 ? "MADV_HUGEPAGE" : mode == 2 ? "MAP_POPULATE (at mmap)" : "plain, 8 threads", now() - t0);
        munmap(p, N);
    }
}
