// g++ -O1 -g -fsanitize=address,undefined -o /tmp/inflate_sanitized tools/inflate_sanitized.cpp -lz -lpthread; /tmp/inflate_sanitized files.gz ...
// mf_inflate.h under ASan + UBSan on valid and damaged gzip / BGZF files, one thread and the several-thread form forced with 16 KB pieces
// (round 5: 416 files of four kinds of content, every level, bit flips / truncations / overwritten bytes / trailing garbage: no report)
#include <cstdio>
#include <algorithm>
#include "../metafast_amd/csrc/mf_inflate.h"
int main(int argc, char **argv) {
    int okc = 0, bad = 0;
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb"); if (!f) continue; fseek(f, 0, SEEK_END); size_t n = ftell(f); fseek(f, 0, SEEK_SET);
        uint8_t *in = (uint8_t *)malloc(n + 64); if (fread(in, 1, n, f) != n) return 1; memset(in + n, 0, 64); fclose(f);
        for (int mode = 0; mode < 2; mode++) {
            char *o = nullptr; size_t m = 0; bool par = false;
            bool ok = mode ? mfz::gunzip(in, n, 4, &o, &m, 0, (size_t)16 << 10, &par) : mfz::gunzip(in, n, 4, &o, &m);
            if (ok) { okc++; free(o); } else bad++;
        }
        free(in);
    }
    printf("accepted %d refused %d\n", okc, bad);
}
