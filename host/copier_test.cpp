// copier_test.cpp -- stress test of sn::Copier (avisynth_sangnom2_amd/csrc/sn_copier.h), meant to run under
// ThreadSanitizer: many back-to-back runs of different shapes, every byte checked.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sn_copier.h"

int main(int argc, char** argv)
{
    const int workers = argc > 1 ? atoi(argv[1]) : 3;
    const int rounds = argc > 2 ? atoi(argv[2]) : 300;
    sn::Copier copier(workers);
    unsigned seed = 12345;
    auto rnd = [&] { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int r = 0; r < rounds; ++r) {
        const int planes = 1 + rnd() % 3;
        std::vector<std::vector<uint8_t>> src(planes), dst(planes);
        sn::Copier::Job jobs[3];
        for (int p = 0; p < planes; ++p) {
            const int row = 1 + rnd() % 6000, rows = rnd() % 700;
            const int sp = row + rnd() % 64, dp = (r & 1) ? row : row + rnd() % 64;
            src[p].resize((size_t)sp * rows + 1);
            dst[p].assign((size_t)dp * rows + 1, 0xEE);
            for (auto& b : src[p]) b = (uint8_t)rnd();
            jobs[p] = {dst[p].data(), src[p].data(), dp, (r & 1) ? row : sp, row, rows};
            if (r & 1) src[p].resize((size_t)row * rows + 1);  // tight on both sides: the single-memcpy path
        }
        copier.run(jobs, planes);
        for (int p = 0; p < planes; ++p)
            for (int y = 0; y < jobs[p].rows; ++y)
                for (int x = 0; x < jobs[p].dpitch; ++x) {
                    const uint8_t got = dst[p][(size_t)y * jobs[p].dpitch + x];
                    const uint8_t want = x < jobs[p].row_bytes ? src[p][(size_t)y * jobs[p].spitch + x] : 0xEE;
                    if (got != want && !(jobs[p].dpitch == jobs[p].row_bytes)) {
                        printf("round %d plane %d (%d,%d): %d != %d\n", r, p, x, y, got, want);
                        return 1;
                    }
                    if (jobs[p].dpitch == jobs[p].row_bytes && got != src[p][(size_t)y * jobs[p].spitch + x]) {
                        printf("round %d plane %d (%d,%d): %d != %d\n", r, p, x, y, got, src[p][(size_t)y * jobs[p].spitch + x]);
                        return 1;
                    }
                }
    }
    printf("ok %d rounds, %d workers\n", rounds, workers);
    return 0;
}
