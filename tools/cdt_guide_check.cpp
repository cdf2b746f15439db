// dev utility (host only): the guide layout the engine picks for a (tail, sigma) table and its probe count S.
// build: hipcc -O2 -I libsafecrypto_amd/csrc tools/cdt_guide_check.cpp -L libsafecrypto_amd/lib -lsc_hip -Wl,-rpath,$PWD/libsafecrypto_amd/lib -o /tmp/cdt_guide_check
#include "engine_internal.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv)
{
    const float tail = argc > 1 ? (float)atof(argv[1]) : 13.42f;
    for (int a = 2; a < argc; a++) {
        const float sigma = (float)atof(argv[a]);
        const int size = sc_hip_cdt_table_size(tail, sigma);
        std::vector<uint64_t> t((size_t)size);
        if (sc_hip_cdt_build_table(tail, sigma, t.data(), size)) { printf("sigma %g: table failed\n", sigma); continue; }
        CdtGuide g; std::vector<uint16_t> gd; std::vector<uint64_t> pt;
        sc::cdt_build_guide(t, &g, &gd, &pt);
        printf("sigma %8.3f size %5d: LB %u K %u MB %u bins %u S %u jump %u (LDS %zu B)\n", sigma, size, 32 - g.lb_shift, g.K, g.MB, g.total, g.S, g.jump,
               (size_t)(size + CDT_PAD) * 8 + gd.size() * 2);
    }
    return 0;
}
