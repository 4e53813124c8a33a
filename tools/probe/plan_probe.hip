// k_stage_cut / k_stage_cut_lds alone on a synthetic tile table (hipEvents; both forms must print the same hash).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ignn-builder_amd/csrc -o tools/probe/plan_probe tools/probe/plan_probe.hip
// usage: plan_probe [tiles]
#include "../../gnn-builder_amd/csrc/k_plan.hip"
#include <cstdio>
#include <vector>
#include <random>
using namespace gnnb;
int main(int argc, char **argv)
{
    int B = argc > 1 ? atoi(argv[1]) : 4096;
    std::mt19937 rng(1);
    std::vector<int32_t> tf(B + 1);
    int n = 0;
    for (int i = 0; i < B; i++) {
        tf[i] = n;
        n += 10 + rng() % 32;
    }
    tf[B] = n;
    int32_t *d_tf, *d_cut, *d_scr;
    const int G = 512;
    hipMalloc(&d_tf, (B + 1) * 4);
    hipMalloc(&d_cut, (G + 2) * 4);
    hipMalloc(&d_scr, (size_t)stage_cut_levels(B) * (B + 1) * 4);
    hipMemcpy(d_tf, tf.data(), (B + 1) * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int form = 0; form < 2; form++) {
        for (int it = 0; it < 3; it++) {
            hipEventRecord(e0, 0);
            for (int r = 0; r < 20; r++) {
                if (form == 0)
                    launch_stage_cut(d_tf, B, n, 64, G, 62, d_scr, d_cut, 0);
                else
                    hipLaunchKernelGGL(k_stage_cut, dim3(1), dim3(PL_WG), 0, 0, d_tf, B, n, 64, G, 62, d_scr, stage_cut_levels(B), d_cut);
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("form %d: %.1f us per launch\n", form, ms * 1e3f / 20);
        }
        std::vector<int32_t> cut(G + 2);
        hipMemcpy(cut.data(), d_cut, (G + 2) * 4, hipMemcpyDeviceToHost);
        long long h = 0;
        for (int i = 0; i < G + 2; i++)
            h = h * 31 + cut[i];
        printf("  rows %d, cut[1]=%d cut[G]=%d ok=%d hash %lld\n", n, cut[1], cut[G], cut[G + 1], h);
    }
    return 0;
}
