// Block-timeline probe of split-KV decode attention (7B shapes): entry / tile-range known / first tile staged / tile loop done / exit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTR1_PROBE -I time-r1_amd/csrc tools/probe_attn.hip -o /tmp/pb/probe_attn && /tmp/pb/probe_attn
#include "../time-r1_amd/csrc/attn_fwd.hip"
#include <algorithm>
#include <vector>

static char g_err[256];
extern "C" void tr1_set_error_(const char* m) { strncpy(g_err, m, 255); }

int main(int argc, char** argv) {
    const int64_t P = 3474, G = 8, C = 200, H = 28, NKV = 4, HD = 128, B = argc > 1 ? atoll(argv[1]) : 2, nsplit = argc > 2 ? atoll(argv[2]) : 28;
    const int64_t step = 100, S = ((P + G * C + 63) / 64) * 64, R = B * G;
    void *q, *k, *vt, *o, *ws;
    hipMalloc(&q, R * H * HD * 2); hipMalloc(&o, R * H * HD * 2);
    hipMalloc(&k, B * S * NKV * HD * 2); hipMalloc(&vt, NKV * HD * B * S * 2);
    hipMemset(q, 0x11, R * H * HD * 2); hipMemset(k, 0x11, B * S * NKV * HD * 2); hipMemset(vt, 0x11, NKV * HD * B * S * 2);
    std::vector<int> pre(R, (int)P), lo(R), hi(R);
    for (int r = 0; r < R; ++r) { lo[r] = (int)(P + (r % G) * C); hi[r] = lo[r] + (int)step; }
    int *dpre, *dlo, *dhi; hipMalloc(&dpre, R * 4); hipMalloc(&dlo, R * 4); hipMalloc(&dhi, R * 4);
    hipMemcpy(dpre, pre.data(), R * 4, hipMemcpyHostToDevice); hipMemcpy(dlo, lo.data(), R * 4, hipMemcpyHostToDevice); hipMemcpy(dhi, hi.data(), R * 4, hipMemcpyHostToDevice);
    const int64_t wsf = B * tr1_attn_fwd_workspace_floats(G, H, NKV, HD, nsplit);
    hipMalloc(&ws, wsf * 4);
    // argv[3] = 1: timelines of a plan_mode 2 launch (tile lists published by one plan_mode 1 launch first)
    const int planned = argc > 3 ? atoi(argv[3]) : 0;
    void* plan; hipMalloc(&plan, tr1_attn_plan_ints(G, H, NKV, B) * 4); hipMemset(plan, 0, tr1_attn_plan_ints(G, H, NKV, B) * 4);
    if (planned) tr1_attn_fwd_planned(q, H * HD, k, NKV * HD, vt, B * S, o, H * HD, nullptr, dpre, dlo, dhi, G, H, NKV, S, HD, 0.088f, nsplit, ws, wsf, B, S, plan, 1, nullptr);
    auto run = [&]() { return tr1_attn_fwd_planned(q, H * HD, k, NKV * HD, vt, B * S, o, H * HD, nullptr, dpre, dlo, dhi, G, H, NKV, S, HD, 0.088f, nsplit, ws, wsf, B, S, plan, planned ? 2 : 0, nullptr); };
    for (int i = 0; i < 10; ++i) if (run()) { printf("error %s\n", g_err); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int i = 0; i < 100; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("B=%lld nsplit=%lld: %.2f us per call (attention + combine, back to back)\n", (long long)B, (long long)nsplit, ms * 10);
    const int64_t nblk = nsplit * NKV * B;
    unsigned long long* probe; hipMalloc(&probe, nblk * 8 * 8); hipMemset(probe, 0, nblk * 8 * 8);
    unsigned long long* nullp = nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(tr1_probe), &probe, sizeof(probe));
    run(); hipDeviceSynchronize();
    hipMemcpyToSymbol(HIP_SYMBOL(tr1_probe), &nullp, sizeof(nullp));
    std::vector<unsigned long long> h(nblk * 8); hipMemcpy(h.data(), probe, nblk * 8 * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull; for (int64_t b = 0; b < nblk; ++b) if (h[b * 8]) t0 = std::min(t0, h[b * 8]);
    auto pct = [](std::vector<double> v, double qq) { std::sort(v.begin(), v.end()); return v[(size_t)(qq * (v.size() - 1))]; };
    const char* names[8] = {"entry", "tile range known", "first tile in LDS", "tile loop done", "exit", "tile0: QK issued", "tile0: softmax done", "tile0: PV issued"};
    for (int s : {0, 1, 2, 5, 6, 7, 3, 4}) {
        std::vector<double> v; for (int64_t b = 0; b < nblk; ++b) if (h[b * 8 + s]) v.push_back((h[b * 8 + s] - t0) * 0.01);
        if (v.empty()) continue;
        printf("%-18s: p0 %.2f p50 %.2f p90 %.2f p100 %.2f   (%zu blocks)\n", names[s], pct(v, 0), pct(v, .5), pct(v, .9), pct(v, 1), v.size());
    }
    return 0;
}
