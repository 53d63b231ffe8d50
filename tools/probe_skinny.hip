// Block-timeline probe of the decode GEMM: when does each block start / finish its k loop / exit?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTR1_PROBE -I time-r1_amd/csrc tools/probe_skinny.hip -o /tmp/probe && /tmp/probe
#include "../time-r1_amd/csrc/gemm.hip"
#include <algorithm>
#include <vector>

static char g_err[256];
extern "C" void tr1_set_error_(const char* m) { strncpy(g_err, m, 255); }

int main(int argc, char** argv) {
    const int64_t M = 16, N = argc > 1 ? atoll(argv[1]) : 37888, K = argc > 2 ? atoll(argv[2]) : 3584;
    const int ncopies = 6;
    std::vector<void*> Ws(ncopies);
    for (auto& w : Ws) { hipMalloc(&w, N * K * 2); hipMemset(w, 0x11, N * K * 2); }
    void *x, *c; hipMalloc(&x, M * K * 2); hipMalloc(&c, M * N * 2); hipMemset(x, 0x11, M * K * 2);
    const int64_t nblk_max = N / 16 + 8;
    unsigned long long* probe; hipMalloc(&probe, nblk_max * 4 * 8);
    unsigned long long* nullp = nullptr;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 20; ++it) tr1_gemm_nt_bf16(x, Ws[it % ncopies], c, nullptr, nullptr, M, N, K, K, K, N, N, 0, 0, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 60; ++it) tr1_gemm_nt_bf16(x, Ws[it % ncopies], c, nullptr, nullptr, M, N, K, K, K, N, N, 0, 0, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("M=%lld N=%lld K=%lld: %.2f us per launch (back to back), %.2f TB/s\n", (long long)M, (long long)N, (long long)K, ms * 1000 / 60, N * K * 2.0 / (ms / 60 * 1e-3) / 1e12);
    hipMemset(probe, 0, nblk_max * 4 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(tr1_probe), &probe, sizeof(probe));
    tr1_gemm_nt_bf16(x, Ws[3], c, nullptr, nullptr, M, N, K, K, K, N, N, 0, 0, nullptr);
    hipDeviceSynchronize();
    hipMemcpyToSymbol(HIP_SYMBOL(tr1_probe), &nullp, sizeof(nullp));
    std::vector<unsigned long long> h(nblk_max * 4);
    hipMemcpy(h.data(), probe, nblk_max * 4 * 8, hipMemcpyDeviceToHost);
    std::vector<double> st, lp, en;
    unsigned long long t0 = ~0ull;
    int nb = 0;
    for (int64_t b = 0; b < nblk_max; ++b) if (h[b * 4]) { t0 = std::min(t0, h[b * 4]); ++nb; }
    for (int64_t b = 0; b < nblk_max; ++b) if (h[b * 4]) { st.push_back((h[b * 4] - t0) * 0.01); lp.push_back((h[b * 4 + 1] - t0) * 0.01); en.push_back((h[b * 4 + 2] - t0) * 0.01); }
    auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    printf("blocks %d  (times in us since the first block entered; 100 MHz counter)\n", nb);
    printf("block entry   : p0 %.2f p25 %.2f p50 %.2f p75 %.2f p90 %.2f p100 %.2f\n", pct(st, 0), pct(st, .25), pct(st, .5), pct(st, .75), pct(st, .9), pct(st, 1));
    printf("k-loop done   : p0 %.2f p25 %.2f p50 %.2f p75 %.2f p90 %.2f p100 %.2f\n", pct(lp, 0), pct(lp, .25), pct(lp, .5), pct(lp, .75), pct(lp, .9), pct(lp, 1));
    printf("block exit    : p0 %.2f p50 %.2f p100 %.2f\n", pct(en, 0), pct(en, .5), pct(en, 1));
    std::vector<double> dur; for (size_t i = 0; i < st.size(); ++i) dur.push_back(lp[i] - st[i]);
    printf("per-block k-loop time: p0 %.2f p25 %.2f p50 %.2f p75 %.2f p100 %.2f\n", pct(dur, 0), pct(dur, .25), pct(dur, .5), pct(dur, .75), pct(dur, 1));
    // histogram of concurrency: number of blocks inside their k loop at each microsecond
    const int T = (int)pct(en, 1) + 2;
    printf("active blocks per us:");
    for (int t = 0; t < T; ++t) { int a = 0; for (size_t i = 0; i < st.size(); ++i) if (st[i] <= t + 0.5 && lp[i] > t + 0.5) ++a; printf(" %d", a); }
    printf("\n");
    return 0;
}
