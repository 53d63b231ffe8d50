// Is the V^T cache layout (128 feature rows x 64 slots = 128-byte pieces at a stride of the whole cache row) a bandwidth problem for the
// decode attention, compared with K's layout (64 slot rows x 256 bytes at a 1 KiB stride)?  Pure-load kernel: `nblk` blocks of 256 threads,
// each reading `tiles` 16 KiB tiles with one of the two address patterns (16 bytes per lane per load, 4 loads per thread and tile, PF tiles
// in flight), checksum so nothing is optimised away.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_kv_pattern.hip -o /tmp/pkv && /tmp/pkv
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int PAT>
__global__ __launch_bounds__(256, 2) void rd(const char* base, size_t ld_bytes, int tiles, int nsplit, unsigned* sink) {
    // block = (group = blockIdx.y, split = blockIdx.x); tile i of the block = tile (split + i * nsplit) of the group's region
    unsigned acc = 0;
    const int t = threadIdx.x;
    for (int i = 0; i < tiles; ++i) {
        const size_t tile = ((size_t)blockIdx.x + (size_t)i * nsplit) % 128;      // 8192 slots per prompt
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = t + j * 256;                       // 1024 16-byte chunks per 16 KiB tile
            const char* p;
            if (PAT == 0) {       // K: 64 rows x 256 B, row stride ld_bytes (1 KiB), tile = 64 consecutive rows, head offset blockIdx.y * 256
                const int row = idx >> 4, c = idx & 15;
                p = base + (tile * 64 + row) * ld_bytes + (size_t)(blockIdx.y & 3) * 256 + c * 16 + (size_t)(blockIdx.y >> 2) * (ld_bytes * 8192);
            } else {              // V^T: 128 rows x 128 B, row stride ld_bytes (the whole cache row), tile = 64 consecutive slots
                const int row = idx >> 3, c = idx & 7;
                p = base + ((size_t)(blockIdx.y & 3) * 128 + row) * ld_bytes + tile * 128 + c * 16 + (size_t)(blockIdx.y >> 2) * 8192 * 2;
            }
            v[j] = *reinterpret_cast<const u32x4*>(p);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t slots = 2 * 8192;                 // two prompts' caches, 8192 slots each
    const int NREG = 48;                           // distinct cache copies ("layers") cycled through, 16 MiB each: far more than L2 + the memory-side cache
    const size_t kreg = slots * 1024, vreg = 512 * (slots * 2);
    const size_t kbytes = NREG * kreg + (1 << 20), vbytes = NREG * vreg + (1 << 20);
    char *k, *v; unsigned* sink;
    (void)hipMalloc(&k, kbytes); (void)hipMalloc(&v, vbytes); (void)hipMalloc(&sink, 4);
    (void)hipMemset(k, 1, kbytes); (void)hipMemset(v, 1, vbytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int tiles : {2, 4, 8, 16}) {
        const int nsplit = 28;
        for (int pat = 0; pat < 2; ++pat) {
            const size_t ld = pat == 0 ? 1024 : slots * 2;
            int reg = 0;
            auto launch = [&]() {
                reg = (reg + 1) % NREG;
                if (pat == 0) hipLaunchKernelGGL(rd<0>, dim3(nsplit, 8), dim3(256), 0, 0, k + reg * kreg, ld, tiles, nsplit, sink);
                else hipLaunchKernelGGL(rd<1>, dim3(nsplit, 8), dim3(256), 0, 0, v + reg * vreg, ld, tiles, nsplit, sink);
            };
            launch(); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            for (int r = 0; r < 50; ++r) launch();
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double bytes = 224.0 * tiles * 16384;
            printf("%s pattern, %d tiles per block: %.2f us per launch, %.0f GB/s\n", pat == 0 ? "K  (64 x 256 B rows, 1 KiB stride)   " : "V^T (128 x 128 B rows, 32 KiB stride)", tiles,
                   ms * 1000 / 50, bytes / (ms / 50 * 1e-3) / 1e9);
        }
    }
    return 0;
}
