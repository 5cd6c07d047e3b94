// Does a read-only stream get faster when it draws from several memory CLASSES at once?  (the 288 GB fall
// into three classes in runs of 4-64 GiB of the allocation order; a read stream with a write stream in the SAME class is 13-17 %
// slower.)  Holds NB blocks of 4 GiB, sorts them into classes with a read-A-write-B probe, then times one read kernel over
// 12 GiB taken (a) from three blocks of one class, (b) from one block of each class -- workgroups alternate between the
// blocks, so that at any moment the chip reads all three.
// build: hipcc --offload-arch=gfx950 -O3 class_read.hip -o class_read      usage: class_read [blocks=24]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <vector>
typedef unsigned int u32;

struct Bufs {
    const uint4* p[4];
};

// chunk c (256 KiB) of the virtual stream comes from buffer c % nb, at chunk index c / nb
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read_multi(const Bufs b, int nb, long nchunks, u32* out) {
    u32 acc = 0;
    constexpr long CH16 = (256L << 10) / 16;                                   // uint4 per chunk
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint4* p = b.p[c % nb] + (c / nb) * CH16;
        for (long i = threadIdx.x; i < CH16; i += 256 * UNROLL) {
            uint4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = p[i + 256L * u];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// the classifier: read all of A, write into B
__global__ __launch_bounds__(256) void k_read_write(const uint4* __restrict__ a, uint4* __restrict__ w, long n16, int do_write, u32* out) {
    u32 acc = 0;
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = (long)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i + 256L * u < n16) ? a[i + 256L * u] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        // 16-byte pieces scattered over 1/24 of the addresses, many lanes per piece: PARTIAL-line stores.  They are what tells the classes apart at
        // this size (1.03 in another class, 1.10 in the read block's); whole 4 KiB runs of the same amount (180 MB per 4 GiB read) stay in
        // the 256 MB memory-side cache for the length of the launch and cost 1.03-1.05 wherever they go
        if (do_write && (i / 256) % 6 == 0) w[i / 24] = make_uint4(acc, 1, 2, 3);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

static float timed(hipEvent_t e0, hipEvent_t e1, int reps, const std::function<void()>& f) {
    float best = 1e9;
    for (int it = 0; it < reps + 1; ++it) {
        float ms;
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const int NB = argc > 1 ? atoi(argv[1]) : 24;
    const long BLK = 4L << 30, n16 = BLK / 16;
    std::vector<uint4*> blk;
    for (int i = 0; i < NB; ++i) {
        uint4* p = nullptr;
        if (hipMalloc(&p, BLK) != hipSuccess) break;
        hipMemset(p, 1 + i, BLK);
        blk.push_back(p);
    }
    const int nb = (int)blk.size();
    u32* out;
    hipMalloc(&out, 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * 8;
    printf("%d blocks of 4 GiB held\n", nb);

    // ---- classes relative to a reference block: ratio (read A + write B) / (read A)
    auto ratio_row = [&](int a, std::vector<float>& r) {
        const float base = timed(e0, e1, 3, [&] { hipLaunchKernelGGL(k_read_write, dim3(grid), dim3(256), 0, 0, blk[a], blk[a], n16, 0, out); });
        r.assign(nb, 0.f);
        for (int j = 0; j < nb; ++j) {
            if (j == a) continue;
            r[j] = timed(e0, e1, 3, [&] { hipLaunchKernelGGL(k_read_write, dim3(grid), dim3(256), 0, 0, blk[a], blk[j], n16, 1, out); }) / base;
        }
        return base;
    };
    std::vector<float> r0, r1;
    const float base0 = ratio_row(0, r0);
    printf("read block 0 alone %.3f ms = %.0f GB/s; ratio with the write stream in block j:\n  ", base0, BLK / base0 / 1e6);
    float lo = 1e9, hi = 0;
    for (int j = 1; j < nb; ++j) {
        printf("%.3f ", r0[j]);
        lo = r0[j] < lo ? r0[j] : lo;
        hi = r0[j] > hi ? r0[j] : hi;
    }
    printf("\n");
    const float cut = 0.5f * (lo + hi);
    std::vector<int> cls(nb, -1);
    cls[0] = 0;
    int other = -1;
    for (int j = 1; j < nb; ++j) {
        if (r0[j] > cut) cls[j] = 0;                                              // slow with block 0: its class
        else if (other < 0) other = j;
    }
    if (other < 0 || hi < 1.05f * lo) {
        printf("no class signal (ratios %.3f .. %.3f)\n", lo, hi);
        return 0;
    }
    ratio_row(other, r1);
    cls[other] = 1;
    printf("relative to block %d:\n  ", other);
    for (int j = 0; j < nb; ++j) {
        printf("%.3f ", r1[j]);
        if (cls[j] < 0) cls[j] = r1[j] > cut ? 1 : 2;
    }
    printf("\nclasses: ");
    std::vector<std::vector<int>> of(3);
    for (int j = 0; j < nb; ++j) {
        printf("%d", cls[j]);
        of[cls[j]].push_back(j);
    }
    printf("   (%zu / %zu / %zu blocks)\n", of[0].size(), of[1].size(), of[2].size());

    // ---- the read kernel over 12 GiB: three blocks of one class / one block of each class / and 1 and 2 blocks for scale
    auto run = [&](const char* what, std::vector<int> ids) {
        Bufs b{};
        for (size_t k = 0; k < ids.size(); ++k) b.p[k] = blk[ids[k]];
        const int n = (int)ids.size();
        const long nchunks = (long)n * (BLK / (256L << 10));
        for (int bpc : {4, 8, 16}) {
            const float ms = timed(e0, e1, 5, [&] { hipLaunchKernelGGL(k_read_multi<4>, dim3(256 * bpc), dim3(256), 0, 0, b, n, nchunks, out); });
            printf("%-44s blocks", what);
            for (int id : ids) printf(" %d", id);
            printf("  %2d wg/CU: %.3f ms = %.0f GB/s\n", bpc, ms, (double)n * BLK / ms / 1e6);
        }
    };
    for (int c = 0; c < 3; ++c)
        if (of[c].size() >= 3) {
            char s[64];
            snprintf(s, sizeof s, "three blocks of class %d", c);
            run(s, {of[c][0], of[c][1], of[c][2]});
        }
    if (!of[0].empty() && !of[1].empty() && !of[2].empty()) {
        run("one block of each class", {of[0][0], of[1][0], of[2][0]});
        run("one block of each class (others)", {of[0].back(), of[1].back(), of[2].back()});
    }
    if (!of[0].empty() && !of[1].empty()) run("two classes (0, 1)", {of[0][0], of[1][0]});
    if (of[0].size() >= 2) run("two blocks of class 0", {of[0][0], of[0][1]});
    return 0;
}
