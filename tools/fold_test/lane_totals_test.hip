// Checks lt_lanes.hip.h against group_sum<64> bit for bit on the GPU:  hipcc --offload-arch=gfx950 -ffp-contract=off
//   -I include -I linkteller_amd/csrc tools/fold_test/lane_totals_test.hip -o /tmp/lane_totals_test && /tmp/lane_totals_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "lt_rows.hip.h"
#include "lt_lanes.hip.h"

template <int V>
__global__ void k_test(const float *in, float *ref, float *got, int *owner) {
    const int lane = threadIdx.x;
    float v[V];
    for (int i = 0; i < V; ++i) v[i] = in[i * 64 + lane];
    for (int i = 0; i < V; ++i) {
        const float t = group_sum<64>(v[i]);
        if (lane == 0) ref[i] = t;
    }
    const float z = lane_totals<V>(v, lane);
    got[lane] = z;
    owner[lane] = lane_totals_owner<V>(lane);
}

template <int V>
static int run() {
    std::vector<float> h(V * 64);
    unsigned s = 12345u + V;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x = (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 20); }
    float *in, *ref, *got; int *own;
    hipMalloc(&in, h.size() * 4); hipMalloc(&ref, V * 4); hipMalloc(&got, 64 * 4); hipMalloc(&own, 64 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test<V>, dim3(1), dim3(64), 0, 0, in, ref, got, own);
    std::vector<float> r(V), g(64); std::vector<int> o(64);
    hipMemcpy(r.data(), ref, V * 4, hipMemcpyDeviceToHost);
    hipMemcpy(g.data(), got, 64 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(o.data(), own, 64 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        if (std::memcmp(&g[l], &r[o[l]], 4) != 0) {
            if (bad < 4) std::printf("  V=%d lane %d owner %d: got %.9g want %.9g\n", V, l, o[l], g[l], r[o[l]]);
            ++bad;
        }
    std::printf("V=%d: %s (%d lanes differ)\n", V, bad ? "FAIL" : "ok", bad);
    return bad;
}

int main() {
    int bad = 0;
    bad += run<1>(); bad += run<2>(); bad += run<4>(); bad += run<8>();
    bad += run<16>(); bad += run<32>(); bad += run<64>();
    return bad ? 1 : 0;
}
