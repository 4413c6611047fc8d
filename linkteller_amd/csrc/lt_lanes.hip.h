// Multi-value 64-lane reduction for the wide stage-A kernels.
//
// A wave holds V lane-partial values (V = probes x classes, a power of two) and needs their V
// totals.  group_sum<64> would run the xor butterfly 32, 16, 8, 4, 2, 1 once per value, every lane
// ending with every total.  Here every stage of that same butterfly -- same lane pairs, same
// order, fp add commutes, hence the SAME BITS as group_sum<64> -- also halves the number of live
// registers: the two lane halves a stage pairs up keep different values.  After log2(V) stages one
// register is left and lane l owns the total of value  l >> (6 - log2 V)  (V <= 64), so the V
// totals of a wave leave in ONE coalesced store instead of V/4 scattered ones.
//
//   stage 32 / 16 : v_permlane32_swap / v_permlane16_swap + add          (2 VALU per pair)
//   stage  8 /  4 : two bank-masked v_add_f32_dpp row rotations in place  (2 VALU per pair)
//   stage  2 /  1 : two v_cndmask + one quad_perm v_add_f32_dpp           (3 VALU per pair)
//
// 64 values: 129 VALU instead of 64 * 6.  Inline asm throughout: the DPP/permlane hazards (two wait
// states after a VALU write of a source) are the s_nop at the head of each block, and hipcc
// (ROCm 7.2) mis-folds the permlane swap builtins' result pair.
#pragma once

// lanes l and l+32 of (a, b) -> lanes 0-31: a[l] + a[l+32], lanes 32-63: b[l-32] + b[l]
// (v_permlane32_swap exchanges lanes 32-63 of %0 with lanes 0-31 of %1)
__device__ __forceinline__ float fold32(float a, float b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// 16-lane rows r and r^1 of (a, b) -> rows [a0+a1 | b0+b1 | a2+a3 | b2+b3]
// (v_permlane16_swap exchanges the odd rows of %0 with the even rows of %1)
__device__ __forceinline__ float fold16(float a, float b) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// lanes with (l & 8) == 0: a[l] + a[l+8]; the others: b[l-8] + b[l].  bank_mask selects 4-lane banks.
__device__ __forceinline__ float fold8(float a, float b) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc"
        : "+v"(a)
        : "v"(b));
    return a;
}
// lanes with (l & 4) == 0: a[l] + a[l+4]; the others: b[l-4] + b[l].
// row_ror:n gives lane l the value of lane (l - n) mod 16, so "+4" is a rotation by 12.
__device__ __forceinline__ float fold4(float a, float b) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xa"
        : "+v"(a)
        : "v"(b));
    return a;
}
// lanes with (l & 2) == 0: a[l] + a[l^2]; the others: b[l] + b[l^2]
__device__ __forceinline__ float fold2(float a, float b, bool hi) {
    const float own = hi ? b : a, give = hi ? a : b;
    float t;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=v"(t)
        : "v"(give), "v"(own));
    return t;
}
// lanes with (l & 1) == 0: a[l] + a[l^1]; the others: b[l] + b[l^1]
__device__ __forceinline__ float fold1(float a, float b, bool hi) {
    const float own = hi ? b : a, give = hi ? a : b;
    float t;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "=v"(t)
        : "v"(give), "v"(own));
    return t;
}
// stages that no longer have two values to merge: x[l] + x[l^k] in every lane
__device__ __forceinline__ float all8(float x) {
    float t;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(x));
    return t;
}
__device__ __forceinline__ float all4(float x) { return fold4(x, x); }   // x[l] + x[l^4] (two masked rotations)
__device__ __forceinline__ float all2(float x) {
    float t;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(x));
    return t;
}
__device__ __forceinline__ float all1(float x) {
    float t;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(x));
    return t;
}

// V (power of two, <= 64) lane-partial values v[0..V) -> their 64-lane totals: on return lane l holds the
// total of value  l >> (6 - log2 V).  A stage pairs value i with value i + (live / 2), so the lane bit
// of stage k (5, 4, ... ) becomes the value-index bit of the same rank (top bit first).
template <int V>
__device__ __forceinline__ float lane_totals(const float *v, int lane) {
    static_assert(V >= 1 && V <= 64 && (V & (V - 1)) == 0, "V must be a power of two <= 64");
    float w[V];
#pragma unroll
    for (int i = 0; i < V; ++i) w[i] = v[i];
    constexpr int L1 = V >= 2 ? V / 2 : 1;      // live registers after stage 32
    constexpr int L2 = L1 >= 2 ? L1 / 2 : 1;    // ... 16
    constexpr int L3 = L2 >= 2 ? L2 / 2 : 1;    // ... 8
    constexpr int L4 = L3 >= 2 ? L3 / 2 : 1;    // ... 4
    constexpr int L5 = L4 >= 2 ? L4 / 2 : 1;    // ... 2
    if constexpr (V >= 2) {
#pragma unroll
        for (int i = 0; i < L1; ++i) w[i] = fold32(w[i], w[i + L1]);
    } else w[0] = fold32(w[0], w[0]);
    if constexpr (L1 >= 2) {
#pragma unroll
        for (int i = 0; i < L2; ++i) w[i] = fold16(w[i], w[i + L2]);
    } else w[0] = fold16(w[0], w[0]);
    if constexpr (L2 >= 2) {
#pragma unroll
        for (int i = 0; i < L3; ++i) w[i] = fold8(w[i], w[i + L3]);
    } else w[0] = all8(w[0]);
    if constexpr (L3 >= 2) {
#pragma unroll
        for (int i = 0; i < L4; ++i) w[i] = fold4(w[i], w[i + L4]);
    } else w[0] = all4(w[0]);
    if constexpr (L4 >= 2) {
        const bool hi = (lane & 2) != 0;
#pragma unroll
        for (int i = 0; i < L5; ++i) w[i] = fold2(w[i], w[i + L5], hi);
    } else w[0] = all2(w[0]);
    if constexpr (L5 >= 2) w[0] = fold1(w[0], w[1], (lane & 1) != 0);
    else w[0] = all1(w[0]);
    return w[0];
}
template <int V>
__device__ __forceinline__ int lane_totals_owner(int lane) {   // value index whose total lane `lane` holds
    int s = 0;
    for (int t = V; t < 64; t <<= 1) ++s;
    return lane >> s;
}
