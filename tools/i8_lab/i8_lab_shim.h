// The handful of lt_internal.h helpers i8_split.hip uses, for the standalone lab build.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define LT_OK 0
#define LT_HIP(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)
#define LT_CHECK_LAUNCH() LT_HIP(hipGetLastError())
static inline int lt_round_up(int v, int m) { return (v + m - 1) / m * m; }
