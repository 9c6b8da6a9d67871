// Shared host-side helpers for libfsg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fsg_hip.h"

#define FSG_WAVE 64

void fsg_set_error(const char *fmt, ...);

#define FSG_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            fsg_set_error(__VA_ARGS__);   \
            return FSG_ERR_ARG;           \
        }                                 \
    } while (0)

#define FSG_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            fsg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return FSG_ERR_HIP;                                                       \
        }                                                                             \
    } while (0)

static inline int fsg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
