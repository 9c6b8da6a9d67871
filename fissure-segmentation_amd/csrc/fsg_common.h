// Shared host-side helpers for libfsg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

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

// Dynamic LDS above the 64 KiB every kernel may use is granted per kernel AND per device (hipFuncSetAttribute acts on the
// current device's copy of the function).  One instance per call site -- a function-local static of the launching function /
// template instantiation --, remembering what each device has been granted; safe when several host threads launch.
struct FsgLdsGrant {
    static constexpr int kMaxDevices = 64;
    std::atomic<size_t> bytes[kMaxDevices];
    FsgLdsGrant() {
        for (auto &b : bytes) b.store(64 * 1024, std::memory_order_relaxed);
    }
    // make sure the kernel may be launched with `need` bytes of dynamic LDS on the current device; false if the runtime refuses
    bool raise(const void *kernel, size_t need) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        const bool tracked = dev >= 0 && dev < kMaxDevices;
        if (tracked && need <= bytes[dev].load(std::memory_order_acquire)) return true;
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need) != hipSuccess) return false;
        if (tracked) {     // monotone maximum (another thread may have raised it further in the meantime)
            size_t cur = bytes[dev].load(std::memory_order_relaxed);
            while (cur < need && !bytes[dev].compare_exchange_weak(cur, need, std::memory_order_release)) {
            }
        }
        return true;
    }
};

#ifdef __HIPCC__
// Row gathers with lanes = channels: the row index is wave-uniform (v_readlane of the neighbour list), so the row offset
// belongs in the SCALAR offset of a buffer load and the lane's channel in its vector offset -- no per-load 64-bit address
// arithmetic on the VALU (the plain-pointer form cost ~5 VALU issues per load, 160 of the ~330 of the gather phase), and a
// row outside the tile is simply an offset outside the resource (reads 0).
struct RowGather {
    __amdgpu_buffer_rsrc_t rs;
    unsigned oob;   // a byte offset outside the resource
    __device__ __forceinline__ RowGather(const float *base, long bytes) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        const int n = __builtin_amdgcn_readfirstlane((int)bytes);
        rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, n, 0x00020000);
        oob = (unsigned)n;
    }
    // element `col` (per lane) of row `row` (uniform) of a matrix with `ld` floats per row; !ok -> 0
    __device__ __forceinline__ float load(bool ok, int row, int ld, int col) const {
        const unsigned so = ok ? (unsigned)row * (unsigned)(ld * 4) : oob;
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)col * 4u,
                                                                             __builtin_amdgcn_readfirstlane(so), 0));
    }
};
#endif
