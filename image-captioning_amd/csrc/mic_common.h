// Internal helpers shared by the kernels of libmic_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include <atomic>

#include "../../include/mic_hip.h"

#define MIC_WAVE 64

// ---- error plumbing: every entry point returns 0, a negative argument error, or a hipError_t ----
void mic_set_error(const char* fmt, ...);

#define MIC_ARG_CHECK(cond, ...)                      \
    do {                                              \
        if (!(cond)) {                                \
            mic_set_error(__VA_ARGS__);               \
            return MIC_ERR_ARG;                       \
        }                                             \
    } while (0)

#define MIC_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            mic_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                              \
        }                                                                                \
    } while (0)

// hipGetLastError() reports the last error of ANY earlier runtime call on this thread (also other
// libraries'), so clear it before a launch and read it right after.
#define MIC_LAUNCH(kernel, grid, block, lds, stream, ...)                        \
    do {                                                                         \
        (void)hipGetLastError();                                                 \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);       \
        MIC_HIP_CHECK(hipGetLastError());                                        \
    } while (0)

#define MIC_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != 0) return rc_; \
    } while (0)

// Raising a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) applies to the current device only, so
// the grant is remembered per (kernel, device): `mask` is the caller's static word for that kernel, one bit per device
// ordinal (ordinals beyond 63 ask every time).  Safe from several host threads: asking twice is harmless.
static inline int mic_grant_dynamic_lds(const void* func, int bytes, std::atomic<uint64_t>& mask) {
    int dev = 0;
    MIC_HIP_CHECK(hipGetDevice(&dev));
    const uint64_t bit = dev >= 0 && dev < 64 ? 1ull << dev : 0;
    if (bit && (mask.load(std::memory_order_acquire) & bit)) return 0;
    MIC_HIP_CHECK(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (bit) mask.fetch_or(bit, std::memory_order_release);
    return 0;
}

// ---- per-launch event timing shared by the GEMM-class kernels (mic_profile_begin / mic_profile_end) ----
// open returns NULL when profiling is off, the stream is capturing or the pool is used up; slot = 2 * (tiles
// of fewer than 128 rows) + (bf16).  close records the closing event.
void* mic_prof_open(hipStream_t s, int slot, double flops);
int mic_prof_close(void* rec, hipStream_t s);
void* mic_prof_open_scope(hipStream_t s, int slot);
// While one of these lives on a thread, the profiled launches of that thread share its single pair of events (their FLOPs
// and count are summed into it) instead of recording a pair each.  Meant for runs of back-to-back launches of one slot.
struct MicProfScope {
    MicProfScope(hipStream_t s, int slot);
    ~MicProfScope();
    MicProfScope(const MicProfScope&) = delete;
    MicProfScope& operator=(const MicProfScope&) = delete;
    hipStream_t stream;
};

// ---- decoder-step internals shared between the drivers ----
// whether a step on (dec, ws) can take the fused-attention path, the temporary ws->qkvf aside
bool mic_tr_fused_eligible(const mic_tr_decoder* dec, const mic_tr_workspace* ws);
// the one-launch step (decode_persist.hip): whether (dec, ws) can take it, and the launch of the layers and the vocabulary
// projection on the residual stream already in ws->x / ws->xt
bool mic_tr_persist_eligible(const mic_tr_decoder* dec, const mic_tr_workspace* ws);
// `tail` (or null): the beam-search tail of the step, run as the launch's last phase (mic_tr_persist_tail_ok)
bool mic_tr_persist_tail_ok(const mic_tr_decoder* dec, const mic_tr_workspace* ws, int K, int V);
int mic_tr_persist_launch(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const int32_t* anc, int t, int attn_mode,
                          const mic_beam_tail_desc* tail, hipStream_t s);

// ---- element types ----
typedef __bf16 bf16_t;
typedef bf16_t bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

static inline size_t mic_dtype_size(int dt) { return dt == MIC_BF16 ? 2 : 4; }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 16-byte vector of T: 4 floats or 8 bf16
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    float v[4];
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    bf16_t v[8];
};

template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T* p) {
    uint4 r = *reinterpret_cast<const uint4*>(p);
    return __builtin_bit_cast(Vec16<T>, r);
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec16<T>& v) {
    *reinterpret_cast<uint4*>(p) = __builtin_bit_cast(uint4, v);
}

// LayerNorm pieces shared by the stand-alone kernel and the GEMM epilogue: written with explicit fused multiply-adds and
// contraction switched off, so both compile to the same arithmetic and produce the same bits from the same inputs.
__device__ __forceinline__ float ln_sum4(float a, float b, float c, float d) { return (a + b) + (c + d); }
__device__ __forceinline__ float ln_sq4(float a, float b, float c, float d, float mean) {
#pragma clang fp contract(off)
    const float da = a - mean, db = b - mean, dc = c - mean, dd = d - mean;
    return __builtin_fmaf(da, da, db * db) + __builtin_fmaf(dc, dc, dd * dd);
}
__device__ __forceinline__ float ln_apply(float v, float mean, float rstd, float g, float b) {
#pragma clang fp contract(off)
    return __builtin_fmaf((v - mean) * rstd, g, b);
}

// Wave-wide reductions on the data-parallel-primitive path of the vector ALU (no LDS crossbar round trips): within each
// row of 16 lanes by quad permutes and row mirrors, then the four row results are read into scalar registers and combined,
// so every lane returns the same value.
#define MIC_DPP_F32(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xf, 0xf, false))
__device__ __forceinline__ float wave_sum(float v) {
    v += MIC_DPP_F32(v, 0xB1);   // quad_perm [1,0,3,2]
    v += MIC_DPP_F32(v, 0x4E);   // quad_perm [2,3,0,1]
    v += MIC_DPP_F32(v, 0x141);  // row_half_mirror
    v += MIC_DPP_F32(v, 0x140);  // row_mirror: every lane holds the sum of its row of 16
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, MIC_DPP_F32(v, 0xB1));
    v = fmaxf(v, MIC_DPP_F32(v, 0x4E));
    v = fmaxf(v, MIC_DPP_F32(v, 0x141));
    v = fmaxf(v, MIC_DPP_F32(v, 0x140));
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
