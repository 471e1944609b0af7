// ResNet stem in one launch (bf16): 7x7 stride-2 convolution + folded BatchNorm + ReLU + 3x3 stride-2 max-pool
// (torchvision resnet50 conv1 / bn1 / relu / maxpool, reached from models/encoders/transformer.py:46).
//
// Layer by layer the 112x112x64 convolution output is written to memory and read back by the pooling kernel
// (0.8 GB per batch of 256), and the convolution as an implicit GEMM fetches every input pixel once per kernel
// row and output pixel that uses it.  Here one workgroup owns a tile of 8 x 16 POOLED pixels: it stages the
// 39 x 72-pixel patch of the zero-bordered 4-channel image that the 17 x 33 convolution outputs under the tile
// need (22 KiB) and the whole 64 x 224 weight matrix (28 KiB) in LDS once, runs the convolution of all 561
// outputs from there on the matrix cores (K = 7 kernel rows x 32: one kernel row of 8 pixels x 4 channels is 64
// contiguous bytes of the patch, so an MFMA operand is a single 16-byte LDS read at an address that is linear in
// the output pixel), leaves the bf16 outputs in LDS and pools them.  The convolution outputs on the tile border
// are computed by two neighbouring tiles (561 outputs for 512 owned: +10 %).
//
// Accumulation order per output (kernel rows ascending, 32 products per MFMA) and the rounding points (fp32
// accumulator + bias, ReLU, bf16) are those of the implicit-GEMM kernel, so the result equals mic_conv2d followed
// by mic_maxpool3x3s2 bit for bit.
#include "mic_common.h"

namespace {

constexpr int PTH = 8, PTW = 16;                 // pooled pixels per tile
constexpr int CTH = 2 * PTH + 1, CTW = 2 * PTW + 1;  // convolution outputs under the tile: 17 x 33
constexpr int NCONV = CTH * CTW;                 // 561
constexpr int MT = (NCONV + 15) / 16;            // 36 MFMA row tiles
constexpr int PROWS = 2 * (CTH - 1) + 7;         // 39 patch rows
constexpr int PCHUNK = (2 * (CTW - 1) + 8) / 2;  // 36 16-byte chunks (2 pixels x 4 channels) per patch row
constexpr int PROW_B = PCHUNK * 16;              // 576 bytes
constexpr int PATCH_B = PROWS * PROW_B;          // 22,464
constexpr int W_B = 7 * 64 * 64;                 // 28,672: [kernel row][out channel][64 bytes]
constexpr int CROW_B = 144;                      // one convolution output in LDS: 64 bf16 + 16 bytes of padding
constexpr int CONV_B = NCONV * CROW_B;           // 80,784
constexpr int SMEM = CONV_B > PATCH_B + W_B ? CONV_B : PATCH_B + W_B;
static_assert(MT % 4 == 0, "row tiles split evenly over four waves");

struct StemArgs {
    const bf16_t* img;   // [B][Hp][Wp][4]
    const bf16_t* w;     // [>= 64][224], k = kernel row * 32 + pixel * 4 + channel
    const float* bias;   // [64]
    bf16_t* out;         // [B][H2][W2][64]
    const bf16_t* w1;    // optional following 1x1 convolution 64 -> 64 (+ ReLU): [>= 64][64]
    const float* b1;     // [64]
    bf16_t* t1;          // [B][H2][W2][64]
    int B, Hp, Wp, H1, W1, H2, W2, tiles_x, tiles_y;
};

__device__ __forceinline__ int w_off(int row, int q) { return row * 64 + ((q ^ (((row >> 3) & 1) << 1)) << 4); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// C1: the first 1x1 convolution of the following bottleneck (64 -> 64, ReLU) rides along: the pooled tile is its operand.
template <bool C1>
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const StemArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* patch = smem;
    unsigned char* wl = smem + PATCH_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y, b = t / a.tiles_y;
    const int py0 = ty * PTH, px0 = tx * PTW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;  // first convolution output under the tile (may be -1: padding)

    // ---- stage the image patch and the weights (all requests first, then the LDS writes) ----
    constexpr int NP = (PROWS * PCHUNK + 255) / 256, NW = 7 * 64 * 4 / 256;
    u32x4 pv[NP], wv[NW];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = tid + 256 * i, pr = c / PCHUNK, pc = c - pr * PCHUNK;
        const int gy = 2 * cy0 + pr, gx = 2 * cx0 + 2 * pc;
        pv[i] = u32x4{0u, 0u, 0u, 0u};
        if (c < PROWS * PCHUNK && (unsigned)gy < (unsigned)a.Hp && gx >= 0 && gx + 1 < a.Wp)
            pv[i] = *reinterpret_cast<const u32x4*>(a.img + (((long long)b * a.Hp + gy) * a.Wp + gx) * 4);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int c = tid + 256 * i, kt = c >> 8, row = (c & 255) >> 2, q = c & 3;
        wv[i] = *reinterpret_cast<const u32x4*>(a.w + row * 224 + kt * 32 + q * 8);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = tid + 256 * i;
        if (c < PROWS * PCHUNK) *reinterpret_cast<u32x4*>(patch + c * 16) = pv[i];
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int c = tid + 256 * i, kt = c >> 8, row = (c & 255) >> 2, q = c & 3;
        *reinterpret_cast<u32x4*>(wl + kt * 64 * 64 + w_off(row, q)) = wv[i];
    }
    __syncthreads();

    // ---- convolution: wave w owns row tiles 9w .. 9w+8, all 64 channels ----
    constexpr int TPW = MT / 4;
    int xoff[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        int m = (wave * TPW + i) * 16 + (lane & 15);
        m = m < NCONV ? m : NCONV - 1;
        const int ry = m / CTW, rx = m - ry * CTW;
        xoff[i] = 2 * ry * PROW_B + rx * 16 + (lane >> 4) * 16;
    }
    int woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) woff[j] = w_off(j * 16 + (lane & 15), lane >> 4);
    f32x4_t acc[TPW][4];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
        u32x4 wf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const u32x4*>(wl + kh * 64 * 64 + woff[j]);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const u32x4 xf = *reinterpret_cast<const u32x4*>(patch + xoff[i] + kh * PROW_B);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j]), __builtin_bit_cast(bf16x8_t, xf),
                                                                    acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();  // every wave is done with the patch and the weights: the convolution tile takes their place

    // ---- bias + ReLU + bf16 into LDS; outputs outside the image count as 0 (every pooling window holds a real output,
    //      and real outputs are >= 0 after the ReLU, so 0 never wins over the true maximum) ----
    unsigned char* ct = smem;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = j * 16 + (lane >> 4) * 4;
        const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int m = (wave * TPW + i) * 16 + (lane & 15);
            if (m >= NCONV) continue;
            const int ry = m / CTW, rx = m - ry * CTW;
            const bool real = (unsigned)(cy0 + ry) < (unsigned)a.H1 && (unsigned)(cx0 + rx) < (unsigned)a.W1;
            const float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            struct alignas(8) B4 { bf16_t v[4]; } o;
            o.v[0] = (bf16_t)(real && v0 > 0.f ? v0 : 0.f);
            o.v[1] = (bf16_t)(real && v1 > 0.f ? v1 : 0.f);
            o.v[2] = (bf16_t)(real && v2 > 0.f ? v2 : 0.f);
            o.v[3] = (bf16_t)(real && v3 > 0.f ? v3 : 0.f);
            *reinterpret_cast<B4*>(ct + m * CROW_B + n * 2) = o;
        }
    }
    __syncthreads();

    // ---- 3x3 / 2 max-pool: non-negative bf16 values order like their bit patterns ----
    constexpr int NIT = PTH * PTW * 8 / 256;
    u16x8 pooled[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int it = tid + 256 * k, pp = it >> 3, ch = it & 7;
        const int ppy = pp / PTW, ppx = pp - ppy * PTW;
        const int py = py0 + ppy, px = px0 + ppx;
        u16x8 m = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const u16x8 v = *reinterpret_cast<const u16x8*>(ct + ((2 * ppy + dy) * CTW + 2 * ppx + dx) * CROW_B + ch * 16);
                m = __builtin_elementwise_max(m, v);
            }
        pooled[k] = m;
        if (py < a.H2 && px < a.W2) *reinterpret_cast<u16x8*>(a.out + (((long long)b * a.H2 + py) * a.W2 + px) * 64 + ch * 8) = m;
    }
    if constexpr (C1) {
        // ---- t1 = relu(pooled * W1^T + b1): the pooled tile (128 pixels x 64 channels) and W1 as MFMA operand images ----
        constexpr int A_B = 2 * 128 * 64, W1_B = 2 * 64 * 64, T_OFF = A_B + W1_B;  // bytes; the t1 tile is staged behind them
        static_assert(T_OFF + 128 * CROW_B <= SMEM, "1x1 stage fits the convolution tile's LDS");
        u32x4 w1v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
            w1v[i] = *reinterpret_cast<const u32x4*>(a.w1 + row * 64 + ch * 8);
        }
        __syncthreads();  // every pooling read of the convolution tile is done
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int it = tid + 256 * k, pp = it >> 3, ch = it & 7;
            *reinterpret_cast<u16x8*>(smem + (ch >> 2) * 128 * 64 + w_off(pp, ch & 3)) = pooled[k];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
            *reinterpret_cast<u32x4*>(smem + A_B + (ch >> 2) * 64 * 64 + w_off(row, ch & 3)) = w1v[i];
        }
        __syncthreads();
        f32x4_t acc1[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            u32x4 xf[2], wf[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) xf[i] = *reinterpret_cast<const u32x4*>(smem + kt * 128 * 64 + w_off(wave * 32 + i * 16 + (lane & 15), lane >> 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const u32x4*>(smem + A_B + kt * 64 * 64 + w_off(j * 16 + (lane & 15), lane >> 4));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j]), __builtin_bit_cast(bf16x8_t, xf[i]),
                                                                         acc1[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = j * 16 + (lane >> 4) * 4;
            const float4 bv = *reinterpret_cast<const float4*>(a.b1 + n);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pp = wave * 32 + i * 16 + (lane & 15);
                struct alignas(8) B4 { bf16_t v[4]; } o;
                o.v[0] = (bf16_t)fmaxf(acc1[i][j][0] + bv.x, 0.f);
                o.v[1] = (bf16_t)fmaxf(acc1[i][j][1] + bv.y, 0.f);
                o.v[2] = (bf16_t)fmaxf(acc1[i][j][2] + bv.z, 0.f);
                o.v[3] = (bf16_t)fmaxf(acc1[i][j][3] + bv.w, 0.f);
                *reinterpret_cast<B4*>(smem + T_OFF + pp * CROW_B + n * 2) = o;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int it = tid + 256 * k, pp = it >> 3, ch = it & 7;
            const int ppy = pp / PTW, ppx = pp - ppy * PTW;
            const int py = py0 + ppy, px = px0 + ppx;
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + T_OFF + pp * CROW_B + ch * 16);
            if (py < a.H2 && px < a.W2) *reinterpret_cast<u32x4*>(a.t1 + (((long long)b * a.H2 + py) * a.W2 + px) * 64 + ch * 8) = v;
        }
    }
}

}  // namespace

extern "C" int mic_stem_pool_supported(int dtype) { return dtype == MIC_BF16; }

namespace {
int stem_pool_impl(const void* packed, const void* w, const float* bias, void* out, const void* w1, const float* b1, void* t1, int dtype,
                   int B, int Hp, int Wp, int H1, int W1, int w_rows, int w1_rows, void* stream) {
    MIC_ARG_CHECK(packed && w && bias && out, "mic_stem_pool: null pointer");
    MIC_ARG_CHECK(mic_stem_pool_supported(dtype), "mic_stem_pool: bf16 only (dtype %d)", dtype);
    MIC_ARG_CHECK(B > 0 && H1 > 0 && W1 > 0 && Wp % 2 == 0, "mic_stem_pool: bad sizes B=%d H1=%d W1=%d Wp=%d", B, H1, W1, Wp);
    // every real convolution output reads rows 2y .. 2y+6 and pixels 2x .. 2x+7 of the packed image
    MIC_ARG_CHECK(2 * (H1 - 1) + 7 <= Hp && 2 * (W1 - 1) + 8 <= Wp, "mic_stem_pool: a %dx%d output does not fit a packed image of %dx%d", H1, W1,
                  Hp, Wp);
    MIC_ARG_CHECK(w_rows >= 64, "mic_stem_pool: weight matrix has %d rows, 64 are read", w_rows);
    MIC_ARG_CHECK((((uintptr_t)packed | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)out) & 15) == 0, "mic_stem_pool: pointers must be 16-byte aligned");
    if (w1) {
        MIC_ARG_CHECK(b1 && t1 && w1_rows >= 64 && t1 != out, "mic_stem_pool_conv1: the following convolution needs b1, t1 and 64 weight rows (%d)", w1_rows);
        MIC_ARG_CHECK((((uintptr_t)w1 | (uintptr_t)b1 | (uintptr_t)t1) & 15) == 0, "mic_stem_pool_conv1: pointers must be 16-byte aligned");
    }
    StemArgs a;
    a.img = (const bf16_t*)packed; a.w = (const bf16_t*)w; a.bias = bias; a.out = (bf16_t*)out;
    a.w1 = (const bf16_t*)w1; a.b1 = b1; a.t1 = (bf16_t*)t1;
    a.B = B; a.Hp = Hp; a.Wp = Wp; a.H1 = H1; a.W1 = W1;
    a.H2 = (H1 + 2 - 3) / 2 + 1; a.W2 = (W1 + 2 - 3) / 2 + 1;
    a.tiles_y = (a.H2 + PTH - 1) / PTH; a.tiles_x = (a.W2 + PTW - 1) / PTW;
    const long long grid = (long long)B * a.tiles_y * a.tiles_x;
    MIC_ARG_CHECK(grid < (1ll << 31), "mic_stem_pool: grid too large");
    // counted with the encoder's other convolution launches: 7*7*3 real products per output and channel
    const double flops = 2.0 * B * H1 * W1 * 64 * 147 + (w1 ? 2.0 * B * a.H2 * a.W2 * 64 * 64 : 0.0);
    void* rec = mic_prof_open((hipStream_t)stream, 1, flops);
    if (w1)
        MIC_LAUNCH(stem_pool_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    else
        MIC_LAUNCH(stem_pool_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    return mic_prof_close(rec, (hipStream_t)stream);
}
}  // namespace

extern "C" int mic_stem_pool(const void* packed, const void* w, const float* bias, void* out, int dtype, int B, int Hp, int Wp, int H1,
                             int W1, int w_rows, void* stream) {
    return stem_pool_impl(packed, w, bias, out, nullptr, nullptr, nullptr, dtype, B, Hp, Wp, H1, W1, w_rows, 0, stream);
}

extern "C" int mic_stem_pool_conv1(const void* packed, const void* w, const float* bias, void* out, const void* w1, const float* b1,
                                   void* t1, int dtype, int B, int Hp, int Wp, int H1, int W1, int w_rows, int w1_rows, void* stream) {
    MIC_ARG_CHECK(w1, "mic_stem_pool_conv1: null pointer");
    return stem_pool_impl(packed, w, bias, out, w1, b1, t1, dtype, B, Hp, Wp, H1, W1, w_rows, w1_rows, stream);
}
