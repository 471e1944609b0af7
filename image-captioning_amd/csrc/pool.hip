// Bandwidth-bound encoder helpers: image packing, 3x3/2 max pooling, global average pooling.
// All three move 16 bytes per lane along the channel (innermost) axis of pixel-major tensors.
#include "mic_common.h"

namespace {

// (B,3,H,W) fp32 -> [B][Hp][Wp][4] T with a zero border; one thread per output pixel.
template <typename T>
__global__ void pack_image_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int H, int W, int Hp, int Wp,
                                  int pad_t, int pad_l) {
    const long long total = (long long)B * Hp * Wp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wp), y = (int)((i / Wp) % Hp), b = (int)(i / ((long long)Wp * Hp));
        const int sy = y - pad_t, sx = x - pad_l;
        float c[3] = {0.f, 0.f, 0.f};
        if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
            const float* p = in + ((long long)b * 3 * H + sy) * W + sx;
            c[0] = p[0];
            c[1] = p[(long long)H * W];
            c[2] = p[2LL * H * W];
        }
        T* o = out + i * 4;
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(c[0], c[1], c[2], 0.f);
        } else {
            struct alignas(8) B4 { bf16_t v[4]; } p4;
            p4.v[0] = (bf16_t)c[0]; p4.v[1] = (bf16_t)c[1]; p4.v[2] = (bf16_t)c[2]; p4.v[3] = (bf16_t)0.f;
            *reinterpret_cast<B4*>(o) = p4;
        }
    }
}

template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C, int OH, int OW) {
    constexpr int VEC = Vec16<T>::N;
    const int cv = C / VEC;
    const long long total = (long long)B * OH * OW * cv;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        long long r = i / cv;
        const int ow = (int)(r % OW);
        r /= OW;
        const int oh = (int)(r % OH), b = (int)(r / OH);
        float m[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ih = oh * 2 - 1 + dy;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int iw = ow * 2 - 1 + dx;
                if ((unsigned)iw >= (unsigned)W) continue;
                const Vec16<T> v = ld16(in + (((long long)b * H + ih) * W + iw) * C + c * VEC);
#pragma unroll
                for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], to_f32<T>(v.v[e]));
            }
        }
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.v[e] = from_f32<T>(m[e]);
        st16(out + (((long long)b * OH + oh) * OW + ow) * C + c * VEC, o);
    }
}

// one thread per (b, 16-byte channel chunk); HW is small (49/64), rows are read coalesced across threads
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int HW, int C, int ld_out) {
    constexpr int VEC = Vec16<T>::N;
    const int cv = C / VEC;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * cv) return;
    const int c = i % cv, b = i / cv;
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    for (int p = 0; p < HW; ++p) {
        const Vec16<T> v = ld16(in + ((long long)b * HW + p) * C + c * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] += to_f32<T>(v.v[e]);
    }
    const float inv = 1.f / (float)HW;
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.v[e] = from_f32<T>(s[e] * inv);
    st16(out + (long long)b * ld_out + c * VEC, o);
}

inline int grid_for(long long total, int block) {
    long long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// Antialiased bilinear resize (triangle filter whose support grows with the downscale factor, the
// algorithm behind interpolate(mode="bilinear", antialias=True)) of uint8 HWC images, optional rounding
// back to uint8 levels, /255, per-channel (x - mean) / std, planar fp32 out.  One thread per output pixel;
// the taps of both axes are bounded by the clamped windows [ymin, ymin+ysize) x [xmin, xmin+xsize).
struct ResizeArgs {
    const unsigned char* src;  // [B][H][W][3]
    float* dst;                // [B][3][OH][OW]
    int B, H, W, OH, OW, round_u8;
    float mean[3], inv_std[3];
};

__device__ __forceinline__ void aa_window(int o, float scale, int in_size, int& lo, int& size, float& center, float& invscale) {
    const float support = scale >= 1.0f ? scale : 1.0f;
    center = scale * ((float)o + 0.5f);
    invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
    lo = max((int)(center - support + 0.5f), 0);
    size = min((int)(center + support + 0.5f), in_size) - lo;
}
__device__ __forceinline__ float aa_weight(int j, int lo, float center, float invscale) {
    const float x = fabsf(((float)(j + lo) - center + 0.5f) * invscale);
    return x < 1.0f ? 1.0f - x : 0.0f;
}

__global__ void resize_normalize_kernel(const ResizeArgs a) {
    const long long total = (long long)a.B * a.OH * a.OW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % a.OW), oy = (int)((idx / a.OW) % a.OH), b = (int)(idx / ((long long)a.OW * a.OH));
    const float sy = (float)a.H / (float)a.OH, sx = (float)a.W / (float)a.OW;
    int ylo, ysz, xlo, xsz;
    float yc, yi, xc, xi;
    aa_window(oy, sy, a.H, ylo, ysz, yc, yi);
    aa_window(ox, sx, a.W, xlo, xsz, xc, xi);
    float wxs = 0.f, wys = 0.f;
    for (int i = 0; i < xsz; ++i) wxs += aa_weight(i, xlo, xc, xi);
    for (int j = 0; j < ysz; ++j) wys += aa_weight(j, ylo, yc, yi);
    float acc[3] = {0.f, 0.f, 0.f};
    const unsigned char* img = a.src + (long long)b * a.H * a.W * 3;
    for (int j = 0; j < ysz; ++j) {
        const float wy = aa_weight(j, ylo, yc, yi);
        const unsigned char* row = img + ((long long)(ylo + j) * a.W + xlo) * 3;
        float r[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < xsz; ++i) {
            const float wx = aa_weight(i, xlo, xc, xi);
            r[0] = fmaf(wx, (float)row[3 * i], r[0]);
            r[1] = fmaf(wx, (float)row[3 * i + 1], r[1]);
            r[2] = fmaf(wx, (float)row[3 * i + 2], r[2]);
        }
        acc[0] = fmaf(wy, r[0], acc[0]);
        acc[1] = fmaf(wy, r[1], acc[1]);
        acc[2] = fmaf(wy, r[2], acc[2]);
    }
    const float norm = 1.0f / (wxs * wys);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = acc[c] * norm;
        if (a.round_u8) v = fminf(fmaxf(floorf(v + 0.5f), 0.f), 255.f);
        a.dst[(((long long)b * 3 + c) * a.OH + oy) * a.OW + ox] = (v * (1.0f / 255.0f) - a.mean[c]) * a.inv_std[c];
    }
    }
}

}  // namespace

extern "C" int mic_resize_normalize(const uint8_t* src, float* dst, int B, int H, int W, int OH, int OW, const float mean[3],
                                    const float std[3], int round_u8, void* stream) {
    MIC_ARG_CHECK(src && dst && mean && std, "mic_resize_normalize: null pointer");
    MIC_ARG_CHECK(B > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "mic_resize_normalize: bad sizes B=%d %dx%d -> %dx%d", B, H, W, OH, OW);
    MIC_ARG_CHECK((long long)H <= 64ll * OH && (long long)W <= 64ll * OW, "mic_resize_normalize: downscale beyond 64x is not supported");
    ResizeArgs a;
    a.src = src; a.dst = dst; a.B = B; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.round_u8 = round_u8;
    for (int c = 0; c < 3; ++c) {
        MIC_ARG_CHECK(std[c] > 0.f, "mic_resize_normalize: std[%d] must be positive", c);
        a.mean[c] = mean[c];
        a.inv_std[c] = 1.0f / std[c];
    }
    const long long total = (long long)B * OH * OW;
    MIC_LAUNCH(resize_normalize_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return 0;
}

extern "C" int mic_pack_image(const float* nchw, void* out, int dtype, int B, int H, int W, int pad_t, int pad_b, int pad_l,
                              int pad_r, void* stream) {
    MIC_ARG_CHECK(nchw && out && B > 0 && H > 0 && W > 0, "mic_pack_image: bad arguments");
    MIC_ARG_CHECK(pad_t >= 0 && pad_b >= 0 && pad_l >= 0 && pad_r >= 0, "mic_pack_image: negative padding");
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "mic_pack_image: bad dtype %d", dtype);
    const int Hp = H + pad_t + pad_b, Wp = W + pad_l + pad_r;
    const long long total = (long long)B * Hp * Wp;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MIC_BF16)
        MIC_LAUNCH(pack_image_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, s, nchw, (bf16_t*)out, B, H, W, Hp, Wp, pad_t, pad_l);
    else
        MIC_LAUNCH(pack_image_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, nchw, (float*)out, B, H, W, Hp, Wp, pad_t, pad_l);
    return 0;
}

extern "C" int mic_maxpool3x3s2(const void* in, void* out, int dtype, int B, int H, int W, int C, void* stream) {
    MIC_ARG_CHECK(in && out && B > 0 && H > 0 && W > 0 && C > 0, "mic_maxpool3x3s2: bad arguments");
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "mic_maxpool3x3s2: bad dtype %d", dtype);
    const int vec = 16 / (int)mic_dtype_size(dtype);
    MIC_ARG_CHECK(C % vec == 0, "mic_maxpool3x3s2: C=%d must be a multiple of %d", C, vec);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)B * OH * OW * (C / vec);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MIC_BF16)
        MIC_LAUNCH(maxpool_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, B, H, W, C, OH, OW);
    else
        MIC_LAUNCH(maxpool_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const float*)in, (float*)out, B, H, W, C, OH, OW);
    return 0;
}

extern "C" int mic_global_avgpool(const void* in, void* out, int dtype, int B, int HW, int C, int ld_out, void* stream) {
    MIC_ARG_CHECK(in && out && B > 0 && HW > 0 && C > 0 && ld_out >= C, "mic_global_avgpool: bad arguments");
    MIC_ARG_CHECK(dtype == MIC_F32 || dtype == MIC_BF16, "mic_global_avgpool: bad dtype %d", dtype);
    const int vec = 16 / (int)mic_dtype_size(dtype);
    MIC_ARG_CHECK(C % vec == 0 && ld_out % vec == 0, "mic_global_avgpool: C and ld_out must be multiples of %d", vec);
    const int total = B * (C / vec);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MIC_BF16)
        MIC_LAUNCH(avgpool_kernel<bf16_t>, dim3((total + 255) / 256), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, B, HW, C, ld_out);
    else
        MIC_LAUNCH(avgpool_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, s, (const float*)in, (float*)out, B, HW, C, ld_out);
    return 0;
}
