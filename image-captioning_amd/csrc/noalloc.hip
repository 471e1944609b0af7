// mic_tr_step_static: one transformer decoder step whose temporaries live in a device array that is
// part of the library's code object, for callers that cannot allocate (code under HIP graph capture,
// tight serving loops, harnesses that forbid hipMalloc).
//
// The kernels are those of mic_tr_step and take the buffer through the same workspace fields; the only
// difference is where the bytes come from.  Every GPU the library runs on holds its own copy of the
// array, at its own address, until the process ends.
#include "mic_common.h"

#include <mutex>

namespace {

// Temporaries of one decoder step (residual stream, its copy in the compute dtype, pre-norm sums, packed
// in-projection output, attention output, feed-forward hidden rows, LayerNorm row statistics).
static __device__ __align__(256) unsigned char scratch[MIC_STATIC_SCRATCH_BYTES];

// In host code the array's name is a stand-in, not a device address: the address of each device's copy
// is asked for once per device (with that device current) and kept here.
constexpr int MAX_DEVICES = 64;
std::mutex g_mu;
unsigned char* g_scratch_ptr[MAX_DEVICES] = {};

int scratch_base(unsigned char** out) {
    int dev = -1;
    MIC_HIP_CHECK(hipGetDevice(&dev));
    MIC_ARG_CHECK(dev >= 0 && dev < MAX_DEVICES, "mic_tr_step_static: device ordinal %d outside [0,%d)", dev, MAX_DEVICES);
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        MIC_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        MIC_ARG_CHECK(p && ((uintptr_t)p & 255) == 0, "mic_tr_step_static: scratch symbol resolved to a misaligned address");
        g_scratch_ptr[dev] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

inline int64_t up256(int64_t v) { return (v + 255) & ~(int64_t)255; }

}  // namespace

extern "C" int mic_tr_step_static(const mic_tr_decoder* dec, const mic_tr_workspace* io, const int32_t* tokens, int ld_tok,
                                  const int32_t* anc, int t, int attn_mode, void* stream) {
    MIC_ARG_CHECK(dec && io, "mic_tr_step_static: null pointer");
    MIC_ARG_CHECK(dec->dtype == MIC_F32 || dec->dtype == MIC_BF16, "mic_tr_step_static: bad dtype");
    MIC_ARG_CHECK(io->N > 0 && dec->H > 0, "mic_tr_step_static: bad sizes N=%d H=%d", io->N, dec->H);
    const int64_t N = io->N, H = dec->H, sz = (int64_t)mic_dtype_size(dec->dtype), parts = (H + 127) / 128;
    // x, xt, y, the packed in-projection output, o, ffh, stats[0], stats[1], each starting on a 256-byte boundary.  The
    // fused-attention step (taken when the caller's workspace carries the folded value caches) has a wider packed
    // projection (q | k | folded values) and no separate attention output.
    MIC_ARG_CHECK(dec->layers >= 1 && dec->layers <= MIC_MAX_LAYERS, "mic_tr_step_static: %d layers (max %d)", dec->layers, MIC_MAX_LAYERS);
    unsigned char* base = nullptr;
    MIC_TRY(scratch_base(&base));
    mic_tr_workspace ws = *io;
    ws.stats[0] = ws.stats[1] = (float*)base;  // present for the path decision; placed for real below
    const bool fused = mic_tr_fused_eligible(dec, &ws);
    const int64_t bytes[8] = {N * H * 4, N * H * sz, N * H * 4, fused ? N * (2 + dec->heads) * H * 4 : N * 3 * H * 4,
                              fused ? 0 : N * H * sz, N * 2 * H * sz, N * parts * 8, N * parts * 8};
    int64_t off[8], need = 0;
    for (int i = 0; i < 8; ++i) {
        off[i] = need;
        need += up256(bytes[i]);
    }
    MIC_ARG_CHECK(need <= (int64_t)MIC_STATIC_SCRATCH_BYTES,
                  "mic_tr_step_static: %d rows of hidden size %d need %lld bytes of temporaries, the built-in buffer holds %lld",
                  io->N, dec->H, (long long)need, (long long)MIC_STATIC_SCRATCH_BYTES);
    ws.x = (float*)(base + off[0]);
    ws.xt = base + off[1];
    ws.y = (float*)(base + off[2]);
    ws.qkv = fused ? nullptr : (float*)(base + off[3]);
    ws.qkvf = fused ? (float*)(base + off[3]) : nullptr;
    ws.o = fused ? nullptr : base + off[4];
    ws.ffh = base + off[5];
    ws.stats[0] = (float*)(base + off[6]);
    ws.stats[1] = (float*)(base + off[7]);
    // every temporary is written by the step before it is read (embedding -> x / xt, the sub-layer tails -> y and
    // the statistics, the projections -> qkv / o / ffh), so nothing has to start at zero
    return mic_tr_step(dec, &ws, tokens, ld_tok, anc, t, attn_mode, stream);
}
