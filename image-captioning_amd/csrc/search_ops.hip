// Search kernels: log-softmax + top-K over the vocabulary, inverse-CDF sampling, beam selection /
// reorder, best-beam extraction, greedy bookkeeping.  Pure bandwidth / latency work: one pass of
// wavefront reductions per row of logits, and integer + fp64 bookkeeping per image.
#include "search_shared.h"

namespace {

// over the NWV (4 or 2) waves of a workgroup
template <int NWV = 4> __device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if constexpr (NWV == 4) return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    else return fmaxf(red[0], red[1]);
}
template <int NWV = 4> __device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if constexpr (NWV == 4) return (red[0] + red[1]) + (red[2] + red[3]);
    else return red[0] + red[1];
}

// one 256-thread workgroup per row of logits
template <int K>
__global__ __launch_bounds__(256) void logsoftmax_topk_kernel(const float* __restrict__ logits, int ld, int V, float inv_temp,
                                                              int rank_raw, float* __restrict__ out_val,
                                                              int32_t* __restrict__ out_idx, float* __restrict__ lse) {
    __shared__ float red[4];
    __shared__ KV wbest[4];
    __shared__ int winner;
    const int tid = threadIdx.x;
    const float* x = logits + (long long)blockIdx.x * ld;

    float m = -INFINITY;
    for (int i = tid; i < V; i += 256) m = fmaxf(m, x[i] * inv_temp);
    m = block_max(m, red);
    float s = 0.f;
    for (int i = tid; i < V; i += 256) s += expf(x[i] * inv_temp - m);
    s = block_sum(s, red);
    const float logs = logf(s);
    if (lse && tid == 0) lse[blockIdx.x] = m + logs;

    // thread-local sorted top-K of its strided elements (ascending index order => strict '>' keeps
    // the lower index ahead on ties)
    KV loc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) loc[k] = KV{-INFINITY, 0x7fffffff};
    for (int i = tid; i < V; i += 256) {
        const float xs = x[i] * inv_temp;
        KV c{rank_raw ? x[i] : (xs - m) - logs, i};
        if (before(c, loc[K - 1])) {
            loc[K - 1] = c;
#pragma unroll
            for (int k = K - 1; k > 0; --k) {
                if (before(loc[k], loc[k - 1])) {
                    const KV t = loc[k - 1];
                    loc[k - 1] = loc[k];
                    loc[k] = t;
                }
            }
        }
    }
    // K rounds of block-wide selection over the heads of the per-thread lists
    for (int r = 0; r < K; ++r) {
        const KV b = wave_best(loc[0]);
        if ((tid & 63) == 0) wbest[tid >> 6] = b;
        __syncthreads();
        KV best = wbest[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (before(wbest[w], best)) best = wbest[w];
        if (loc[0].idx == best.idx && best.idx != 0x7fffffff) {
            // this thread owned the winner: emit and pop
            if (tid >= 0) {
                const float xs = x[best.idx] * inv_temp;
                out_val[(long long)blockIdx.x * K + r] = (xs - m) - logs;
                out_idx[(long long)blockIdx.x * K + r] = best.idx;
            }
#pragma unroll
            for (int k = 0; k < K - 1; ++k) loc[k] = loc[k + 1];
            loc[K - 1] = KV{-INFINITY, 0x7fffffff};
        } else if (best.idx == 0x7fffffff && tid == 0) {
            out_val[(long long)blockIdx.x * K + r] = -INFINITY;  // fewer than K entries in the row
            out_idx[(long long)blockIdx.x * K + r] = 0;
        }
        __syncthreads();
    }
    (void)winner;
}

// Same result for rows of at most NTH * R entries (NTH threads: 256, or 128 so that the sums are grouped as in the beam
// tail's 128-thread row groups), read from memory ONCE: every load of the row is issued
// up front (one memory round trip instead of three passes of dependent ones) and the maximum, the sum
// and the selection all work on registers.
template <int K, int R, int NTH = 256>
__global__ __launch_bounds__(NTH) void logsoftmax_topk_reg_kernel(const float* __restrict__ logits, int ld, int V, float inv_temp,
                                                                  int rank_raw, float* __restrict__ out_val,
                                                                  int32_t* __restrict__ out_idx, float* __restrict__ lse) {
    __shared__ float red[4];
    __shared__ KV wbest[4];
    const int tid = threadIdx.x;
    const float* x = logits + (long long)blockIdx.x * ld;
    float raw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + NTH * r;
        raw[r] = i < V ? x[i] : -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) m = fmaxf(m, raw[r] * inv_temp);
    m = block_max<NTH / 64>(m, red);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (tid + NTH * r < V) s += expf(raw[r] * inv_temp - m);
    s = block_sum<NTH / 64>(s, red);
    const float logs = logf(s);
    if (lse && tid == 0) lse[blockIdx.x] = m + logs;

    KV loc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) loc[k] = KV{-INFINITY, 0x7fffffff};
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + NTH * r;
        if (i < V) {
            KV c{rank_raw ? raw[r] : (raw[r] * inv_temp - m) - logs, i};
            if (before(c, loc[K - 1])) {
                loc[K - 1] = c;
#pragma unroll
                for (int k = K - 1; k > 0; --k) {
                    if (before(loc[k], loc[k - 1])) {
                        const KV t = loc[k - 1];
                        loc[k - 1] = loc[k];
                        loc[k] = t;
                    }
                }
            }
        }
    }
    for (int r = 0; r < K; ++r) {
        const KV b = wave_best(loc[0]);
        if ((tid & 63) == 0) wbest[tid >> 6] = b;
        __syncthreads();
        KV best = wbest[0];
#pragma unroll
        for (int w = 1; w < NTH / 64; ++w)
            if (before(wbest[w], best)) best = wbest[w];
        if (loc[0].idx == best.idx && best.idx != 0x7fffffff) {
            // this thread owned the winner: emit (always as a log-probability) and pop
            const float xs = x[best.idx] * inv_temp;
            out_val[(long long)blockIdx.x * K + r] = (xs - m) - logs;
            out_idx[(long long)blockIdx.x * K + r] = best.idx;
#pragma unroll
            for (int k = 0; k < K - 1; ++k) loc[k] = loc[k + 1];
            loc[K - 1] = KV{-INFINITY, 0x7fffffff};
        } else if (best.idx == 0x7fffffff && tid == 0) {
            out_val[(long long)blockIdx.x * K + r] = -INFINITY;  // fewer than K entries in the row
            out_idx[(long long)blockIdx.x * K + r] = 0;
        }
        __syncthreads();
    }
}

// one 256-thread workgroup per row; thread i owns the contiguous slice [i*chunk, (i+1)*chunk)
__global__ __launch_bounds__(256) void sample_multinomial_kernel(const float* __restrict__ logits, int ld, int V, float inv_temp,
                                                                 const float* __restrict__ u, float* __restrict__ out_val,
                                                                 int32_t* __restrict__ out_idx) {
    __shared__ float red[4];
    __shared__ double scan[256];
    __shared__ int first;
    const int tid = threadIdx.x;
    const float* x = logits + (long long)blockIdx.x * ld;
    float m = -INFINITY;
    for (int i = tid; i < V; i += 256) m = fmaxf(m, x[i] * inv_temp);
    m = block_max(m, red);
    float s = 0.f;
    for (int i = tid; i < V; i += 256) s += expf(x[i] * inv_temp - m);
    s = block_sum(s, red);
    const float logs = logf(s);
    const int chunk = (V + 255) / 256;
    const int lo = tid * chunk, hi = min(V, lo + chunk);
    double part = 0.0;
    for (int i = lo; i < hi; ++i) part += (double)expf((x[i] * inv_temp - m) - logs);
    scan[tid] = part;
    if (tid == 0) first = V - 1;
    __syncthreads();
    if (tid == 0) {  // 256-entry exclusive scan, serial: negligible next to the row reads
        double run = 0.0;
        for (int i = 0; i < 256; ++i) {
            const double t = scan[i];
            scan[i] = run;
            run += t;
        }
    }
    __syncthreads();
    const double uu = (double)u[blockIdx.x];
    double run = scan[tid];
    for (int i = lo; i < hi; ++i) {
        run += (double)expf((x[i] * inv_temp - m) - logs);
        if (run > uu) {
            atomicMin(&first, i);
            break;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int i = first;
        out_idx[blockIdx.x] = i;
        out_val[blockIdx.x] = (x[i] * inv_temp - m) - logs;
    }
}

__global__ __launch_bounds__(64) void beam_select_kernel(const BeamArgs a) {
    __shared__ SelShared sh;
    const long long base = (long long)blockIdx.x * a.K;
    if ((int)threadIdx.x < a.n_live_slots) slot_store(a, sh, threadIdx.x, slot_load(a, blockIdx.x, threadIdx.x));
    beam_select_body(a, blockIdx.x, threadIdx.x, true, sh, a.top_val + base * a.K, a.top_idx + base * a.K);
}

// One workgroup per image: G = min(K, 1024 / GW) rows at a time.
template <int K, int GW>
__global__ __launch_bounds__(GW * (K < 1024 / GW ? K : 1024 / GW)) void beam_tail_kernel(const TailArgs a) {
    constexpr int G = K < 1024 / GW ? K : 1024 / GW;
    __shared__ TailShared<K, G, GW / 64> ts;
    beam_tail_body<K, GW, G, false>(a, blockIdx.x, threadIdx.x, ts, [] { return true; });
}

__global__ __launch_bounds__(64) void beam_finalize_kernel(const mic_beam_state st, const double* __restrict__ sqrt_tab, int B, int K,
                                                           int t_cap, int32_t* __restrict__ tokens, int32_t* __restrict__ lengths,
                                                           float* __restrict__ score, const float* __restrict__ hist, int S,
                                                           float* __restrict__ attn_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int best = 0;
    double bk = st.score[b * K] / sqrt_tab[st.len[b * K]];
    for (int r = 1; r < K; ++r) {
        const double k = st.score[b * K + r] / sqrt_tab[st.len[b * K + r]];
        if (k > bk) { bk = k; best = r; }
    }
    const int g = b * K + best, len = st.len[g];
    for (int p = lane; p < t_cap; p += 64) tokens[(long long)b * t_cap + p] = p < len ? st.seq[(long long)g * t_cap + p] : -1;
    if (lane == 0) {
        lengths[b] = len;
        score[b] = (float)st.score[g];
    }
    if (hist && attn_out && st.anc) {
        for (int s = lane; s < S; s += 64) {
            float run = 0.f;
            for (int p = 0; p < len - 1; ++p) {
                const int src = st.anc[(long long)g * t_cap + p];
                run += hist[((long long)src * t_cap + p) * S + s];
                attn_out[((long long)b * t_cap + p) * S + s] = run / (float)(p + 1);
            }
        }
    }
}

__global__ void sample_update_kernel(const float* __restrict__ val, const int32_t* __restrict__ idx, int32_t* __restrict__ tokens,
                                     int t_cap, int pos, uint8_t* __restrict__ finished, float* __restrict__ total,
                                     int32_t* live_rows, int N, int pad_id, int eos_id) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const bool fin = finished[n] != 0;
    if (total && !fin) total[n] += val[n];
    const int tok = fin ? pad_id : idx[n];
    tokens[(long long)n * t_cap + pos] = tok;
    const bool nf = fin || tok == eos_id;
    finished[n] = nf;
    if (!nf && live_rows) atomicAdd(live_rows, 1);
}

}  // namespace

extern "C" int mic_logsoftmax_topk(const float* logits, int ld, int N, int V, int K, float inv_temp, int rank_raw, float* out_val,
                                   int32_t* out_idx, float* lse, void* stream) {
    MIC_ARG_CHECK(logits && out_val && out_idx, "mic_logsoftmax_topk: null pointer");
    MIC_ARG_CHECK(N > 0 && V > 0 && ld >= V, "mic_logsoftmax_topk: bad sizes N=%d V=%d ld=%d", N, V, ld);
    MIC_ARG_CHECK(K >= 1 && K <= KMAX, "mic_logsoftmax_topk: K=%d outside [1,%d]", K, KMAX);
    hipStream_t s = (hipStream_t)stream;
#define MIC_TOPK_CASE(KK)                                                                                                  \
    case KK:                                                                                                               \
        if (V <= 256 * 16)                                                                                                 \
            MIC_LAUNCH((logsoftmax_topk_reg_kernel<KK, 16>), dim3(N), dim3(256), 0, s, logits, ld, V, inv_temp, rank_raw,  \
                       out_val, out_idx, lse);                                                                             \
        else                                                                                                               \
            MIC_LAUNCH(logsoftmax_topk_kernel<KK>, dim3(N), dim3(256), 0, s, logits, ld, V, inv_temp, rank_raw, out_val,   \
                       out_idx, lse);                                                                                      \
        break;
    switch (K) {
        MIC_TOPK_CASE(1) MIC_TOPK_CASE(2) MIC_TOPK_CASE(3) MIC_TOPK_CASE(4)
        MIC_TOPK_CASE(5) MIC_TOPK_CASE(6) MIC_TOPK_CASE(7) MIC_TOPK_CASE(8)
    }
#undef MIC_TOPK_CASE
    return 0;
}

extern "C" int mic_sample_multinomial(const float* logits, int ld, int N, int V, float inv_temp, const float* u, float* out_val,
                                      int32_t* out_idx, void* stream) {
    MIC_ARG_CHECK(logits && u && out_val && out_idx, "mic_sample_multinomial: null pointer");
    MIC_ARG_CHECK(N > 0 && V > 0 && ld >= V, "mic_sample_multinomial: bad sizes N=%d V=%d ld=%d", N, V, ld);
    MIC_LAUNCH(sample_multinomial_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, logits, ld, V, inv_temp, u, out_val, out_idx);
    return 0;
}

extern "C" int mic_beam_select(const mic_beam_state* cur, const mic_beam_state* nxt, const float* top_val, const int32_t* top_idx,
                               int32_t* parent, int32_t* live_images, const double* sqrt_tab, int B, int K, int n_live_slots, int t,
                               int t_cap, int eos_id, int finished_in_place, void* stream) {
    MIC_ARG_CHECK(cur && nxt && top_val && top_idx && parent && sqrt_tab, "mic_beam_select: null pointer");
    MIC_ARG_CHECK(cur->score && cur->len && cur->seq && nxt->score && nxt->len && nxt->seq, "mic_beam_select: null state");
    MIC_ARG_CHECK((cur->anc == nullptr) == (nxt->anc == nullptr), "mic_beam_select: anc must be set in both states or neither");
    MIC_ARG_CHECK(B > 0 && K >= 1 && K <= KMAX, "mic_beam_select: B=%d K=%d (K <= %d)", B, K, KMAX);
    MIC_ARG_CHECK(n_live_slots >= 1 && n_live_slots <= K, "mic_beam_select: n_live_slots=%d", n_live_slots);
    MIC_ARG_CHECK(t >= 0 && t + 2 <= t_cap, "mic_beam_select: step %d does not fit t_cap=%d", t, t_cap);
    BeamArgs a;
    a.cur = *cur; a.nxt = *nxt; a.top_val = top_val; a.top_idx = top_idx; a.parent = parent; a.live_images = live_images;
    a.sqrt_tab = sqrt_tab; a.B = B; a.K = K; a.n_live_slots = n_live_slots; a.t = t; a.t_cap = t_cap; a.eos_id = eos_id;
    a.in_place = finished_in_place;
    MIC_LAUNCH(beam_select_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    return 0;
}

extern "C" int mic_beam_tail_supported(int K, int V) { return K >= 1 && K <= KMAX && V >= 1 && V <= 256 * 16; }

extern "C" int mic_beam_tail_ex(const mic_beam_tail_desc* d, void* stream) {
    TailArgs a;
    MIC_TRY(tail_args_from_desc(d, a));
    const int B = d->B, K = d->K;
    hipStream_t s = (hipStream_t)stream;
    MIC_ARG_CHECK(d->row_group_threads == 0 || d->row_group_threads == 256 || d->row_group_threads == 128,
                  "mic_beam_tail: row_group_threads is 0, 128 or 256");
    MIC_ARG_CHECK(d->row_group_threads != 256 || K <= 4, "mic_beam_tail: 256 threads per row take at most four beams");
    const bool narrow = d->row_group_threads == 128 || K > 4;
#define MIC_TAIL_CASE(KK)                                                                                  \
    case KK:                                                                                               \
        if (narrow || KK > 4) MIC_LAUNCH((beam_tail_kernel<KK, 128>), dim3(B), dim3(128 * KK), 0, s, a);   \
        else MIC_LAUNCH((beam_tail_kernel<KK, (KK <= 4 ? 256 : 128)>), dim3(B), dim3(256 * (KK <= 4 ? KK : 1)), 0, s, a); \
        break;
    switch (K) {
        MIC_TAIL_CASE(1) MIC_TAIL_CASE(2) MIC_TAIL_CASE(3) MIC_TAIL_CASE(4)
        MIC_TAIL_CASE(5) MIC_TAIL_CASE(6) MIC_TAIL_CASE(7) MIC_TAIL_CASE(8)
    }
#undef MIC_TAIL_CASE
    return 0;
}

extern "C" int mic_beam_tail(const float* logits, int ld, int V, const mic_beam_state* cur, const mic_beam_state* nxt, int32_t* parent,
                             int32_t* live_images, const double* sqrt_tab, int B, int K, int n_live_slots, int t, int t_cap, int eos_id,
                             const float* tok_emb, const float* pos_emb, float* x, void* xt, int dtype, int H, int vocab, void* stream) {
    MIC_ARG_CHECK(!tok_emb || pos_emb, "mic_beam_tail: the embedding of the next position needs pos_emb");
    mic_beam_tail_desc d{};
    d.logits = logits; d.ld = ld; d.V = V; d.cur = cur; d.nxt = nxt; d.parent = parent; d.live_images = live_images; d.sqrt_tab = sqrt_tab;
    d.B = B; d.K = K; d.n_live_slots = n_live_slots; d.t = t; d.t_cap = t_cap; d.eos_id = eos_id;
    d.tok_emb = tok_emb; d.pos_emb = pos_emb; d.x = x; d.xt = xt; d.ld_xt = H; d.dtype = dtype; d.H = H; d.vocab = vocab;
    return mic_beam_tail_ex(&d, stream);
}

namespace {
__global__ void beam_init_kernel(const mic_beam_state a, const mic_beam_state b, int32_t* __restrict__ live, int N, int t_cap, int sos) {
    const long long total = (long long)N * t_cap;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        a.seq[i] = sos; b.seq[i] = sos;
        if (a.anc) a.anc[i] = 0;
        if (b.anc) b.anc[i] = 0;
        if (i < N) { a.score[i] = 0.0; b.score[i] = 0.0; a.len[i] = 1; b.len[i] = 1; }
        if (live && i < t_cap) live[i] = 0;
    }
}
}  // namespace

extern "C" int mic_beam_init(const mic_beam_state st[2], int32_t* live_images, int N, int t_cap, int sos_id, void* stream) {
    MIC_ARG_CHECK(st && N > 0 && t_cap > 0, "mic_beam_init: bad arguments");
    for (int i = 0; i < 2; ++i) MIC_ARG_CHECK(st[i].score && st[i].len && st[i].seq, "mic_beam_init: null state");
    long long g = ((long long)N * t_cap + 255) / 256;
    if (g > 2048) g = 2048;
    MIC_LAUNCH(beam_init_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, st[0], st[1], live_images, N, t_cap, sos_id);
    return 0;
}

extern "C" int mic_beam_finalize(const mic_beam_state* st, const double* sqrt_tab, int B, int K, int t_cap, int32_t* tokens,
                                 int32_t* lengths, float* score, const float* attn_hist, int S, float* attn_out, void* stream) {
    MIC_ARG_CHECK(st && st->score && st->len && st->seq && sqrt_tab && tokens && lengths && score, "mic_beam_finalize: null pointer");
    MIC_ARG_CHECK(B > 0 && K >= 1 && K <= KMAX && t_cap > 0, "mic_beam_finalize: bad sizes");
    MIC_ARG_CHECK(!attn_out || (attn_hist && st->anc && S > 0), "mic_beam_finalize: attention output needs hist, anc and S");
    MIC_LAUNCH(beam_finalize_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, *st, sqrt_tab, B, K, t_cap, tokens, lengths, score,
                       attn_hist, S, attn_out);
    return 0;
}

extern "C" int mic_sample_update(const float* val, const int32_t* idx, int32_t* tokens, int t_cap, int pos, uint8_t* finished,
                                 float* total, int32_t* live_rows, int N, int pad_id, int eos_id, void* stream) {
    MIC_ARG_CHECK(idx && tokens && finished && N > 0, "mic_sample_update: bad arguments");
    MIC_ARG_CHECK(!total || val, "mic_sample_update: total needs val");
    MIC_ARG_CHECK(pos >= 0 && pos < t_cap, "mic_sample_update: pos=%d outside t_cap=%d", pos, t_cap);
    MIC_LAUNCH(sample_update_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, val, idx, tokens, t_cap, pos,
                       finished, total, live_rows, N, pad_id, eos_id);
    return 0;
}
