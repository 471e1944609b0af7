// Native drivers: chain the kernels of one phase (encoder trunk, one decoder step, a run of search
// steps) on one stream so the host makes one call per phase instead of one per kernel.
#include "mic_common.h"

#include <stdlib.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

// ---- HIP-graph cache for the launch-bound decode loops ---------------------------------------
// A run of decode steps is ~40 dependent small kernels per step whose arguments depend only on the
// driver's arguments, so the whole run is captured once per distinct argument set and replayed.
//
// The cache keeps the MIC_GRAPH_CACHE most recently used executables.  hipGraphLaunch is asynchronous
// and lanes launch on different streams, so an executable that falls out of the cache may still be
// running: it is parked with the event recorded after its last launch and destroyed only once that
// event has completed.
constexpr size_t MIC_GRAPH_CACHE = 32;
struct GraphEntry {
    hipGraphExec_t exec = nullptr;
    hipEvent_t done = nullptr;  // recorded after every launch, on the launch stream
    uint64_t stamp = 0;         // value of GraphCache::clock at the last launch
};
struct GraphCache {
    std::mutex mu;
    std::map<std::string, GraphEntry> map;
    std::vector<GraphEntry> retired;
    uint64_t clock = 0;
};
GraphCache g_graphs;

// under g_graphs.mu: free the parked executables whose last launch has finished
void sweep_retired() {
    auto& r = g_graphs.retired;
    for (size_t i = 0; i < r.size();) {
        if (hipEventQuery(r[i].done) == hipSuccess) {
            (void)hipGraphExecDestroy(r[i].exec);
            (void)hipEventDestroy(r[i].done);
            r[i] = r.back();
            r.pop_back();
        } else {
            ++i;
        }
    }
    (void)hipGetLastError();  // hipErrorNotReady of a pending event is not an error of the caller
}

// body(stream) enqueues the work.  Capture runs on a stream the library owns (the caller's may be
// the legacy default stream, which cannot be captured); the executable is launched on the caller's.
// Capture and instantiation take milliseconds and run outside the lock, so other lanes keep launching.
template <class Body> int run_graphed(const std::string& key, hipStream_t s, Body&& body) {
    std::unique_lock<std::mutex> lk(g_graphs.mu);
    auto it = g_graphs.map.find(key);
    if (it == g_graphs.map.end()) {
        lk.unlock();
        hipGraph_t graph = nullptr;
        hipStream_t cap = nullptr;
        MIC_HIP_CHECK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) {
            (void)hipStreamDestroy(cap);
            MIC_HIP_CHECK(e);
        }
        const int rc = body(cap);
        e = hipStreamEndCapture(cap, &graph);
        (void)hipStreamDestroy(cap);
        if (rc != 0) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        MIC_HIP_CHECK(e);
        GraphEntry fresh;
        const hipError_t ei = hipGraphInstantiate(&fresh.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        MIC_HIP_CHECK(ei);
        e = hipEventCreateWithFlags(&fresh.done, hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)hipGraphExecDestroy(fresh.exec);
            MIC_HIP_CHECK(e);
        }
        lk.lock();
        it = g_graphs.map.find(key);
        if (it != g_graphs.map.end()) {  // another lane captured the same key meanwhile: ours was never launched
            (void)hipGraphExecDestroy(fresh.exec);
            (void)hipEventDestroy(fresh.done);
        } else {
            sweep_retired();
            while (g_graphs.map.size() >= MIC_GRAPH_CACHE) {  // park the least recently launched executable
                auto old = g_graphs.map.begin();
                for (auto j = g_graphs.map.begin(); j != g_graphs.map.end(); ++j)
                    if (j->second.stamp < old->second.stamp) old = j;
                g_graphs.retired.push_back(old->second);
                g_graphs.map.erase(old);
            }
            it = g_graphs.map.emplace(key, fresh).first;
        }
    }
    // launch and event record stay under the lock: an entry is parked only together with the event of its last launch
    it->second.stamp = ++g_graphs.clock;
    MIC_HIP_CHECK(hipGraphLaunch(it->second.exec, s));
    MIC_HIP_CHECK(hipEventRecord(it->second.done, s));
    return 0;
}

template <class... Ts> std::string make_key(const char* tag, const Ts&... parts) {
    std::string k(tag);
    auto add = [&k](const void* p, size_t n) { k.append(reinterpret_cast<const char*>(p), n); };
    (add(&parts, sizeof(parts)), ...);
    return k;
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// dense linear layer y[M][ldc] = act(x[M][ld_in] * W^T + b + res)
int linear(int dtype, const void* x, int ld_in, int M, int K, const void* w, const float* bias, const void* res, int res_dtype,
           int ld_res, void* out, int out_dtype, int ldc, int N, int relu, hipStream_t s, int k_valid = 0) {
    mic_conv_desc d{};
    d.k_valid = k_valid;
    d.in = x; d.w = w; d.bias = bias; d.res = res; d.out = out;
    d.dtype = dtype; d.out_dtype = out_dtype; d.res_dtype = res_dtype;
    d.M = M; d.N = N; d.K = K;
    d.OH = d.OW = d.IH = d.IW = 1;
    d.in_img_stride = ld_in; d.in_row_stride = 0; d.in_pix_stride = 4;
    d.kw_taps = 1; d.run = K; d.stride = 1; d.pad_h = d.pad_w = 0;
    d.ldc = ldc; d.ld_res = ld_res; d.relu = relu;
    return mic_conv2d(&d, s);
}

// fused conv layer on a dense NHWC tensor [B][IH][IW][cin]
// scratch of the launches of the forward pass under way on this thread (split-K convolutions; mic_conv_desc::scratch)
thread_local void* t_conv_scratch = nullptr;
thread_local int64_t t_conv_scratch_bytes = 0;

int conv_layer(int dtype, const mic_conv_layer& L, const void* in, int B, int IH, int IW, const void* res, void* out, int* OHo,
               int* OWo, hipStream_t s) {
    const int OH = (IH + 2 * L.pad - L.kh) / L.stride + 1, OW = (IW + 2 * L.pad - L.kw) / L.stride + 1;
    mic_conv_desc d{};
    d.in = in; d.w = L.w; d.bias = L.bias; d.res = res; d.out = out;
    d.dtype = d.out_dtype = d.res_dtype = dtype;
    d.M = B * OH * OW; d.N = L.cout; d.K = L.kh * L.kw * L.cin;
    d.OH = OH; d.OW = OW; d.IH = IH; d.IW = IW;
    d.in_img_stride = (int64_t)IH * IW * L.cin; d.in_row_stride = (int64_t)IW * L.cin; d.in_pix_stride = L.cin;
    d.kw_taps = L.kw; d.run = L.cin; d.stride = L.stride; d.pad_h = d.pad_w = L.pad;
    d.ldc = L.cout; d.ld_res = L.cout; d.relu = L.relu;
    d.scratch = t_conv_scratch; d.scratch_bytes = t_conv_scratch_bytes;
    *OHo = OH; *OWo = OW;
    return mic_conv2d(&d, s);
}

struct ResnetPlan {
    int64_t pack, big, mid, scratch, total;
    int Hp, Wp, H1, W1, H2, W2;
};

ResnetPlan plan_resnet(const mic_resnet50* net, int B, int H, int W) {
    ResnetPlan p;
    const int64_t sz = (int64_t)mic_dtype_size(net->dtype);
    p.Hp = H + 6; p.Wp = W + 8 + (W & 1);  // even row length keeps 16-byte chunk alignment
    p.H1 = (H + 6 - 7) / 2 + 1; p.W1 = (W + 6 - 7) / 2 + 1;      // stem output
    p.H2 = (p.H1 + 2 - 3) / 2 + 1; p.W2 = (p.W1 + 2 - 3) / 2 + 1;  // after max-pool
    p.pack = align_up((int64_t)B * p.Hp * p.Wp * 4 * sz, 256);
    const int64_t stem = (int64_t)B * p.H1 * p.W1 * 64;
    const int64_t l1 = (int64_t)B * p.H2 * p.W2 * 256;
    p.big = align_up((stem > l1 ? stem : l1) * sz, 256);
    p.mid = align_up((int64_t)B * p.H2 * p.W2 * 128 * sz, 256);
    p.scratch = net->dtype == MIC_BF16 ? mic_conv2d_scratch_bytes() : 0;
    p.total = p.pack + 3 * p.big + 2 * p.mid + p.scratch;
    return p;
}

}  // namespace

extern "C" int mic_graph_cache_stats(int32_t* cached, int32_t* parked) {
    MIC_ARG_CHECK(cached && parked, "mic_graph_cache_stats: null pointer");
    std::lock_guard<std::mutex> lk(g_graphs.mu);
    sweep_retired();
    *cached = (int32_t)g_graphs.map.size();
    *parked = (int32_t)g_graphs.retired.size();
    return 0;
}

extern "C" int64_t mic_resnet50_workspace_bytes(const mic_resnet50* net, int B, int H, int W) {
    if (!net || B <= 0 || H < 32 || W < 32) return -1;
    return plan_resnet(net, B, H, W).total;
}

extern "C" int mic_resnet50_forward(const mic_resnet50* net, const float* images, int B, int H, int W, void* workspace,
                                    int64_t workspace_bytes, void* out, void* stream) {
    MIC_ARG_CHECK(net && images && workspace && out, "mic_resnet50_forward: null pointer");
    MIC_ARG_CHECK(B > 0 && H >= 32 && W >= 32, "mic_resnet50_forward: bad image size %dx%d (B=%d)", H, W, B);
    MIC_ARG_CHECK(net->dtype == MIC_F32 || net->dtype == MIC_BF16, "mic_resnet50_forward: bad dtype");
    MIC_ARG_CHECK(((uintptr_t)workspace & 255) == 0, "mic_resnet50_forward: workspace must be 256-byte aligned");
    const ResnetPlan p = plan_resnet(net, B, H, W);
    MIC_ARG_CHECK(workspace_bytes >= p.total, "mic_resnet50_forward: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)p.total);
    hipStream_t s = (hipStream_t)stream;
    const int dt = net->dtype;
    char* base = (char*)workspace;
    void* pack = base;
    void* X[2] = {base + p.pack, base + p.pack + p.big};
    void* DS = base + p.pack + 2 * p.big;
    void* T1 = base + p.pack + 3 * p.big;             // first-convolution output of the running block
    void* T2 = base + p.pack + 3 * p.big + p.mid;     // 3x3 output, or the next block's first-convolution output
    // split-K scratch of the late layers: its tickets start every forward pass at zero
    struct ScratchScope {
        ScratchScope(void* p, int64_t n) { t_conv_scratch = p; t_conv_scratch_bytes = n; }
        ~ScratchScope() { t_conv_scratch = nullptr; t_conv_scratch_bytes = 0; }
    } scratch_scope(p.scratch ? base + p.pack + 3 * p.big + 2 * p.mid : nullptr, p.scratch);
    if (p.scratch) MIC_HIP_CHECK(hipMemsetAsync(t_conv_scratch, 0, 4096, s));

    // stem: image -> zero-bordered 4-channel pixels; each of the 7 kernel rows is one run of
    // 8 pixels x 4 channels = 32 contiguous elements (the 8th pixel and 4th channel meet zero weights)
    MIC_TRY(mic_pack_image(images, pack, dt, B, H, W, 3, 3, 3, 5 + (W & 1), s));
    // every launch from here to the end of the trunk is a convolution on the matrix cores: one pair of profiling events
    MicProfScope prof_scope(s, dt == MIC_BF16 ? 1 : 0);
    bool stem_t1 = false;  // T1 already holds the first bottleneck's first convolution
    static const bool no_stem_pool = getenv("MIC_NO_STEM_POOL") != nullptr;  // comparison runs: convolution and pooling as two launches
    {
        const mic_conv_layer& L = net->stem;
        MIC_ARG_CHECK(L.w && L.cout == 64, "mic_resnet50_forward: stem layer not set");
        if (!no_stem_pool && mic_stem_pool_supported(dt) && L.relu) {
            // the first bottleneck's 1x1 convolution takes the pooled tile while it is on chip
            const mic_conv_layer& c1 = net->conv[0][0];
            if (c1.w && c1.kh == 1 && c1.kw == 1 && c1.stride == 1 && c1.pad == 0 && c1.cin == 64 && c1.cout == 64 && c1.relu) {
                MIC_TRY(mic_stem_pool_conv1(pack, L.w, L.bias, X[0], c1.w, c1.bias, T1, dt, B, p.Hp, p.Wp, p.H1, p.W1, 128, 128, s));
                stem_t1 = true;
            } else {
                MIC_TRY(mic_stem_pool(pack, L.w, L.bias, X[0], dt, B, p.Hp, p.Wp, p.H1, p.W1, 128, s));
            }
        } else {
            mic_conv_desc d{};
            d.in = pack; d.w = L.w; d.bias = L.bias; d.out = X[1];
            d.dtype = d.out_dtype = d.res_dtype = dt;
            d.M = B * p.H1 * p.W1; d.N = 64; d.K = 7 * 32;
            d.OH = p.H1; d.OW = p.W1; d.IH = p.Hp; d.IW = p.Wp;
            d.in_img_stride = (int64_t)p.Hp * p.Wp * 4; d.in_row_stride = (int64_t)p.Wp * 4; d.in_pix_stride = 4;
            d.kw_taps = 1; d.run = 32; d.stride = 2; d.pad_h = d.pad_w = 0;
            d.ldc = 64; d.relu = L.relu; d.k_valid = 7 * 7 * 3;
            MIC_TRY(mic_conv2d(&d, s));
            MIC_TRY(mic_maxpool3x3s2(X[1], X[0], dt, B, p.H1, p.W1, 64, s));
        }
    }
    int cur = 0, h = p.H2, w = p.W2, bi = 0;
    int nblocks = 0;
    for (int st = 0; st < 4; ++st) nblocks += net->blocks_per_stage[st];
    MIC_ARG_CHECK(nblocks > 0 && nblocks <= 16, "mic_resnet50_forward: %d bottlenecks (max 16)", nblocks);
    // Identity blocks of the stages whose layers are bound by HBM traffic run as one launch per block: 3x3 convolution,
    // expansion + residual + ReLU and the NEXT block's first 1x1 convolution (mic_bottleneck_fused).  `have_t1` says that
    // T1 already holds the first convolution of the block about to run (written by the previous block's launch).
    static const bool no_fusion = getenv("MIC_NO_BLOCK_FUSION") != nullptr;
    bool have_t1 = stem_t1;
    for (int st = 0; st < 4; ++st) {
        for (int k = 0; k < net->blocks_per_stage[st]; ++k, ++bi) {
            const mic_conv_layer* L = net->conv[bi];
            int h1 = h, w1 = w, h2, w2, h3, w3;
            const bool identity = L[3].cout == 0 && L[1].stride == 1 && L[1].kh == 3 && L[1].kw == 3 && L[1].pad == 1 && L[0].stride == 1 &&
                                  L[0].kh == 1 && L[2].kh == 1 && L[2].stride == 1 && L[1].cin == L[1].cout && L[2].cin == L[1].cout &&
                                  L[2].cout == 4 * L[1].cout && L[0].cout == L[1].cin && L[0].relu && L[1].relu && L[2].relu;
            if (!no_fusion && identity && mic_bottleneck_fused_supported(dt, L[1].cout, 0)) {
                const int C = L[1].cout;
                if (!have_t1) MIC_TRY(conv_layer(dt, L[0], X[cur], B, h, w, nullptr, T1, &h1, &w1, s));
                // the next block's first convolution rides along when it is a plain 1x1 over this block's output
                const mic_conv_layer* nx = bi + 1 < nblocks ? &net->conv[bi + 1][0] : nullptr;
                int C2 = 0;
                if (nx && nx->w && nx->kh == 1 && nx->kw == 1 && nx->stride == 1 && nx->pad == 0 && nx->cin == 4 * C && nx->relu &&
                    mic_bottleneck_fused_supported(dt, C, nx->cout))
                    C2 = nx->cout;
                void* dst = (bi == nblocks - 1) ? out : X[cur ^ 1];
                mic_fused_block_desc d{};
                d.t1 = T1; d.w2 = L[1].w; d.b2 = L[1].bias; d.w3 = L[2].w; d.b3 = L[2].bias; d.res = X[cur]; d.xout = dst;
                d.dtype = dt; d.B = B; d.H = h; d.W = w; d.C = C; d.C2 = C2;
                // the host pads every weight matrix to a multiple of 128 rows (mic_conv_layer contract)
                d.w2_rows = (C + 127) / 128 * 128; d.w3_rows = (4 * C + 127) / 128 * 128;
                if (C2) { d.w1n = nx->w; d.b1n = nx->bias; d.t1n = T2; d.w1n_rows = (C2 + 127) / 128 * 128; }
                MIC_TRY(mic_bottleneck_fused(&d, s));
                if (C2) { void* t = T1; T1 = T2; T2 = t; }
                have_t1 = C2 != 0;
                cur ^= 1;
                continue;
            }
            {
                // first block of a stage whose shortcut convolution has stride 1 (ResNet-50: stage 0): the same launch with
                // the shortcut as a second K segment of the expansion (the stage's fused tail matrix [W3 | W_downsample])
                const mic_fused_tail& F0 = net->tail[st];
                const mic_conv_layer* nx = bi + 1 < nblocks ? &net->conv[bi + 1][0] : nullptr;
                const bool shape = k == 0 && L[3].cout > 0 && F0.w && F0.skip_stride == 1 && L[3].stride == 1 && L[3].kh == 1 && L[3].kw == 1 &&
                                   L[3].pad == 0 && L[0].stride == 1 && L[0].kh == 1 && L[0].relu && L[1].stride == 1 && L[1].kh == 3 &&
                                   L[1].kw == 3 && L[1].pad == 1 && L[1].relu && L[1].cin == L[1].cout && L[2].kh == 1 && L[2].stride == 1 &&
                                   L[2].relu && L[2].cin == L[1].cout && L[2].cout == 4 * L[1].cout && F0.cin_main == L[1].cout &&
                                   F0.cin_skip == L[3].cin && F0.cout == L[2].cout && nx && nx->w && nx->kh == 1 && nx->kw == 1 &&
                                   nx->stride == 1 && nx->pad == 0 && nx->cin == L[2].cout && nx->relu;
                if (!no_fusion && shape && mic_bottleneck_fused_shortcut_supported(dt, L[1].cout, nx->cout, F0.cin_skip)) {
                    const int C = L[1].cout;
                    if (!have_t1) MIC_TRY(conv_layer(dt, L[0], X[cur], B, h, w, nullptr, T1, &h1, &w1, s));
                    mic_fused_block_desc d{};
                    d.t1 = T1; d.w2 = L[1].w; d.b2 = L[1].bias; d.w3 = F0.w; d.b3 = F0.bias; d.xs = X[cur]; d.CS = F0.cin_skip;
                    d.xout = (bi == nblocks - 1) ? out : X[cur ^ 1];
                    d.dtype = dt; d.B = B; d.H = h; d.W = w; d.C = C; d.C2 = nx->cout;
                    d.w2_rows = (C + 127) / 128 * 128; d.w3_rows = (4 * C + 127) / 128 * 128;
                    d.w1n = nx->w; d.b1n = nx->bias; d.t1n = T2; d.w1n_rows = (nx->cout + 127) / 128 * 128;
                    MIC_TRY(mic_bottleneck_fused(&d, s));
                    { void* t = T1; T1 = T2; T2 = t; }
                    have_t1 = true;
                    cur ^= 1;
                    continue;
                }
            }
            if (!have_t1) MIC_TRY(conv_layer(dt, L[0], X[cur], B, h, w, nullptr, T1, &h1, &w1, s));
            have_t1 = false;
            MIC_TRY(conv_layer(dt, L[1], T1, B, h1, w1, nullptr, T2, &h2, &w2, s));
            const void* idt = X[cur];
            void* dst = (bi == nblocks - 1) ? out : X[cur ^ 1];
            const mic_fused_tail& F = net->tail[st];
            if (k == 0 && L[3].cout > 0 && F.w) {
                // conv3 + shortcut convolution + ReLU as one GEMM over [T2 | strided X]
                MIC_ARG_CHECK(F.cin_main == L[2].cin && F.cin_skip == L[3].cin && F.cout == L[2].cout && F.skip_stride == L[3].stride,
                              "mic_resnet50_forward: fused tail of stage %d does not match its conv3 / downsample", st);
                mic_conv_desc d{};
                d.in = T2; d.w = F.w; d.bias = F.bias; d.out = dst;
                d.dtype = d.out_dtype = d.res_dtype = dt;
                d.M = B * h2 * w2; d.N = F.cout; d.K = F.cin_main;
                d.OH = h2; d.OW = w2; d.IH = h2; d.IW = w2;
                d.in_img_stride = (int64_t)h2 * w2 * F.cin_main; d.in_row_stride = (int64_t)w2 * F.cin_main; d.in_pix_stride = F.cin_main;
                d.kw_taps = 1; d.run = F.cin_main; d.stride = 1;
                d.ldc = F.cout; d.relu = 1;
                d.scratch = t_conv_scratch; d.scratch_bytes = t_conv_scratch_bytes;
                d.in2 = X[cur]; d.K2 = F.cin_skip; d.stride2 = F.skip_stride;
                d.in2_img_stride = (int64_t)h * w * F.cin_skip; d.in2_row_stride = (int64_t)w * F.cin_skip; d.in2_pix_stride = F.cin_skip;
                MIC_TRY(mic_conv2d(&d, s));
                cur ^= 1; h = h2; w = w2;
                continue;
            }
            if (L[3].cout > 0) {
                int hd, wd;
                MIC_TRY(conv_layer(dt, L[3], X[cur], B, h, w, nullptr, DS, &hd, &wd, s));
                idt = DS;
            }
            MIC_TRY(conv_layer(dt, L[2], T2, B, h2, w2, idt, dst, &h3, &w3, s));
            cur ^= 1; h = h3; w = w3;
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// transformer decoder
// ---------------------------------------------------------------------------------------------
namespace {
// y = act(in W^T + b): skinny weight-streaming kernel for small row counts when a fragment-ordered
// copy of the weights exists, tiled implicit-GEMM kernel otherwise
int proj(int dt, const void* wf, const void* in, int ld_in, int N_rows, int K, const void* w, const float* b, void* out, int out_dt,
         int ldc, int N_cols, int relu, hipStream_t s) {
    if (wf && N_rows <= 2048) return mic_linear_skinny(in, ld_in, wf, b, out, out_dt, ldc, dt, N_rows, N_cols, K, relu, s);
    return linear(dt, in, ld_in, N_rows, K, w, b, nullptr, 0, 0, out, out_dt, ldc, N_cols, relu, s);
}

// x <- LayerNorm(in W^T + b + x): one fused launch when the hidden size has a fused variant,
// otherwise GEMM (+bias +residual) into y followed by the LayerNorm kernel
int res_ln(int dt, const void* wf, const void* in, int ld_in, int K, const void* w, const float* b, const float* g, const float* be,
           const mic_tr_workspace* ws, int N, int H, hipStream_t s) {
    if (wf) return mic_linear_res_ln(in, ld_in, wf, b, ws->x, H, g, be, ws->x, ws->xt, H, dt, N, K, H, s);
    MIC_TRY(linear(dt, in, ld_in, N, K, w, b, ws->x, MIC_F32, H, ws->y, MIC_F32, H, H, 0, s));
    return mic_layernorm(ws->y, g, be, ws->x, ws->xt, H, dt, N, H, s);
}

// the folded cross-attention sub-layer runs when the decoder carries the folded weights of layer l, the
// workspace the folded K/V buffers, and the shape is one the kernel takes
bool use_fold(const mic_tr_decoder* dec, const mic_tr_workspace* ws, int l) {
    static const bool off = getenv("MIC_NO_CROSS_FOLD") != nullptr;
    const mic_tr_layer& L = dec->layer[l];
    if (off || dec->heads > MIC_FOLD_MAX_HEADS || !L.cross_qb_scaled || !ws->cross_kf[l] || !ws->cross_vf[l] || !ws->cross_c[l]) return false;
    for (int h = 0; h < dec->heads; ++h)
        if (!L.cross_kf_w[h] || !L.cross_vf_w[h]) return false;
    const int epr = 64 / (int)mic_dtype_size(dec->dtype);
    return (dec->H / dec->heads) % epr == 0 && mic_cross_attn_folded_supported(dec->dtype, dec->H, dec->heads, ws->S, ws->rows_per_image) != 0;
}

int check_tr(const mic_tr_decoder* dec, const mic_tr_workspace* ws) {
    MIC_ARG_CHECK(dec && ws, "transformer driver: null pointer");
    MIC_ARG_CHECK(dec->dtype == MIC_F32 || dec->dtype == MIC_BF16, "transformer driver: bad dtype");
    MIC_ARG_CHECK(dec->layers >= 1 && dec->layers <= MIC_MAX_LAYERS, "transformer driver: %d layers (max %d)", dec->layers, MIC_MAX_LAYERS);
    const int epr = 64 / (int)mic_dtype_size(dec->dtype);
    MIC_ARG_CHECK(dec->H > 0 && dec->H % epr == 0, "transformer driver: hidden size %d must be a multiple of %d", dec->H, epr);
    MIC_ARG_CHECK(dec->heads > 0 && dec->H % dec->heads == 0, "transformer driver: heads=%d", dec->heads);
    MIC_ARG_CHECK(ws->N > 0 && ws->t_cap > 0 && ws->S > 0 && ws->rows_per_image > 0 && ws->N % ws->rows_per_image == 0,
                  "transformer driver: bad workspace geometry");
    MIC_ARG_CHECK(ws->x && ws->xt && ws->y && ((ws->qkv && ws->o) || ws->qkvf) && ws->ffh && ws->logits, "transformer driver: workspace buffers missing");
    return 0;
}
}  // namespace

namespace {
bool chain_shape_ok(const mic_tr_decoder* dec, const mic_tr_workspace* ws);
bool chain_ok(const mic_tr_decoder* dec, const mic_tr_workspace* ws);
}  // namespace

// Only the chained steps read the folded keys / values; the plain step (more rows than the chained path takes) reads k and v.
// A workspace whose shape allows chaining but that has no statistics buffers may still be stepped by mic_tr_step_static,
// which brings its own: it gets both forms.
extern "C" int mic_tr_cross_kv(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const void* feats, void* stream) {
    MIC_TRY(check_tr(dec, ws));
    MIC_ARG_CHECK(feats, "mic_tr_cross_kv: null features");
    hipStream_t s = (hipStream_t)stream;
    const int H = dec->H, rows = ws->N / ws->rows_per_image * ws->S;
    MicProfScope prof_scope(s, dec->dtype == MIC_BF16 ? 1 : 0);  // a run of GEMMs (and one small bias kernel per layer)
    static const bool no_combine = getenv("MIC_NO_CROSS_COMBINE") != nullptr;  // comparison runs: project k, v, then fold
    for (int l = 0; l < dec->layers; ++l) {
        const mic_tr_layer& L = dec->layer[l];
        const bool fold = use_fold(dec, ws, l) && chain_shape_ok(dec, ws);
        if (fold && chain_ok(dec, ws) && !no_combine && L.cross_kfc_w && L.cross_vfc_w && L.cross_kfc_b && L.cross_vfc_b && L.cross_cu && L.cross_c0) {
            // folded keys and values straight from the image tokens: [rows][heads][H] each, then the per-position constant
            const int heads = dec->heads;
            // FLOP accounting: each stands for the H x H key (value) projection of the unfolded layer, not for its heads-fold width
            MIC_TRY(linear(dec->dtype, feats, H, rows, H, L.cross_kfc_w, L.cross_kfc_b, nullptr, 0, 0, ws->cross_kf[l], dec->dtype,
                           heads * H, heads * H, 0, s, H / heads));
            MIC_TRY(linear(dec->dtype, feats, H, rows, H, L.cross_vfc_w, L.cross_vfc_b, nullptr, 0, 0, ws->cross_vf[l], dec->dtype,
                           heads * H, heads * H, 0, s, H / heads));
            MIC_TRY(mic_cross_fold_c(feats, L.cross_cu, L.cross_c0, ws->cross_c[l], dec->dtype, rows, H, heads, s));
            continue;
        }
        MIC_ARG_CHECK(L.cross_kv_w && ws->cross_kv[l], "mic_tr_cross_kv: layer %d pointers missing", l);
        MIC_TRY(linear(dec->dtype, feats, H, rows, H, L.cross_kv_w, L.cross_kv_b, nullptr, 0, 0, ws->cross_kv[l], dec->dtype, 2 * H,
                       2 * H, 0, s));
        if (fold) {
            // K'_h = k_h (W_q,h / sqrt d), V'_h = v_h W_o,h^T as per-head GEMMs over the rows of kv; c_h = b_q,h . k_h
            const int heads = dec->heads, hd = H / heads;
            const size_t sz = mic_dtype_size(dec->dtype);
            for (int h = 0; h < heads; ++h) {
                const char* kin = (const char*)ws->cross_kv[l] + (size_t)h * hd * sz;
                const char* vin = kin + (size_t)H * sz;
                MIC_TRY(linear(dec->dtype, kin, 2 * H, rows, hd, L.cross_kf_w[h], nullptr, nullptr, 0, 0,
                               (char*)ws->cross_kf[l] + (size_t)h * H * sz, dec->dtype, heads * H, H, 0, s));
                MIC_TRY(linear(dec->dtype, vin, 2 * H, rows, hd, L.cross_vf_w[h], nullptr, nullptr, 0, 0,
                               (char*)ws->cross_vf[l] + (size_t)h * H * sz, dec->dtype, heads * H, H, 0, s));
            }
            MIC_TRY(mic_cross_fold_bias(ws->cross_kv[l], L.cross_qb_scaled, ws->cross_c[l], dec->dtype, rows, H, heads, s));
        }
    }
    return 0;
}

namespace {
mic_ln_ref ln_view(const float* raw, const float* stats, const float* g, const float* b, int H) {
    mic_ln_ref v{};
    v.raw = raw; v.stats = stats; v.gamma = g; v.beta = b; v.ld = H; v.parts = (H + 127) / 128; v.width = H;
    return v;
}

// One decoder step with every LayerNorm folded into its neighbours (mic_linear_skinny_ex): the
// sub-layer tails store the pre-norm sum y and its row statistics, the next projection and the next
// residual add normalise on load.  8 launches per layer, none narrower than (N/16) x (H/128) workgroups.
int tr_step_chained(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const int32_t* tokens, int ld_tok, const int32_t* anc, int t,
                    int attn_mode, bool skip_embed, hipStream_t s) {
    const int dt = dec->dtype, H = dec->H, N = ws->N, S = ws->S, L = dec->layers;
    if (!skip_embed) MIC_TRY(mic_embed_pos(tokens, ld_tok, t, dec->tok_emb, dec->pos_emb, ws->x, ws->xt, H, dt, N, H, dec->vocab, s));
    bool plain = true;  // x is ws->x / ws->xt; otherwise x = LN(ws->y; ws->stats[cs]) with (pg, pb)
    int cs = 0;
    const float *pg = nullptr, *pb = nullptr;
    auto base = [&](const void* wf, const float* bias, int K, int Ncols) {
        mic_skinny_desc d{};
        d.wf = wf; d.bias = bias; d.dtype = dt; d.M = N; d.N = Ncols; d.K = K;
        return d;
    };
    auto set_x = [&](mic_skinny_desc& d) {  // input = current x
        if (plain) { d.x = ws->xt; d.ld_x = H; }
        else d.x_ln = ln_view(ws->y, ws->stats[cs], pg, pb, H);
    };
    auto tail = [&](const void* in, int ld_in, int K, const void* wf, const float* bias, const float* g, const float* b) -> int {
        // y <- in W^T + bias + x ; statistics of y ; afterwards x = LN(y) with (g, b)
        mic_skinny_desc d = base(wf, bias, K, H);
        d.x = in; d.ld_x = ld_in;
        if (plain) { d.res = ws->x; d.ld_res = H; }
        else d.res_ln = ln_view(ws->y, ws->stats[cs], pg, pb, H);
        const int ns = plain ? 0 : cs ^ 1;
        d.out = ws->y; d.out_dtype = MIC_F32; d.ldc = H; d.stats_out = ws->stats[ns];
        MIC_TRY(mic_linear_skinny_ex(&d, s));
        plain = false; cs = ns; pg = g; pb = b;
        return 0;
    };
    for (int l = 0; l < L; ++l) {
        const mic_tr_layer& Y = dec->layer[l];
        {
            mic_skinny_desc d = base(Y.self_in_wf, Y.self_in_b, H, 3 * H);
            set_x(d);
            d.out = ws->qkv; d.out_dtype = MIC_F32; d.ldc = 3 * H;
            MIC_TRY(mic_linear_skinny_ex(&d, s));
        }
        MIC_TRY(mic_self_attn_step(ws->qkv, ws->kcache[l], ws->vcache[l], anc, t, ws->t_cap, ws->o, H, dt, N, H, dec->heads, s));
        MIC_TRY(tail(ws->o, H, H, Y.self_out_wf, Y.self_out_b, Y.ln1_g, Y.ln1_b));
        float* ap = nullptr;
        int ld_attn = 0, accumulate = 0;
        float scale = 1.f;
        if (attn_mode == 1) {
            ap = ws->attn + (int64_t)t * S; ld_attn = ws->t_cap * S; scale = 1.f / (float)L; accumulate = l > 0;
        } else if (attn_mode == 2) {
            ap = ws->attn + ((int64_t)l * ws->t_cap + t) * N * S; ld_attn = S;
        }
        if (use_fold(dec, ws, l)) {
            // query projection, attention, output projection, residual and statistics in one launch
            const mic_ln_ref xv = ln_view(ws->y, ws->stats[cs], pg, pb, H);
            MIC_TRY(mic_cross_attn_folded(&xv, ws->cross_kf[l], ws->cross_vf[l], ws->cross_c[l], Y.cross_out_b, S, ws->rows_per_image,
                                          ws->y, ws->stats[cs ^ 1], ap, ld_attn, scale, accumulate, dt, N, H, dec->heads, s));
            cs ^= 1; pg = Y.ln2_g; pb = Y.ln2_b;
        } else {
            {
                mic_skinny_desc d = base(Y.cross_q_wf, Y.cross_q_b, H, H);
                set_x(d);
                d.out = ws->qkv; d.out_dtype = MIC_F32; d.ldc = H;
                MIC_TRY(mic_linear_skinny_ex(&d, s));
            }
            MIC_TRY(mic_cross_attn_step(ws->qkv, H, ws->cross_kv[l], S, ws->rows_per_image, ws->o, H, ap, ld_attn, scale, accumulate, dt,
                                        N, H, dec->heads, s));
            MIC_TRY(tail(ws->o, H, H, Y.cross_out_wf, Y.cross_out_b, Y.ln2_g, Y.ln2_b));
        }
        {
            mic_skinny_desc d = base(Y.ff1_wf, Y.ff1_b, H, 2 * H);
            set_x(d);
            d.out = ws->ffh; d.out_dtype = dt; d.ldc = 2 * H; d.relu = 1;
            MIC_TRY(mic_linear_skinny_ex(&d, s));
        }
        MIC_TRY(tail(ws->ffh, 2 * H, 2 * H, Y.ff2_wf, Y.ff2_b, Y.ln3_g, Y.ln3_b));
    }
    mic_skinny_desc d = base(dec->out_wf, dec->out_b, H, dec->vocab);
    set_x(d);
    d.out = ws->logits; d.out_dtype = MIC_F32; d.ldc = dec->vocab;
    return mic_linear_skinny_ex(&d, s);
}

// One decoder step with both attention sub-layers of a layer in one launch (mic_attn_fused): the self-attention
// output projection is folded into the cached values, so a layer is 4 launches (packed in-projection, attention,
// two feed-forward GEMMs) instead of 8, and the LayerNorms ride on their neighbours as in tr_step_chained.
int tr_step_fused(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const int32_t* tokens, int ld_tok, const int32_t* anc, int t,
                  int attn_mode, bool skip_embed, hipStream_t s) {
    const int dt = dec->dtype, H = dec->H, N = ws->N, S = ws->S, L = dec->layers, QW = (2 + dec->heads) * H;
    if (!skip_embed) MIC_TRY(mic_embed_pos(tokens, ld_tok, t, dec->tok_emb, dec->pos_emb, ws->x, ws->xt, H, dt, N, H, dec->vocab, s));
    bool plain = true;  // x is ws->x / ws->xt; otherwise x = LN(ws->y; ws->stats[cs]) with (pg, pb)
    int cs = 0;
    const float *pg = nullptr, *pb = nullptr;
    auto base = [&](const void* wf, const float* bias, int K, int Ncols) {
        mic_skinny_desc d{};
        d.wf = wf; d.bias = bias; d.dtype = dt; d.M = N; d.N = Ncols; d.K = K;
        return d;
    };
    auto set_x = [&](mic_skinny_desc& d) {
        if (plain) { d.x = ws->xt; d.ld_x = H; }
        else d.x_ln = ln_view(ws->y, ws->stats[cs], pg, pb, H);
    };
    for (int l = 0; l < L; ++l) {
        const mic_tr_layer& Y = dec->layer[l];
        {
            mic_skinny_desc d = base(Y.self_qkvf_wf, Y.self_qkvf_b, H, QW);
            set_x(d);
            d.out = ws->qkvf; d.out_dtype = MIC_F32; d.ldc = QW;
            MIC_TRY(mic_linear_skinny_ex(&d, s));
        }
        {
            mic_attn_fused_desc f{};
            f.qkvf = ws->qkvf; f.kcache = ws->kcache[l]; f.vfcache = ws->vfcache[l]; f.anc = anc;
            if (plain) f.x = ws->x;
            else f.x_ln = ln_view(ws->y, ws->stats[cs], pg, pb, H);
            f.self_b = Y.self_of_b; f.ln1_g = Y.ln1_g; f.ln1_b = Y.ln1_b;
            f.kf = ws->cross_kf[l]; f.vf = ws->cross_vf[l]; f.c = ws->cross_c[l]; f.bo = Y.cross_out_b;
            const int ns = plain ? 0 : cs ^ 1;
            f.y = ws->y; f.stats_out = ws->stats[ns];
            f.attn_scale = 1.f;
            if (attn_mode == 1) {
                f.attn_out = ws->attn + (int64_t)t * S; f.ld_attn = ws->t_cap * S; f.attn_scale = 1.f / (float)L; f.accumulate = l > 0;
            } else if (attn_mode == 2) {
                f.attn_out = ws->attn + ((int64_t)l * ws->t_cap + t) * N * S; f.ld_attn = S;
            }
            f.dtype = dt; f.N = N; f.H = H; f.heads = dec->heads; f.S = S; f.rows_per_image = ws->rows_per_image; f.t = t; f.t_cap = ws->t_cap;
            MIC_TRY(mic_attn_fused(&f, s));
            plain = false; cs = ns; pg = Y.ln2_g; pb = Y.ln2_b;
        }
        {
            mic_skinny_desc d = base(Y.ff1_wf, Y.ff1_b, H, 2 * H);
            set_x(d);
            d.out = ws->ffh; d.out_dtype = dt; d.ldc = 2 * H; d.relu = 1;
            MIC_TRY(mic_linear_skinny_ex(&d, s));
        }
        {
            mic_skinny_desc d = base(Y.ff2_wf, Y.ff2_b, 2 * H, H);
            d.x = ws->ffh; d.ld_x = 2 * H;
            d.res_ln = ln_view(ws->y, ws->stats[cs], pg, pb, H);
            d.out = ws->y; d.out_dtype = MIC_F32; d.ldc = H; d.stats_out = ws->stats[cs ^ 1];
            MIC_TRY(mic_linear_skinny_ex(&d, s));
            cs ^= 1; pg = Y.ln3_g; pb = Y.ln3_b;
        }
    }
    mic_skinny_desc d = base(dec->out_wf, dec->out_b, H, dec->vocab);
    set_x(d);
    d.out = ws->logits; d.out_dtype = MIC_F32; d.ldc = dec->vocab;
    return mic_linear_skinny_ex(&d, s);
}

int chain_max_rows() {
    static const int v = [] { const char* e = getenv("MIC_CHAIN_MAX_ROWS"); return e ? atoi(e) : 2048; }();
    return v;
}

// the chained step's conditions on the decoder and the row count (the workspace's statistics buffers aside)
bool chain_shape_ok(const mic_tr_decoder* dec, const mic_tr_workspace* ws) {
    const int H = dec->H;
    bool chain = ws->N <= chain_max_rows() && dec->out_wf && H % 4 == 0 && H <= 1024;
    for (int l = 0; l < dec->layers && chain; ++l) {
        const mic_tr_layer& Y = dec->layer[l];
        chain = Y.self_in_wf && Y.self_out_wf && Y.cross_q_wf && Y.cross_out_wf && Y.ff1_wf && Y.ff2_wf;
    }
    return chain;
}

bool chain_ok(const mic_tr_decoder* dec, const mic_tr_workspace* ws) { return ws->stats[0] && ws->stats[1] && chain_shape_ok(dec, ws); }
}  // namespace

// Whether a step on (dec, ws) can take the fused-attention path, the temporary ws->qkvf aside (mic_tr_step_static
// provides that one itself).
bool mic_tr_fused_eligible(const mic_tr_decoder* dec, const mic_tr_workspace* ws) {
    static const bool off = getenv("MIC_NO_ATTN_FUSE") != nullptr;
    if (off || !chain_ok(dec, ws)) return false;
    if (!mic_attn_fused_supported(dec->dtype, dec->H, dec->heads, ws->S, ws->rows_per_image, ws->t_cap)) return false;
    for (int l = 0; l < dec->layers; ++l) {
        const mic_tr_layer& Y = dec->layer[l];
        if (!Y.self_qkvf_wf || !Y.self_qkvf_b || !Y.self_of_b || !ws->vfcache[l] || !use_fold(dec, ws, l)) return false;
    }
    return true;
}

int mic_tr_step_impl(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const int32_t* tokens, int ld_tok, const int32_t* anc, int t,
                     int attn_mode, bool skip_embed, void* stream, bool allow_persist = false, const mic_beam_tail_desc* tail = nullptr,
                     bool* tail_done = nullptr);

extern "C" int mic_tr_step(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const int32_t* tokens, int ld_tok,
                           const int32_t* anc, int t, int attn_mode, void* stream) {
    return mic_tr_step_impl(dec, ws, tokens, ld_tok, anc, t, attn_mode, false, stream);
}

extern "C" int mic_tr_step_persistent(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const int32_t* tokens, int ld_tok,
                                      const int32_t* anc, int t, int attn_mode, void* stream) {
    MIC_TRY(check_tr(dec, ws));
    MIC_ARG_CHECK(mic_tr_persist_eligible(dec, ws), "mic_tr_step_persistent: this decoder / workspace cannot take the one-launch step "
                  "(folded weights, qkvf, vfcache, stats and sync buffers, and a shape mic_tr_persist_supported accepts)");
    return mic_tr_step_impl(dec, ws, tokens, ld_tok, anc, t, attn_mode, false, stream, true);
}

// which schedule a step on (dec, ws) takes inside the *_steps drivers: 3 one launch, 2 fused attention, 1 chained, 0 plain
extern "C" int mic_tr_step_schedule(const mic_tr_decoder* dec, const mic_tr_workspace* ws) {
    if (!dec || !ws) return -1;
    if (mic_tr_persist_eligible(dec, ws)) return 3;
    if (ws->qkvf && mic_tr_fused_eligible(dec, ws)) return 2;
    return chain_ok(dec, ws) ? 1 : 0;
}

int mic_tr_step_impl(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const int32_t* tokens, int ld_tok, const int32_t* anc, int t,
                     int attn_mode, bool skip_embed, void* stream, bool allow_persist, const mic_beam_tail_desc* tail, bool* tail_done) {
    MIC_TRY(check_tr(dec, ws));
    MIC_ARG_CHECK(tokens && t >= 0 && t < ws->t_cap && t < dec->pos_rows, "mic_tr_step: position %d outside t_cap=%d / table=%d", t,
                  ws->t_cap, dec->pos_rows);
    MIC_ARG_CHECK(attn_mode == 0 || ws->attn, "mic_tr_step: attention requested without a buffer");
    hipStream_t s = (hipStream_t)stream;
    const int dt = dec->dtype, H = dec->H, N = ws->N, S = ws->S, L = dec->layers;
    if (allow_persist && mic_tr_persist_eligible(dec, ws)) {
        if (!skip_embed) MIC_TRY(mic_embed_pos(tokens, ld_tok, t, dec->tok_emb, dec->pos_emb, ws->x, ws->xt, H, dt, N, H, dec->vocab, s));
        // the beam-search tail rides along as the launch's last phase where it fits
        const bool with_tail = tail && mic_tr_persist_tail_ok(dec, ws, tail->K, tail->V);
        if (tail_done) *tail_done = with_tail;
        return mic_tr_persist_launch(dec, ws, anc, t, attn_mode, with_tail ? tail : nullptr, s);
    }
    if (ws->qkvf && mic_tr_fused_eligible(dec, ws)) return tr_step_fused(dec, ws, tokens, ld_tok, anc, t, attn_mode, skip_embed, s);
    MIC_ARG_CHECK(ws->qkv && ws->o, "mic_tr_step: this shape does not take the fused-attention step and needs the qkv and o buffers");
    if (chain_ok(dec, ws)) return tr_step_chained(dec, ws, tokens, ld_tok, anc, t, attn_mode, skip_embed, s);
    const bool fused = mic_linear_res_ln_supported(H) != 0;
    if (!skip_embed) MIC_TRY(mic_embed_pos(tokens, ld_tok, t, dec->tok_emb, dec->pos_emb, ws->x, ws->xt, H, dt, N, H, dec->vocab, s));
    for (int l = 0; l < L; ++l) {
        const mic_tr_layer& Y = dec->layer[l];
        // masked self-attention over the cache + residual + LayerNorm
        MIC_TRY(proj(dt, Y.self_in_wf, ws->xt, H, N, H, Y.self_in_w, Y.self_in_b, ws->qkv, MIC_F32, 3 * H, 3 * H, 0, s));
        MIC_TRY(mic_self_attn_step(ws->qkv, ws->kcache[l], ws->vcache[l], anc, t, ws->t_cap, ws->o, H, dt, N, H, dec->heads, s));
        MIC_TRY(res_ln(dt, fused ? Y.self_out_wf : nullptr, ws->o, H, H, Y.self_out_w, Y.self_out_b, Y.ln1_g, Y.ln1_b, ws, N, H, s));
        // cross-attention against the image's cached K/V + residual + LayerNorm
        MIC_TRY(proj(dt, Y.cross_q_wf, ws->xt, H, N, H, Y.cross_q_w, Y.cross_q_b, ws->qkv, MIC_F32, H, H, 0, s));
        float* ap = nullptr;
        int ld_attn = 0, accumulate = 0;
        float scale = 1.f;
        if (attn_mode == 1) {
            ap = ws->attn + (int64_t)t * S; ld_attn = ws->t_cap * S; scale = 1.f / (float)L; accumulate = l > 0;
        } else if (attn_mode == 2) {
            ap = ws->attn + ((int64_t)l * ws->t_cap + t) * N * S; ld_attn = S;
        }
        MIC_TRY(mic_cross_attn_step(ws->qkv, H, ws->cross_kv[l], S, ws->rows_per_image, ws->o, H, ap, ld_attn, scale, accumulate, dt,
                                    N, H, dec->heads, s));
        MIC_TRY(res_ln(dt, fused ? Y.cross_out_wf : nullptr, ws->o, H, H, Y.cross_out_w, Y.cross_out_b, Y.ln2_g, Y.ln2_b, ws, N, H, s));
        // feed-forward + residual + LayerNorm
        MIC_TRY(proj(dt, Y.ff1_wf, ws->xt, H, N, H, Y.ff1_w, Y.ff1_b, ws->ffh, dt, 2 * H, 2 * H, 1, s));
        MIC_TRY(res_ln(dt, fused ? Y.ff2_wf : nullptr, ws->ffh, 2 * H, 2 * H, Y.ff2_w, Y.ff2_b, Y.ln3_g, Y.ln3_b, ws, N, H, s));
    }
    return proj(dt, dec->out_wf, ws->xt, H, N, H, dec->out_w, dec->out_b, ws->logits, MIC_F32, dec->vocab, dec->vocab, 0, s);
}

extern "C" int mic_tr_beam_steps(const mic_tr_decoder* dec, const mic_tr_workspace* ws, const mic_beam_state st[2], int32_t* parent,
                                 int32_t* live_images, const double* sqrt_tab, int B, int K, int t0, int t1, int eos_id,
                                 int want_attn, int use_graph, void* stream) {
    MIC_TRY(check_tr(dec, ws));
    MIC_ARG_CHECK(st && parent && live_images && sqrt_tab && ws->top_val && ws->top_idx, "mic_tr_beam_steps: null pointer");
    MIC_ARG_CHECK(B * K == ws->N && ws->rows_per_image == K, "mic_tr_beam_steps: workspace rows %d != B*K = %d*%d", ws->N, B, K);
    MIC_ARG_CHECK(t0 >= 0 && t1 >= t0 && t1 + 1 <= ws->t_cap, "mic_tr_beam_steps: steps [%d,%d) do not fit t_cap=%d", t0, t1, ws->t_cap);
    MIC_ARG_CHECK(t1 <= dec->pos_rows, "mic_tr_beam_steps: steps [%d,%d) reach past the position table (%d rows)", t0, t1, dec->pos_rows);
    // log-softmax + top-K, selection and the next step's embedding as one launch per image where the shape allows
    static const bool no_tail = getenv("MIC_NO_BEAM_TAIL") != nullptr;
    const bool tail = !no_tail && mic_beam_tail_supported(K, dec->vocab) && dec->H % 4 == 0;
    auto body = [&](void* stream) -> int {
        for (int t = t0; t < t1; ++t) {
            const mic_beam_state* cur = &st[t & 1];
            const mic_beam_state* nxt = &st[(t + 1) & 1];
            if (tail) {
                mic_beam_tail_desc d{};
                d.logits = ws->logits; d.ld = dec->vocab; d.V = dec->vocab; d.cur = cur; d.nxt = nxt; d.parent = parent;
                d.live_images = live_images + t; d.sqrt_tab = sqrt_tab;
                d.B = B; d.K = K; d.n_live_slots = t == 0 ? 1 : K; d.t = t; d.t_cap = ws->t_cap; d.eos_id = eos_id;
                d.tok_emb = t + 1 < t1 ? dec->tok_emb : nullptr; d.pos_emb = dec->pos_emb; d.x = ws->x; d.xt = ws->xt; d.ld_xt = dec->H;
                d.dtype = dec->dtype; d.H = dec->H; d.vocab = dec->vocab;
                // the grouping of the one-launch step's tail phase, whichever schedule runs: every schedule rounds alike
                d.row_group_threads = K >= 3 ? 128 : 0;
                bool done = false;
                MIC_TRY(mic_tr_step_impl(dec, ws, cur->seq, ws->t_cap, cur->anc, t, want_attn ? 1 : 0, t > t0, stream, true, &d, &done));
                if (!done) MIC_TRY(mic_beam_tail_ex(&d, stream));
                continue;
            }
            MIC_TRY(mic_tr_step_impl(dec, ws, cur->seq, ws->t_cap, cur->anc, t, want_attn ? 1 : 0, false, stream, true));
            MIC_TRY(mic_logsoftmax_topk(ws->logits, dec->vocab, ws->N, dec->vocab, K, 1.0f, 0, ws->top_val, ws->top_idx, nullptr, stream));
            MIC_TRY(mic_beam_select(cur, nxt, ws->top_val, ws->top_idx, parent, live_images + t, sqrt_tab, B, K, t == 0 ? 1 : K, t,
                                    ws->t_cap, eos_id, 0, stream));
        }
        return 0;
    };
    if (!use_graph || t1 == t0) return body(stream);
    return run_graphed(make_key("beam", *dec, *ws, st[0], st[1], parent, live_images, sqrt_tab, B, K, t0, t1, eos_id, want_attn), (hipStream_t)stream, body);
}

extern "C" int mic_tr_sample_steps(const mic_tr_decoder* dec, const mic_tr_workspace* ws, int32_t* tokens, uint8_t* finished,
                                   float* total, int32_t* live_rows, float inv_temp, const float* u, int t0, int t1, int pad_id,
                                   int eos_id, int want_attn, int use_graph, void* stream) {
    MIC_TRY(check_tr(dec, ws));
    MIC_ARG_CHECK(tokens && finished && total && live_rows && ws->top_val && ws->top_idx, "mic_tr_sample_steps: null pointer");
    MIC_ARG_CHECK(ws->rows_per_image == 1, "mic_tr_sample_steps: one row per image expected");
    MIC_ARG_CHECK(t0 >= 0 && t1 >= t0 && t1 + 1 <= ws->t_cap, "mic_tr_sample_steps: steps [%d,%d) do not fit t_cap=%d", t0, t1, ws->t_cap);
    auto body = [&](void* stream) -> int {
        for (int t = t0; t < t1; ++t) {
            MIC_TRY(mic_tr_step_impl(dec, ws, tokens, ws->t_cap, nullptr, t, want_attn ? 2 : 0, false, stream, true));
            if (u)
                MIC_TRY(mic_sample_multinomial(ws->logits, dec->vocab, ws->N, dec->vocab, inv_temp, u + (int64_t)t * ws->N, ws->top_val,
                                               ws->top_idx, stream));
            else
                MIC_TRY(mic_logsoftmax_topk(ws->logits, dec->vocab, ws->N, dec->vocab, 1, 1.0f, 0, ws->top_val, ws->top_idx, nullptr, stream));
            MIC_TRY(mic_sample_update(ws->top_val, ws->top_idx, tokens, ws->t_cap, t + 1, finished, total, live_rows + t, ws->N, pad_id,
                                      eos_id, stream));
        }
        return 0;
    };
    if (!use_graph || t1 == t0) return body(stream);
    return run_graphed(make_key("samp", *dec, *ws, tokens, finished, total, live_rows, inv_temp, u, t0, t1, pad_id, eos_id, want_attn), (hipStream_t)stream, body);
}

// ---------------------------------------------------------------------------------------------
// LSTM decoders
// ---------------------------------------------------------------------------------------------
namespace {
// skip_embed: the input half of xh[0] already holds the embedding of this step's tokens (written by the previous step's tail)
// whether layer l's cell can run in the epilogue of its gate GEMM: unit-major weights, a staging row for h' and the
// weight-streaming GEMM's row limit.  The top layer of the intermediate model then takes the LayerNorm as a launch of its
// own, which is shorter than the cell + LayerNorm launch it replaces (that one reads the whole gate matrix back).
bool lstm_cell_fusable(const mic_lstm_decoder* dec, const mic_lstm_workspace* ws, int l) {
    static const bool off = getenv("MIC_NO_CELL_FUSE") != nullptr;  // comparison runs
    return !off && dec->wf_cell[l] && dec->b_cell[l] && ws->hstage[l] && ws->N <= 2048;
}

// fuse_cell: the caller moves h' from ws->hstage[l] into the recurrent half of the next step's [x | h] rows (the beam tail's
// gather); a step that leaves its state in place cannot fuse, because the gate GEMM is still reading those rows
int lstm_step_impl(const mic_lstm_decoder* dec, const mic_lstm_workspace* ws, const int32_t* tokens, int ld_tok, int pos,
                   const void* feat_t, int ld_feat, bool skip_embed, void* stream, bool fuse_cell = false) {
    MIC_ARG_CHECK(dec && ws && (tokens || feat_t || skip_embed), "mic_lstm_step: null pointer");
    MIC_ARG_CHECK(dec->dtype == MIC_F32 || dec->dtype == MIC_BF16, "mic_lstm_step: bad dtype");
    MIC_ARG_CHECK(dec->layers >= 1 && dec->layers <= MIC_MAX_LAYERS && dec->E > 0 && dec->Hd > 0 && ws->N > 0, "mic_lstm_step: bad sizes");
    MIC_ARG_CHECK(dec->E % 4 == 0 && dec->Hd % 4 == 0, "mic_lstm_step: E=%d and Hd=%d must be multiples of 4", dec->E, dec->Hd);
    hipStream_t s = (hipStream_t)stream;
    const int dt = dec->dtype, N = ws->N, E = dec->E, Hd = dec->Hd, L = dec->layers;
    const size_t sz = mic_dtype_size(dt);
    if (feat_t) {
        MIC_HIP_CHECK(hipMemcpy2DAsync(ws->xh[0], (size_t)dec->kpad[0] * sz, feat_t, (size_t)ld_feat * sz, (size_t)E * sz, N,
                                       hipMemcpyDeviceToDevice, s));
    } else if (!skip_embed) {
        MIC_TRY(mic_embed_pos(tokens, ld_tok, pos, dec->embed, nullptr, nullptr, ws->xh[0], dec->kpad[0], dt, N, E, dec->vocab, s));
    }
    static const bool no_cell_ln = getenv("MIC_NO_CELL_LN") != nullptr;  // comparison runs: LayerNorm as its own launch
    bool ln_done = false;
    for (int l = 0; l < L; ++l) {
        const int in_dim = l == 0 ? E : Hd;
        void* up = nullptr;
        int ld_up = 0;
        if (l + 1 < L) { up = ws->xh[l + 1]; ld_up = dec->kpad[l + 1]; }
        else if (dec->kind == 0) { up = ws->hnt; ld_up = dec->out_k; }
        if (fuse_cell && lstm_cell_fusable(dec, ws, l)) {
            mic_skinny_desc d{};
            d.x = ws->xh[l]; d.ld_x = dec->kpad[l]; d.wf = dec->wf_cell[l]; d.bias = dec->b_cell[l];
            d.dtype = dt; d.out_dtype = MIC_F32; d.M = N; d.N = 4 * Hd; d.K = dec->kpad[l];
            MIC_TRY(mic_lstm_gates_cell(&d, ws->c[l], ws->h[l], (char*)ws->hstage[l] + (size_t)in_dim * sz, dec->kpad[l], up, ld_up, Hd, s));
            continue;
        }
        MIC_TRY(proj(dt, dec->wf[l], ws->xh[l], dec->kpad[l], N, dec->kpad[l], dec->w[l], dec->b[l], ws->gates, MIC_F32, 4 * Hd, 4 * Hd, 0, s));
        void* own = (char*)ws->xh[l] + (size_t)in_dim * sz;
        if (l + 1 == L && dec->kind == 1 && !no_cell_ln && mic_lstm_cell_ln_supported(Hd)) {
            MIC_TRY(mic_lstm_cell_ln(ws->gates, ws->c[l], ws->h[l], own, dec->kpad[l], dec->ln_g, dec->ln_b, ws->hn, ws->hnt, dec->out_k, dt, N,
                                     Hd, s));
            ln_done = true;
        } else {
            MIC_TRY(mic_lstm_cell(ws->gates, ws->c[l], ws->h[l], own, dec->kpad[l], up, ld_up, dt, N, Hd, s));
        }
    }
    if (dec->kind == 1 && !ln_done)
        MIC_TRY(mic_layernorm(ws->h[L - 1], dec->ln_g, dec->ln_b, ws->hn, ws->hnt, dec->out_k, dt, N, Hd, s));
    return proj(dt, dec->out_wf, ws->hnt, dec->out_k, N, dec->out_k, dec->out_w, dec->out_b, ws->logits, MIC_F32, dec->vocab, dec->vocab, 0, s);
}
}  // namespace

extern "C" int mic_lstm_step(const mic_lstm_decoder* dec, const mic_lstm_workspace* ws, const int32_t* tokens, int ld_tok, int pos,
                             const void* feat_t, int ld_feat, void* stream) {
    MIC_ARG_CHECK(dec && ws && (tokens || feat_t), "mic_lstm_step: null pointer");
    return lstm_step_impl(dec, ws, tokens, ld_tok, pos, feat_t, ld_feat, false, stream);
}

namespace {
int check_lstm_pair(const mic_lstm_decoder* dec, const mic_lstm_workspace* ws, int n_ws, int t0, int t1, int t_cap, const char* who) {
    MIC_ARG_CHECK(dec && ws, "%s: null pointer", who);
    MIC_ARG_CHECK(t0 >= 0 && t1 >= t0 && t1 + 1 <= t_cap, "%s: steps [%d,%d) do not fit t_cap=%d", who, t0, t1, t_cap);
    for (int i = 0; i < n_ws; ++i) MIC_ARG_CHECK(ws[i].N == ws[0].N && ws[i].logits, "%s: workspaces differ in rows or lack logits", who);
    return 0;
}
}  // namespace

// Steps [t0, t1) of LSTM beam search.  Step t runs on workspace t & 1; the survivors' recurrent state
// (c and the [x | h] GEMM input of every layer) is gathered by parent row into workspace (t + 1) & 1.
extern "C" int mic_lstm_beam_steps(const mic_lstm_decoder* dec, const mic_lstm_workspace ws[2], const mic_beam_state st[2],
                                   int32_t* parent, int32_t* live_images, const double* sqrt_tab, float* top_val, int32_t* top_idx,
                                   int B, int K, int t0, int t1, int t_cap, int eos_id, int use_graph, void* stream) {
    MIC_TRY(check_lstm_pair(dec, ws, 2, t0, t1, t_cap, "mic_lstm_beam_steps"));
    MIC_ARG_CHECK(st && parent && live_images && sqrt_tab && top_val && top_idx, "mic_lstm_beam_steps: null pointer");
    MIC_ARG_CHECK(B > 0 && K > 0 && B * K == ws[0].N, "mic_lstm_beam_steps: workspace rows %d != B*K = %d*%d", ws[0].N, B, K);
    MIC_ARG_CHECK(dec->layers >= 1 && dec->layers <= MIC_MAX_LAYERS, "mic_lstm_beam_steps: bad layer count %d", dec->layers);
    const int N = B * K;
    const int sz = (int)mic_dtype_size(dec->dtype);
    static const bool no_tail = getenv("MIC_NO_BEAM_TAIL") != nullptr;
    // one launch for top-k, selection, the survivors' state and the next embedding: the recurrent halves of every layer's
    // [x | h] matrix and the cell states are all that must travel (the input halves are rewritten before they are read)
    const bool tail = !no_tail && mic_beam_tail_supported(K, dec->vocab) && 2 * dec->layers <= MIC_GATHER_MAX && dec->E % 4 == 0;
    auto body = [&](void* stream) -> int {
        for (int t = t0; t < t1; ++t) {
            const mic_lstm_workspace* wc = &ws[t & 1];
            const mic_lstm_workspace* wn = &ws[(t + 1) & 1];
            const mic_beam_state* cur = &st[t & 1];
            const mic_beam_state* nxt = &st[(t + 1) & 1];
            MIC_TRY(lstm_step_impl(dec, wc, cur->seq, t_cap, t, nullptr, 0, tail && t > t0, stream, tail));
            if (tail) {
                mic_beam_tail_desc d{};
                d.logits = wc->logits; d.ld = dec->vocab; d.V = dec->vocab; d.cur = cur; d.nxt = nxt; d.parent = parent;
                d.live_images = live_images + t; d.sqrt_tab = sqrt_tab;
                d.B = B; d.K = K; d.n_live_slots = t == 0 ? 1 : K; d.t = t; d.t_cap = t_cap; d.eos_id = eos_id; d.finished_in_place = 1;
                if (t + 1 < t1) {
                    d.tok_emb = dec->embed; d.xt = wn->xh[0]; d.ld_xt = dec->kpad[0]; d.dtype = dec->dtype; d.H = dec->E; d.vocab = dec->vocab;
                }
                for (int l = 0; l < dec->layers; ++l) {
                    const int in_dim = l == 0 ? dec->E : dec->Hd;
                    d.gather_src[2 * l] = wc->c[l]; d.gather_dst[2 * l] = wn->c[l];
                    d.gather_row_bytes[2 * l] = d.gather_stride_bytes[2 * l] = dec->Hd * 4;
                    // h': from the layer's own recurrent half, or from the staging rows when its cell ran in the gate GEMM
                    const bool staged = lstm_cell_fusable(dec, wc, l);
                    d.gather_src[2 * l + 1] = (const char*)(staged ? wc->hstage[l] : wc->xh[l]) + (size_t)in_dim * sz;
                    d.gather_dst[2 * l + 1] = (char*)wn->xh[l] + (size_t)in_dim * sz;
                    d.gather_row_bytes[2 * l + 1] = dec->Hd * sz; d.gather_stride_bytes[2 * l + 1] = dec->kpad[l] * sz;
                }
                d.n_gather = 2 * dec->layers;
                MIC_TRY(mic_beam_tail_ex(&d, stream));
                continue;
            }
            MIC_TRY(mic_logsoftmax_topk(wc->logits, dec->vocab, N, dec->vocab, K, 1.0f, 0, top_val, top_idx, nullptr, stream));
            MIC_TRY(mic_beam_select(cur, nxt, top_val, top_idx, parent, live_images + t, sqrt_tab, B, K, t == 0 ? 1 : K, t, t_cap,
                                    eos_id, 1, stream));
            // the fp32 h of a layer is rewritten by the next step's cell before anything reads it: only the cell
            // state and the GEMM input [x | h] travel with the survivors, all layers in one launch
            const void* src[2 * MIC_MAX_LAYERS];
            void* dst[2 * MIC_MAX_LAYERS];
            int32_t bytes[2 * MIC_MAX_LAYERS];
            for (int l = 0; l < dec->layers; ++l) {
                src[2 * l] = wc->c[l]; dst[2 * l] = wn->c[l]; bytes[2 * l] = dec->Hd * 4;
                src[2 * l + 1] = wc->xh[l]; dst[2 * l + 1] = wn->xh[l]; bytes[2 * l + 1] = dec->kpad[l] * sz;
            }
            MIC_TRY(mic_gather_rows_multi(src, dst, bytes, 2 * dec->layers, parent, N, stream));
        }
        return 0;
    };
    if (!use_graph || t1 == t0) return body(stream);
    return run_graphed(make_key("lstm_beam", *dec, ws[0], ws[1], st[0], st[1], parent, live_images, sqrt_tab, top_val, top_idx, B, K, t0,
                                t1, t_cap, eos_id),
                       (hipStream_t)stream, body);
}

// Steps [t0, t1) of batched greedy (u == NULL) or temperature sampling (u = uniforms [steps][N]) on one workspace.
extern "C" int mic_lstm_sample_steps(const mic_lstm_decoder* dec, const mic_lstm_workspace* ws, int32_t* tokens, uint8_t* finished,
                                     int32_t* live_rows, float inv_temp, const float* u, float* val, int32_t* idx, int t0, int t1,
                                     int t_cap, int pad_id, int eos_id, int use_graph, void* stream) {
    MIC_TRY(check_lstm_pair(dec, ws, 1, t0, t1, t_cap, "mic_lstm_sample_steps"));
    MIC_ARG_CHECK(tokens && finished && live_rows && val && idx, "mic_lstm_sample_steps: null pointer");
    const int N = ws->N;
    auto body = [&](void* stream) -> int {
        for (int t = t0; t < t1; ++t) {
            MIC_TRY(mic_lstm_step(dec, ws, tokens, t_cap, t, nullptr, 0, stream));
            if (u)
                MIC_TRY(mic_sample_multinomial(ws->logits, dec->vocab, N, dec->vocab, inv_temp, u + (int64_t)t * N, val, idx, stream));
            else
                MIC_TRY(mic_logsoftmax_topk(ws->logits, dec->vocab, N, dec->vocab, 1, 1.0f, 1, val, idx, nullptr, stream));
            MIC_TRY(mic_sample_update(nullptr, idx, tokens, t_cap, t + 1, finished, nullptr, live_rows + t, N, pad_id, eos_id, stream));
        }
        return 0;
    };
    if (!use_graph || t1 == t0) return body(stream);
    return run_graphed(make_key("lstm_samp", *dec, *ws, tokens, finished, live_rows, inv_temp, u, val, idx, t0, t1, t_cap, pad_id, eos_id),
                       (hipStream_t)stream, body);
}
