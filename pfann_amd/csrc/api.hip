// C-ABI of libpfann_amd.so (include/pfann_amd.h): handles, weight re-layout, workspaces,
// per-kernel HIP-event profiling and the reference's native seam (version / seq_score).
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>

#include "kernels.h"

namespace pfann { int launch_rows_to_half(const float *x, int64_t n, int d, void *xh, float *norm_max_dev, hipStream_t s); }

__global__ void noop_api_kernel() {}
static int launch_noop_api() {
    hipLaunchKernelGGL(noop_api_kernel, dim3(1), dim3(1), 0, 0);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

namespace pfann {

static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- profiling ------------------------------------------------------------------------
struct ProfRec { std::string tag; hipEvent_t e0, e1; double work; };
static bool g_prof = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static hipEvent_t g_pending = nullptr;
bool prof_on() { return g_prof; }
static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
void prof_begin(const char *, hipStream_t s) {
    g_pending = get_event();
    (void)hipEventRecord(g_pending, s);
}
void prof_end(const char *tag, hipStream_t s, double work) {
    hipEvent_t e1 = get_event();
    (void)hipEventRecord(e1, s);
    g_recs.push_back({tag, g_pending, e1, work});
    g_pending = nullptr;
}

}  // namespace pfann

using namespace pfann;

// =========================================================================================
// encoder context
// =========================================================================================
struct pfann_ctx {
    pfann_config cfg;
    int device;
    MelPlan mel;
    bool mel_ready = false;
    int F, T;                       // encoder input dims (n_mels, n_frames)
    SubLayer sub[16];
    float *g_w1 = nullptr, *g_b1 = nullptr, *g_w2 = nullptr, *g_b2 = nullptr;
    std::map<std::string, bool> loaded;
    int n_expected = 0;
    float *buf[2] = {nullptr, nullptr};
    int64_t buf_elems[2] = {0, 0};  // per sample
    float *mel_buf = nullptr;
    double *scratch = nullptr;
    std::vector<float> w1_host, b1_host;   // first conv [3][co] / [co] (Gram-form LayerNorm statistics)
    float gram[14] = {};
    bool gram_ready = false;
    bool fused = false;             // LayerNorm fused into the GEMMs (encoder_fused.hip)
    int precision = 0;              // 0: fp32 MFMA (exact); 1: 3-term fp16 split on the fp16 MFMA (opt-in)
    int64_t plan_batch = 0;         // pfann_set_plan_batch: kernel variants chosen for this batch size (0: each call's own)
    int n_streams = 1;              // sub-batches run on this many internal streams (MFMA-bound GEMMs of one
                                    // sub-batch overlap the HBM-bound LayerNorm passes of another)
    hipStream_t side[8] = {};
    hipEvent_t ev_fork = nullptr, ev_join[8] = {};
    float *part[2] = {nullptr, nullptr};
    float *splitk = nullptr;        // partial tiles of the split-K GEMMs (small batches), allocated on first use
    size_t splitk_bytes = 0;
    float *stats = nullptr;         // [max_batch][2] (mean, rstd) of the current GEMM input
    int64_t part_slots = 0;         // per sample
    bool keep = false;
    int64_t keep_B = 0;
    float *dbg[16] = {};
    float *dbg_tmp = nullptr;
    int64_t dbg_cap = 0;
};

extern "C" int pfann_set_streams(pfann_ctx *c, int n);

static int build_plan(pfann_ctx *c) {
    const pfann_config &g = c->cfg;
    const int ch[9] = {1, g.d, g.d, 2 * g.d, 2 * g.d, 4 * g.d, 4 * g.d, g.h, g.h};
    int F = c->F, T = c->T;
    for (int i = 0; i < 8; ++i) {
        const int st = g.stride_t[i] > 0 ? g.stride_t[i] : 2;
        const int sf = g.stride_f[i] > 0 ? g.stride_f[i] : 2;
        const int p1 = (T - 1) / st * st + 3 - T, p2 = (F - 1) / sf * sf + 3 - F;
        const int T1 = (T - 1) / st + 1, F2 = (F - 1) / sf + 1;
        SubLayer &a = c->sub[2 * i], &b = c->sub[2 * i + 1];
        a = SubLayer{};
        b = SubLayer{};
        a.ci = ch[i]; a.co = ch[i + 1]; a.F = F; a.T = T; a.Fo = F; a.To = T1;
        a.axis = 0; a.stride = st; a.pad_lo = p1 / 2; a.depthwise = 0;
        b.ci = ch[i + 1]; b.co = ch[i + 1]; b.F = F; b.T = T1; b.Fo = F2; b.To = T1;
        b.axis = 1; b.stride = sf; b.pad_lo = p2 / 2; b.depthwise = g.fuller ? 0 : 1;
        F = F2; T = T1;
    }
    if (F != 1 || T != 1) { set_error("encoder output must be 1x1, got %dx%d", F, T); return -1; }
    if (g.h % g.d != 0) { set_error("h must be divisible by d"); return -1; }
    if (g.d % 4 != 0) { set_error("d must be a multiple of 4"); return -1; }
    return 0;
}

static int upload(float **dst, const float *src, size_t n) {
    if (*dst == nullptr) PF_HIP(hipMalloc(dst, n * sizeof(float)));
    PF_HIP(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

__global__ void pfann_bench_region_marker() {}

extern "C" {

long long version(void) { return PFANN_SEQSCORE_VERSION; }

const char *pfann_last_error(void) { return g_err; }

pfann_ctx *pfann_create(const pfann_config *cfg, int device) {
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed: no usable GPU", device); return nullptr; }
    pfann_ctx *c = new pfann_ctx();
    c->cfg = *cfg;
    c->device = device;
    if (c->cfg.max_batch <= 0) c->cfg.max_batch = 512;
    const int n_fft = cfg->stft_n;
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    if ((1 << log2n) != n_fft || n_fft < 64 || n_fft > 4096) { set_error("stft_n must be a power of two in 64..4096"); delete c; return nullptr; }
    if (cfg->n_mels % 1 != 0 || cfg->n_mels <= 0) { set_error("bad n_mels"); delete c; return nullptr; }
    MelPlan &m = c->mel;
    memset(&m, 0, sizeof(m));
    m.seg_len = cfg->segment_len; m.n_fft = n_fft; m.hop = cfg->stft_hop; m.n_mels = cfg->n_mels;
    m.n_frames = 1 + cfg->segment_len / cfg->stft_hop;
    m.n_freqs = n_fft / 2 + 1; m.log2n = log2n;
    m.power = cfg->power; m.pad_reflect = cfg->pad_reflect; m.log_mode = cfg->log_mode;
    m.spec_norm_max = cfg->spec_norm_max; m.log_eps = cfg->log_eps;
    if (m.pad_reflect && n_fft / 2 >= cfg->segment_len) { set_error("reflect padding needs stft_n/2 < segment_len"); delete c; return nullptr; }
    // periodic hann window and twiddles, computed in double on the host
    std::vector<float> win(n_fft);
    std::vector<float2> tw(n_fft / 2);
    for (int n = 0; n < n_fft; ++n) win[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / n_fft));
    for (int j = 0; j < n_fft / 2; ++j) {
        tw[j].x = (float)cos(-2.0 * M_PI * j / n_fft);
        tw[j].y = (float)sin(-2.0 * M_PI * j / n_fft);
    }
    if (hipMalloc(&m.window, n_fft * sizeof(float)) != hipSuccess ||
        hipMalloc(&m.twiddle, (n_fft / 2) * sizeof(float2)) != hipSuccess) { set_error("hipMalloc failed"); delete c; return nullptr; }
    (void)hipMemcpy(m.window, win.data(), n_fft * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemcpy(m.twiddle, tw.data(), (n_fft / 2) * sizeof(float2), hipMemcpyHostToDevice);
    c->F = cfg->n_mels;
    c->T = m.n_frames;
    if (build_plan(c)) { delete c; return nullptr; }
    c->n_expected = 8 * 8 + 4;
    // workspaces: ping-pong activation buffers (even sub-layers -> buf[0], odd -> buf[1])
    for (int i = 0; i < 16; ++i) {
        const int64_t e = (int64_t)c->sub[i].co * c->sub[i].Fo * c->sub[i].To;
        c->buf_elems[i & 1] = std::max(c->buf_elems[i & 1], e);
    }
    c->fused = fused_supported(c->sub, 16) && getenv("PFANN_NO_FUSE") == nullptr;
    if (!c->fused && cfg->fuller && getenv("PFANN_NO_FUSE") == nullptr)
        fprintf(stderr, "pfann_amd: this model's layer shapes (an Fo*To that is not a power of two, or channel counts not "
                        "multiples of 4) are outside the LayerNorm-fused GEMM path; using the separate-LayerNorm kernels\n");
    if (getenv("PFANN_STREAMS")) pfann_set_streams(c, atoi(getenv("PFANN_STREAMS")));
    for (int i = 0; i < 16; ++i) {
        const SubLayer &L = c->sub[i];
        const int rps = L.Fo * L.To;
        const int64_t slots = (int64_t)(rps >= 64 ? rps / 64 : 1) * cdiv(L.co, 64);
        c->part_slots = std::max(c->part_slots, slots);
    }
    if (hipMalloc(&c->scratch, 16 * sizeof(double)) != hipSuccess) {
        set_error("hipMalloc failed");
        delete c;
        return nullptr;
    }
    return c;
}

void pfann_destroy(pfann_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 16; ++i) {
        if (c->sub[i].w) (void)hipFree(c->sub[i].w);
        if (c->sub[i].w_hi) { (void)hipFree(c->sub[i].w_hi); (void)hipFree(c->sub[i].w_lo); }
        if (c->sub[i].w22) (void)hipFree(c->sub[i].w22);
        if (c->sub[i].bias) (void)hipFree(c->sub[i].bias);
        if (c->sub[i].ln_w) (void)hipFree(c->sub[i].ln_w);
        if (c->sub[i].ln_b) (void)hipFree(c->sub[i].ln_b);
        if (c->dbg[i]) (void)hipFree(c->dbg[i]);
    }
    float *ptrs[] = {c->g_w1, c->g_b1, c->g_w2, c->g_b2, c->buf[0], c->buf[1], c->part[0], c->part[1], c->splitk, c->stats, c->mel_buf, c->mel.window,
                     c->mel.fb_val, c->dbg_tmp};
    for (float *p : ptrs) if (p) (void)hipFree(p);
    if (c->mel.twiddle) (void)hipFree(c->mel.twiddle);
    if (c->mel.fb_ptr) (void)hipFree(c->mel.fb_ptr);
    if (c->mel.fb_idx) (void)hipFree(c->mel.fb_idx);
    if (c->scratch) (void)hipFree(c->scratch);
    delete c;
}

int pfann_set_melbank(pfann_ctx *c, const float *fb, int n_freqs, int n_mels) {
    if (n_freqs != c->mel.n_freqs || n_mels != c->mel.n_mels) {
        set_error("melbank shape [%d,%d] != [%d,%d]", n_freqs, n_mels, c->mel.n_freqs, c->mel.n_mels);
        return -3;
    }
    PF_HIP(hipSetDevice(c->device));
    std::vector<int> ptr(n_mels + 1, 0), idx;
    std::vector<float> val;
    int mx = 0;
    for (int m = 0; m < n_mels; ++m) {
        for (int k = 0; k < n_freqs; ++k) {
            const float v = fb[(size_t)k * n_mels + m];
            if (v != 0.0f) { idx.push_back(k); val.push_back(v); }
        }
        ptr[m + 1] = (int)idx.size();
        mx = std::max(mx, ptr[m + 1] - ptr[m]);
    }
    if (idx.empty()) { idx.push_back(0); val.push_back(0.f); }
    MelPlan &mp = c->mel;
    if (mp.fb_ptr) { (void)hipFree(mp.fb_ptr); (void)hipFree(mp.fb_idx); (void)hipFree(mp.fb_val); }
    PF_HIP(hipMalloc(&mp.fb_ptr, ptr.size() * sizeof(int)));
    PF_HIP(hipMalloc(&mp.fb_idx, idx.size() * sizeof(int)));
    PF_HIP(hipMalloc(&mp.fb_val, val.size() * sizeof(float)));
    PF_HIP(hipMemcpy(mp.fb_ptr, ptr.data(), ptr.size() * sizeof(int), hipMemcpyHostToDevice));
    PF_HIP(hipMemcpy(mp.fb_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    PF_HIP(hipMemcpy(mp.fb_val, val.data(), val.size() * sizeof(float), hipMemcpyHostToDevice));
    mp.max_nnz_row = mx;
    mp.fb_nnz = (int)idx.size();
    c->mel_ready = true;
    return 0;
}

int pfann_load_weight(pfann_ctx *c, const char *name, const float *host, int64_t numel) {
    PF_HIP(hipSetDevice(c->device));
    const pfann_config &g = c->cfg;
    int blk = -1;
    char mod[16] = "", kind[16] = "";
    if (sscanf(name, "f.convs.%d.%15[a-z0-9].%15s", &blk, mod, kind) == 3 && blk >= 0 && blk < 8) {
        const bool second = mod[strlen(mod) - 1] == '2';
        SubLayer &L = c->sub[2 * blk + (second ? 1 : 0)];
        const bool is_w = strcmp(kind, "weight") == 0;
        if (!is_w && strcmp(kind, "bias") != 0) { set_error("unknown tensor %s", name); return -2; }
        if (strncmp(mod, "conv", 4) == 0) {
            if (!is_w) {
                if (numel != L.co) { set_error("%s: numel %lld != %d", name, (long long)numel, L.co); return -3; }
                if (upload(&L.bias, host, numel)) return -1;
                if (blk == 0 && !second) { c->b1_host.assign(host, host + numel); c->gram_ready = false; }
            } else {
                const int ci = L.depthwise ? 1 : L.ci;
                if (numel != (int64_t)L.co * ci * 3) { set_error("%s: numel %lld != %lld", name, (long long)numel, (long long)L.co * ci * 3); return -3; }
                std::vector<float> w((size_t)numel);
                if (ci == 1) {          // [co][1][3] -> [3][co]
                    for (int o = 0; o < L.co; ++o)
                        for (int t = 0; t < 3; ++t) w[(size_t)t * L.co + o] = host[(size_t)o * 3 + t];
                } else {                // [co][ci][3] -> [co][3][ci]
                    for (int o = 0; o < L.co; ++o)
                        for (int i = 0; i < ci; ++i)
                            for (int t = 0; t < 3; ++t)
                                w[((size_t)o * 3 + t) * ci + i] = host[((size_t)o * ci + i) * 3 + t];
                }
                if (upload(&L.w, w.data(), numel)) return -1;
                if (blk == 0 && !second) { c->w1_host = w; c->gram_ready = false; }
                if (ci > 1) {
                    // {W1, W0, -W2, W0 + W2} per output channel for the five-block kernel (encoder_fused.hip)
                    std::vector<float> w4((size_t)L.co * 4 * ci);
                    for (int o = 0; o < L.co; ++o)
                        for (int i = 0; i < ci; ++i) {
                            const float w0 = w[((size_t)o * 3 + 0) * ci + i], w1 = w[((size_t)o * 3 + 1) * ci + i],
                                        w2 = w[((size_t)o * 3 + 2) * ci + i];
                            float *d4 = &w4[(size_t)o * 4 * ci + i];
                            d4[0] = w1; d4[ci] = w0; d4[2 * (size_t)ci] = -w2; d4[3 * (size_t)ci] = w0 + w2;
                        }
                    if (upload(&L.w22, w4.data(), (int64_t)w4.size())) return -1;
                }
                if (ci > 1) {
                    // 2-term fp16 split of w * 2^e (largest magnitude in [2^14, 2^15)): hi = fl16(ws),
                    // lo = fl16(ws - hi) exactly representable residual -> 2^-22 relative, both in the normal range
                    float mx = 0.f;
                    for (float v : w) mx = std::max(mx, std::fabs(v));
                    int e = 0;
                    if (mx > 0.f) { (void)std::frexp(mx, &e); e = 15 - e; }      // mx * 2^e in [2^14, 2^15)
                    const float sc = std::ldexp(1.0f, e);
                    std::vector<_Float16> hi((size_t)numel), lo((size_t)numel);
                    for (size_t i = 0; i < (size_t)numel; ++i) {
                        const float ws = w[i] * sc;
                        hi[i] = (_Float16)ws;
                        lo[i] = (_Float16)(ws - (float)hi[i]);
                    }
                    if (!L.w_hi) { PF_HIP(hipMalloc(&L.w_hi, numel * 2)); PF_HIP(hipMalloc(&L.w_lo, numel * 2)); }
                    PF_HIP(hipMemcpy(L.w_hi, hi.data(), numel * 2, hipMemcpyHostToDevice));
                    PF_HIP(hipMemcpy(L.w_lo, lo.data(), numel * 2, hipMemcpyHostToDevice));
                    L.w_inv_scale = std::ldexp(1.0f, -e);
                }
            }
        } else if (strncmp(mod, "ln", 2) == 0) {
            const int64_t hw = (int64_t)L.Fo * L.To;
            if (numel != L.co * hw) { set_error("%s: numel %lld != %lld", name, (long long)numel, (long long)(L.co * hw)); return -3; }
            std::vector<float> w((size_t)numel);   // [co][Fo][To] -> [Fo][To][co]
            for (int o = 0; o < L.co; ++o)
                for (int64_t p = 0; p < hw; ++p) w[(size_t)p * L.co + o] = host[(size_t)o * hw + p];
            if (upload(is_w ? &L.ln_w : &L.ln_b, w.data(), numel)) return -1;
        } else {
            set_error("unknown tensor %s", name);
            return -2;
        }
    } else if (strncmp(name, "g.linear", 8) == 0) {
        const int v = g.h / g.d;
        float **dst = nullptr;
        int64_t want = 0;
        if (!strcmp(name, "g.linear1.weight")) { dst = &c->g_w1; want = (int64_t)g.d * g.u * v; }
        else if (!strcmp(name, "g.linear1.bias")) { dst = &c->g_b1; want = (int64_t)g.d * g.u; }
        else if (!strcmp(name, "g.linear2.weight")) { dst = &c->g_w2; want = (int64_t)g.d * g.u; }
        else if (!strcmp(name, "g.linear2.bias")) { dst = &c->g_b2; want = g.d; }
        else { set_error("unknown tensor %s", name); return -2; }
        if (numel != want) { set_error("%s: numel %lld != %lld", name, (long long)numel, (long long)want); return -3; }
        if (upload(dst, host, numel)) return -1;
    } else {
        set_error("unknown tensor %s", name);
        return -2;
    }
    c->loaded[name] = true;
    return 0;
}

int pfann_weights_missing(pfann_ctx *c) { return c->n_expected - (int)c->loaded.size(); }

int pfann_melspec(pfann_ctx *c, const float *segs, int64_t B, int64_t seg_stride, int remove_mean, float *out,
                  void *stream) {
    PF_HIP(hipSetDevice(c->device));
    if (!c->mel_ready) { set_error("pfann_melspec: mel filterbank not set"); return -5; }
    return launch_melspec(c->mel, segs, B, seg_stride, nullptr, remove_mean, out, (hipStream_t)stream);
}

static int keep_tap(pfann_ctx *c, int idx, const float *act, int64_t B, hipStream_t s) {
    const SubLayer &L = c->sub[idx];
    const int64_t e = (int64_t)L.co * L.Fo * L.To;
    const int64_t nb = std::min<int64_t>(B, 8);
    if (!c->dbg[idx]) PF_HIP(hipMalloc(&c->dbg[idx], 8 * e * sizeof(float)));
    PF_HIP(hipMemcpyAsync(c->dbg[idx], act, nb * e * sizeof(float), hipMemcpyDeviceToDevice, s));
    c->keep_B = nb;
    return 0;
}

// activation / mel workspaces are allocated on first use (a mel-only context stays small)
static int ensure_workspace(pfann_ctx *c, bool need_mel) {
    const int64_t mb = c->cfg.max_batch;
    if (!c->buf[0]) {
        // the activation workspace: max_batch x the two largest sub-layer outputs (29 GB at 9728 segments of the default
        // model -- sized for a 288 GB MI355X; a device that cannot give it gets told which knob to turn)
        const size_t need = (size_t)mb * (c->buf_elems[0] + c->buf_elems[1]) * sizeof(float);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < need) {
            set_error("the encoder workspace for max_batch = %lld segments needs %.1f GB but only %.1f GB of device memory are "
                      "free: lower max_batch (the tools: PFANN_MAX_BATCH)", (long long)mb, need / 1e9, free_b / 1e9);
            return -1;
        }
        PF_HIP(hipMalloc(&c->buf[0], mb * c->buf_elems[0] * sizeof(float)));
        PF_HIP(hipMalloc(&c->buf[1], mb * c->buf_elems[1] * sizeof(float)));
        PF_HIP(hipMalloc(&c->part[0], mb * c->part_slots * 2 * sizeof(float)));
        PF_HIP(hipMalloc(&c->part[1], mb * c->part_slots * 2 * sizeof(float)));
        PF_HIP(hipMalloc(&c->stats, mb * 2 * sizeof(float)));
    }
    if (need_mel && !c->mel_buf) PF_HIP(hipMalloc(&c->mel_buf, mb * (int64_t)c->F * c->T * sizeof(float)));
    return 0;
}

static int keep_tap_fused(pfann_ctx *c, int idx, const float *z, const float *part, int P, int64_t B, hipStream_t s) {
    const SubLayer &L = c->sub[idx];
    const int64_t e = (int64_t)L.co * L.Fo * L.To;
    const int64_t nb = std::min<int64_t>(B, 8);
    if (!c->dbg[idx]) PF_HIP(hipMalloc(&c->dbg[idx], 8 * e * sizeof(float)));
    c->keep_B = nb;
    return launch_ln_apply(L, z, part, P, c->dbg[idx], nb, c->cfg.activation, c->cfg.relu_after_bn, s);
}

// Gram form of the first conv's LayerNorm statistics (see conv_first_gram_stats_kernel), in fp64
static void build_gram(pfann_ctx *c) {
    const int co = c->sub[0].co;
    const std::vector<float> &w = c->w1_host, &b = c->b1_host;      // w: [3][co]
    double Bsum = 0, Ws[3] = {0, 0, 0}, bb = 0, h[3] = {0, 0, 0}, G[3][3] = {};
    for (int o = 0; o < co; ++o) {
        Bsum += b[o];
        bb += (double)b[o] * b[o];
        for (int t = 0; t < 3; ++t) {
            Ws[t] += w[t * co + o];
            h[t] += (double)b[o] * w[t * co + o];
            for (int u = 0; u < 3; ++u) G[t][u] += (double)w[t * co + o] * w[u * co + o];
        }
    }
    const double v[14] = {Bsum, Ws[0], Ws[1], Ws[2], bb, h[0], h[1], h[2], G[0][0], G[0][1], G[0][2], G[1][1], G[1][2], G[2][2]};
    for (int i = 0; i < 14; ++i) c->gram[i] = (float)v[i];
    c->gram_ready = true;
}

static int encode_chunk_fused(pfann_ctx *c, const float *mel, int64_t B, float *emb, int normalize, hipStream_t s,
                              int64_t slot0) {
    const pfann_config &g = c->cfg;
    float *buf[2] = {c->buf[0] + slot0 * c->buf_elems[0], c->buf[1] + slot0 * c->buf_elems[1]};
    float *part[2] = {c->part[0] + slot0 * c->part_slots * 2, c->part[1] + slot0 * c->part_slots * 2};
    // Sub-layer 0 (C_in = 1 conv): normally only its LayerNorm statistics are computed here and the
    // conv itself is folded into sub-layer 1's A-loader; the 2 MiB/segment tensor is materialised
    // only when verification taps are requested.
    const bool fold_first = !c->keep && c->sub[1].axis == 1 && c->sub[0].co <= 256 &&
                            getenv("PFANN_NO_FOLD_FIRST") == nullptr;
    const int64_t Bp = c->plan_batch > 0 ? c->plan_batch : B;
    {   // split-K scratch: the largest [n_splits][B rows][N] partial tensor among the layers the plan splits at this batch
        // (grow-only; one query: ~5 MB, 304 windows: ~30 MB; layers whose partials would pass 256 MB are not split)
        size_t need = 0;
        for (int i = 1; i < 16; ++i) {
            const size_t n_i = splitk_scratch_need(c->sub[i], B, Bp);
            if (n_i <= ((size_t)256 << 20)) need = std::max(need, n_i);
        }
        if (need > c->splitk_bytes) {
            if (c->splitk) { PF_HIP(hipStreamSynchronize(s)); (void)hipFree(c->splitk); }
            c->splitk = nullptr; c->splitk_bytes = 0;
            if (hipMalloc(&c->splitk, need) == hipSuccess) c->splitk_bytes = need;
            else { c->splitk = nullptr; (void)hipGetLastError(); }
        }
    }
    if (fold_first && g.relu_after_bn && (int)c->w1_host.size() == 3 * c->sub[0].co && (int)c->b1_host.size() == c->sub[0].co) {
        if (!c->gram_ready) build_gram(c);
        if (launch_conv_first_gram_stats(c->sub[0], mel, part[0], B, c->gram, s)) return -1;
    } else if (launch_conv_first_stats(c->sub[0], mel, fold_first ? nullptr : buf[0], part[0], B, g.activation, g.relu_after_bn, s)) {
        return -1;
    }
    int P = fused_out_slots(c->sub[0], Bp);
    if (c->keep && keep_tap_fused(c, 0, buf[0], part[0], P, B, s)) return -1;
    int stats_final = 0;             // c->stats already holds (mean, rstd) of the next layer's input (split-K reduction)
    for (int i = 1; i < 16; ++i) {
        const bool first = fold_first && i == 1;
        if (c->sub[i].depthwise) {
            stats_final = 0;
            if (launch_conv_dw_ln(c->sub[i], c->sub[i - 1], first ? mel : buf[(i - 1) & 1], part[(i - 1) & 1], P, c->stats + slot0 * 2,
                                  buf[i & 1], part[i & 1], B, g.activation, g.relu_after_bn, first ? &c->sub[0] : nullptr, s)) return -1;
        } else if (launch_conv_gemm_ln(c->sub[i], c->sub[i - 1], first ? mel : buf[(i - 1) & 1], part[(i - 1) & 1], P, c->stats + slot0 * 2, buf[i & 1],
                                part[i & 1], B, g.activation, g.relu_after_bn, first ? &c->sub[0] : nullptr, c->precision, s,
                                c->n_streams == 1 || B < 128 ? c->splitk : nullptr, c->splitk_bytes, &stats_final, Bp)) return -1;
        P = fused_out_slots(c->sub[i], Bp);
        if (c->keep && keep_tap_fused(c, i, buf[i & 1], part[i & 1], P, B, s)) return -1;
    }
    return launch_myg_ln(c->sub[15], buf[1], part[1], P, g.activation, g.relu_after_bn, c->g_w1, c->g_b1,
                         c->g_w2, c->g_b2, g.d, g.u, g.h / g.d, B, emb, normalize, s, Bp);
}

static int encode_chunk1(pfann_ctx *c, const float *mel, int64_t B, float *emb, int normalize, hipStream_t s,
                         int64_t slot0) {
    if (c->fused) return encode_chunk_fused(c, mel, B, emb, normalize, s, slot0);
    const float *x = mel;
    for (int i = 0; i < 16; ++i) {
        const SubLayer &L = c->sub[i];
        float *y = c->buf[i & 1] + slot0 * c->buf_elems[i & 1];
        int rc;
        if (L.ci == 1 && !L.depthwise) rc = launch_conv_first(L, x, y, B, s);
        else if (L.depthwise) rc = launch_conv_depthwise(L, x, y, B, s);
        else rc = launch_conv_gemm(L, x, y, B, s);
        if (rc) return rc;
        if (launch_ln_act(L, y, B, c->cfg.activation, c->cfg.relu_after_bn, s)) return -1;
        if (c->keep && keep_tap(c, i, y, B, s)) return -1;
        x = y;
    }
    const pfann_config &g = c->cfg;
    return launch_myg(x, c->g_w1, c->g_b1, c->g_w2, c->g_b2, g.d, g.u, g.h / g.d, B, emb, normalize, s);
}

// Splits a chunk over the internal streams: fork after the caller's stream, join back into it.
static int encode_chunk(pfann_ctx *c, const float *mel, int64_t B, float *emb, int normalize, hipStream_t s) {
    const int ns = (int)std::min<int64_t>(c->n_streams, B / 128);
    if (ns <= 1) return encode_chunk1(c, mel, B, emb, normalize, s, 0);
    if (!c->ev_fork) {
        PF_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        for (int k = 0; k < 8; ++k) {
            PF_HIP(hipStreamCreateWithFlags(&c->side[k], hipStreamNonBlocking));
            PF_HIP(hipEventCreateWithFlags(&c->ev_join[k], hipEventDisableTiming));
        }
    }
    PF_HIP(hipEventRecord(c->ev_fork, s));
    const int64_t per = (int64_t)c->F * c->T;
    int64_t b0 = 0;
    for (int k = 0; k < ns; ++k) {
        const int64_t nb = B / ns + (k < B % ns ? 1 : 0);
        PF_HIP(hipStreamWaitEvent(c->side[k], c->ev_fork, 0));
        if (encode_chunk1(c, mel + b0 * per, nb, emb + b0 * c->cfg.d, normalize, c->side[k], b0)) return -1;
        PF_HIP(hipEventRecord(c->ev_join[k], c->side[k]));
        PF_HIP(hipStreamWaitEvent(s, c->ev_join[k], 0));
        b0 += nb;
    }
    return 0;
}

int pfann_encode(pfann_ctx *c, const float *mel, int64_t B, float *emb, int normalize, void *stream) {
    PF_HIP(hipSetDevice(c->device));
    if (pfann_weights_missing(c) != 0) { set_error("pfann_encode: %d state_dict tensors not loaded", pfann_weights_missing(c)); return -5; }
    hipStream_t s = (hipStream_t)stream;
    if (ensure_workspace(c, false)) return -1;
    const int64_t per = (int64_t)c->F * c->T;
    for (int64_t b0 = 0; b0 < B; b0 += c->cfg.max_batch) {
        const int64_t nb = std::min<int64_t>(c->cfg.max_batch, B - b0);
        if (encode_chunk(c, mel + b0 * per, nb, emb + b0 * c->cfg.d, normalize, s)) return -1;
    }
    return 0;
}

static int segment_embed_impl(pfann_ctx *c, const float *wav, int64_t B, int64_t seg_stride, const int64_t *starts,
                              float *emb, int normalize, void *stream) {
    PF_HIP(hipSetDevice(c->device));
    if (!c->mel_ready) { set_error("pfann_segment_embed: mel filterbank not set"); return -5; }
    if (pfann_weights_missing(c) != 0) { set_error("pfann_segment_embed: %d state_dict tensors not loaded", pfann_weights_missing(c)); return -5; }
    hipStream_t s = (hipStream_t)stream;
    if (ensure_workspace(c, true)) return -1;
    for (int64_t b0 = 0; b0 < B; b0 += c->cfg.max_batch) {
        const int64_t nb = std::min<int64_t>(c->cfg.max_batch, B - b0);
        const float *src = starts ? wav : wav + b0 * seg_stride;
        if (launch_melspec(c->mel, src, nb, seg_stride, starts ? starts + b0 : nullptr, 1, c->mel_buf, s)) return -1;
        if (encode_chunk(c, c->mel_buf, nb, emb + b0 * c->cfg.d, normalize, s)) return -1;
    }
    return 0;
}

int pfann_segment_embed(pfann_ctx *c, const float *wav, int64_t B, int64_t seg_stride, float *emb, int normalize,
                        void *stream) {
    return segment_embed_impl(c, wav, B, seg_stride, nullptr, emb, normalize, stream);
}

int pfann_segment_embed_at(pfann_ctx *c, const float *wav, const int64_t *starts_dev, int64_t B, float *emb,
                           int normalize, void *stream) {
    return segment_embed_impl(c, wav, B, 0, starts_dev, emb, normalize, stream);
}

int pfann_resample_to_mono(pfann_ctx *c, const int16_t *pcm, int n_ch, const float *kernels, int old_rate, int new_rate, int width,
                           const int64_t *plan, int n_pieces, int64_t n_out, float *tmp, float *wav, void *stream) {
    PF_HIP(hipSetDevice(c->device));
    return launch_resample_to_mono(pcm, n_ch, kernels, old_rate, new_rate, width, plan, n_pieces, n_out, tmp, wav,
                                   reinterpret_cast<float *>(c->scratch), (hipStream_t)stream);
}

int pfann_pcm16_to_mono(pfann_ctx *c, const int16_t *pcm, int64_t n_frames, int n_ch, float *wav, void *stream) {
    PF_HIP(hipSetDevice(c->device));
    if (n_ch < 1) { set_error("n_ch < 1"); return -1; }
    return launch_pcm16_to_mono(pcm, n_frames, n_ch, wav, reinterpret_cast<float *>(c->scratch), (hipStream_t)stream);
}

int pfann_pcm16_files_to_mono(pfann_ctx *c, const void *const *host_pcm, const int64_t *n_samples, const int64_t *dst_off,
                              int n_files, int16_t *pcm_dev, int64_t total, float *wav_dev, void *stream) {
    PF_HIP(hipSetDevice(c->device));
    for (int i = 0; i < n_files; ++i) {
        if (n_samples[i] <= 0) continue;
        if (dst_off[i] < 0 || dst_off[i] + n_samples[i] > total) { set_error("pcm16_files_to_mono: file %d outside the slab", i); return -1; }
        PF_HIP(hipMemcpyAsync(pcm_dev + dst_off[i], host_pcm[i], (size_t)n_samples[i] * sizeof(int16_t), hipMemcpyHostToDevice,
                              (hipStream_t)stream));
    }
    if (total <= 0) return 0;
    return launch_pcm16_to_mono(pcm_dev, total, 1, wav_dev, reinterpret_cast<float *>(c->scratch), (hipStream_t)stream);
}

void pfann_debug_keep(pfann_ctx *c, int on) { c->keep = on != 0; }

int pfann_prewarm(int device) {
    if (hipSetDevice(device) != hipSuccess) { set_error("pfann_prewarm: no HIP device %d", device); return -1; }
    int rc = launch_noop_api();
    rc |= prewarm_mel() | prewarm_encoder() | prewarm_encoder_fused() | prewarm_search() | prewarm_search_f16() | prewarm_rerank();
    if (hipDeviceSynchronize() != hipSuccess) rc = -1;
    return rc ? -1 : 0;
}

int64_t pfann_set_plan_batch(pfann_ctx *c, int64_t n) {
    c->plan_batch = n <= 0 ? 0 : (n < 65 ? 65 : n);      // a plan below 65 would select the small-batch kernels, whose
    return c->plan_batch;                                // scratch is sized for actual batches of at most 64
}

int pfann_set_streams(pfann_ctx *c, int n) {
    c->n_streams = n < 1 ? 1 : (n > 8 ? 8 : n);
    return c->n_streams;
}

int pfann_set_encoder_precision(pfann_ctx *c, int mode) {
    c->precision = (mode == 1 && c->fused) ? 1 : 0;
    return c->precision;
}

int pfann_set_fused_layernorm(pfann_ctx *c, int on) {
    if (on && !fused_supported(c->sub, 16)) return 0;
    c->fused = on != 0;
    return c->fused ? 1 : 0;
}

int64_t pfann_debug_activation(pfann_ctx *c, int idx, int64_t B, float *host, int64_t cap) {
    if (idx < 0 || idx > 15 || !c->dbg[idx]) { set_error("no tap %d (call pfann_debug_keep(ctx,1) before encoding)", idx); return -1; }
    if (hipSetDevice(c->device) != hipSuccess) return -1;
    const SubLayer &L = c->sub[idx];
    const int64_t e = (int64_t)L.co * L.Fo * L.To;
    B = std::min(B, c->keep_B);
    if (B * e > cap) { set_error("tap buffer too small"); return -1; }
    if (c->dbg_cap < B * e) {
        if (c->dbg_tmp) (void)hipFree(c->dbg_tmp);
        if (hipMalloc(&c->dbg_tmp, B * e * sizeof(float)) != hipSuccess) return -1;
        c->dbg_cap = B * e;
    }
    if (launch_cl_to_nchw(c->dbg[idx], c->dbg_tmp, B, L.co, L.Fo * L.To, 0)) return -1;
    if (hipMemcpy(host, c->dbg_tmp, B * e * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return B * e;
}

}  // extern "C"

// =========================================================================================
// database shard
// =========================================================================================
struct pfann_db {
    int d, device;
    int64_t n = 0, label_base = 0;
    float *emb = nullptr;
    int64_t *song_pos = nullptr;
    std::vector<int64_t> song_pos_h;
    int n_songs = 0, song_lo = 0, song_hi = 0;
    SearchWorkspace ws;
    void *emb_h = nullptr;              // fp16 rows: a copy for the batched scan's pre-filter (fp32 storage), or the
                                        // ONLY rows kept (fp16 storage)
    int storage = PFANN_DB_F32;
    float xnorm_max = 0.f;
    bool prefilter = true;
    void *match_scratch = nullptr;      // long-query candidate slab (keys + sums), grown on demand
    size_t match_scratch_bytes = 0;
    // the seq_score seam (the reference's ctypes call, database.py:178-189): device slab, PINNED host image and a
    // private stream, kept between calls; seq_mu serialises concurrent callers on one handle (the reference's seam is
    // re-entrant: cpp/seqscore.cpp keeps no state)
    void *seq_scratch = nullptr;
    size_t seq_scratch_bytes = 0;
    char *seq_host = nullptr;
    size_t seq_host_bytes = 0;
    hipStream_t seq_stream = nullptr;
    std::mutex seq_mu;
};

extern "C" {

pfann_db *pfann_db_create(int d, int device) {
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed: no usable GPU", device); return nullptr; }
    pfann_db *db = new pfann_db();
    db->d = d;
    db->device = device;
    return db;
}

void pfann_db_destroy(pfann_db *db) {
    if (!db) return;
    (void)hipSetDevice(db->device);
    (void)hipDeviceSynchronize();
    if (db->emb) (void)hipFree(db->emb);
    if (db->song_pos) (void)hipFree(db->song_pos);
    if (db->ws.thr) { (void)hipFree(db->ws.thr); (void)hipFree(db->ws.cnt); (void)hipFree(db->ws.cl); }
    if (db->ws.overflow) (void)hipFree(db->ws.overflow);
    if (db->ws.thr_adj) { (void)hipFree(db->ws.thr_adj); (void)hipFree(db->ws.eps); }
    if (db->ws.qh) (void)hipFree(db->ws.qh);
    if (db->ws.row_ovf) (void)hipFree(db->ws.row_ovf);
    if (db->ws.left) (void)hipFree(db->ws.left);
    if (db->match_scratch) (void)hipFree(db->match_scratch);
    if (db->seq_scratch) (void)hipFree(db->seq_scratch);
    if (db->seq_host) (void)hipHostFree(db->seq_host);
    if (db->seq_stream) (void)hipStreamDestroy(db->seq_stream);
    if (db->emb_h) (void)hipFree(db->emb_h);
    delete db;
}

int pfann_db_set_prefilter(pfann_db *db, int on) {
    db->prefilter = on != 0;
    return (db->prefilter && db->emb_h != nullptr) ? 1 : 0;
}
int pfann_db_set_storage(pfann_db *db, int mode) {
    if (mode != PFANN_DB_F32 && mode != PFANN_DB_F16) { set_error("pfann_db_set_storage: unknown mode %d", mode); return -1; }
    if (db->n != 0 && mode != db->storage) { set_error("pfann_db_set_storage: call it before pfann_db_load"); return -1; }
    if (mode == PFANN_DB_F16 && db->d % 8 != 0) { set_error("pfann_db_set_storage: fp16 rows need d %% 8 == 0 (d=%d)", db->d); return -1; }
    db->storage = mode;
    return mode;
}
int pfann_db_dim(pfann_db *db) { return db->d; }
int64_t pfann_db_ntotal(pfann_db *db) { return db->n; }
int64_t pfann_db_bytes(pfann_db *db) { return db->n * db->d * (int64_t)(db->storage == PFANN_DB_F16 ? 2 : 4); }

int pfann_db_load(pfann_db *db, const float *emb, int emb_is_device, int64_t n, const int64_t *song_pos,
                  int n_songs, int64_t label_base) {
    PF_HIP(hipSetDevice(db->device));
    if (db->emb) { (void)hipFree(db->emb); db->emb = nullptr; }
    if (db->song_pos) { (void)hipFree(db->song_pos); db->song_pos = nullptr; }
    if (db->emb_h) { (void)hipFree(db->emb_h); db->emb_h = nullptr; }
    db->n = 0;
    db->label_base = label_base;
    db->n_songs = n_songs;
    db->xnorm_max = 0.f;
    if (n > 0 && db->storage == PFANN_DB_F16) {
        // fp16-only storage: the fp32 rows pass through a bounded staging buffer and are never kept
        float *nm = nullptr, *stage = nullptr;
        const int64_t chunk = std::max<int64_t>(1, (64ll << 20) / ((int64_t)db->d * 4));
        PF_HIP(hipMalloc(&db->emb_h, (size_t)n * db->d * 2));
        PF_HIP(hipMalloc(&nm, sizeof(float)));
        PF_HIP(hipMemset(nm, 0, sizeof(float)));
        if (!emb_is_device) PF_HIP(hipMalloc(&stage, (size_t)std::min(chunk, n) * db->d * sizeof(float)));
        for (int64_t r0 = 0; r0 < n; r0 += chunk) {
            const int64_t nr = std::min(chunk, n - r0);
            const float *src = emb + r0 * db->d;
            if (!emb_is_device) {
                PF_HIP(hipMemcpy(stage, src, (size_t)nr * db->d * sizeof(float), hipMemcpyHostToDevice));
                src = stage;
            }
            if (launch_rows_to_half(src, nr, db->d, reinterpret_cast<char *>(db->emb_h) + (size_t)r0 * db->d * 2, nm, 0)) return -1;
            PF_HIP(hipDeviceSynchronize());
        }
        PF_HIP(hipMemcpy(&db->xnorm_max, nm, sizeof(float), hipMemcpyDeviceToHost));
        (void)hipFree(nm);
        if (stage) (void)hipFree(stage);
        if (!(db->xnorm_max < 6.0e4f)) {
            (void)hipFree(db->emb_h);
            db->emb_h = nullptr;
            set_error("db_load: rows with norm %g do not fit fp16 storage", (double)db->xnorm_max);
            return -3;
        }
    } else if (n > 0) {
        PF_HIP(hipMalloc(&db->emb, (size_t)n * db->d * sizeof(float)));
        PF_HIP(hipMemcpy(db->emb, emb, (size_t)n * db->d * sizeof(float),
                         emb_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
        // fp16 copy + largest row norm for the batched scan's pre-filter (search_f16.hip)
        if (db->d % 8 == 0 && getenv("PFANN_NO_F16_PREFILTER") == nullptr) {
            float *nm = nullptr;
            PF_HIP(hipMalloc(&db->emb_h, (size_t)n * db->d * 2));
            PF_HIP(hipMalloc(&nm, sizeof(float)));
            PF_HIP(hipMemset(nm, 0, sizeof(float)));
            if (launch_rows_to_half(db->emb, n, db->d, db->emb_h, nm, 0)) return -1;
            PF_HIP(hipMemcpy(&db->xnorm_max, nm, sizeof(float), hipMemcpyDeviceToHost));
            (void)hipFree(nm);
            if (!(db->xnorm_max < 1.0e4f)) {       // fp16 range / NaN guard: keep the exact-fp32 scan only
                (void)hipFree(db->emb_h);
                db->emb_h = nullptr;
            }
        }
    }
    db->n = n;
    db->song_pos_h.assign(song_pos, song_pos + n_songs + 1);
    PF_HIP(hipMalloc(&db->song_pos, (size_t)(n_songs + 1) * sizeof(int64_t)));
    PF_HIP(hipMemcpy(db->song_pos, song_pos, (size_t)(n_songs + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    // songs whose rows all live in [label_base, label_base + n)
    int lo = 0;
    while (lo < n_songs && song_pos[lo] < label_base) ++lo;
    int hi = lo;
    while (hi < n_songs && song_pos[hi + 1] <= label_base + n) ++hi;
    db->song_lo = lo;
    db->song_hi = hi;
    if (n > 0 && (lo >= n_songs || song_pos[lo] != label_base || song_pos[hi] != label_base + n)) {
        set_error("db_load: shard rows [%lld,%lld) do not align with song boundaries", (long long)label_base,
                  (long long)(label_base + n));
        return -3;
    }
    return 0;
}

int pfann_search_topk(pfann_db *db, const float *q, int64_t nq, int k, float *D, int64_t *I, void *stream) {
    PF_HIP(hipSetDevice(db->device));
    // the survivor workspace is 64 KB per query row: bound it by walking big batches in chunks
    const int64_t chunk = 16384;
    for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
        const int64_t n = std::min(chunk, nq - q0);
        const int rc = search_topk(db->emb, (db->prefilter || db->emb == nullptr) ? db->emb_h : nullptr, db->xnorm_max, db->n, db->d,
                                   db->label_base, q + q0 * db->d, n, k, D + q0 * k, I + q0 * k, db->ws,
                                   (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

int pfann_search_bound(pfann_db *db, const float *q, int64_t nq, int k, int m, float *lb, void *stream) {
    PF_HIP(hipSetDevice(db->device));
    if (nq > 16384) { set_error("pfann_search_bound: at most 16384 query rows per call (got %lld)", (long long)nq); return -1; }
    if (m < 1 || m > 1024) { set_error("pfann_search_bound: m=%d outside 1..1024", m); return -1; }
    return search_topk(db->emb, (db->prefilter || db->emb == nullptr) ? db->emb_h : nullptr, db->xnorm_max, db->n, db->d,
                       db->label_base, q, nq, k, nullptr, nullptr, db->ws, (hipStream_t)stream, 1, lb, m);
}

int pfann_search_topk_bounded(pfann_db *db, const float *q, int64_t nq, int k, const float *lb, float *D, int64_t *I,
                              void *stream) {
    PF_HIP(hipSetDevice(db->device));
    if (nq > 16384) { set_error("pfann_search_topk_bounded: at most 16384 query rows per call (got %lld)", (long long)nq); return -1; }
    return search_topk(db->emb, (db->prefilter || db->emb == nullptr) ? db->emb_h : nullptr, db->xnorm_max, db->n, db->d,
                       db->label_base, q, nq, k, D, I, db->ws, (hipStream_t)stream, 2, const_cast<float *>(lb));
}

int pfann_topk_merge(pfann_db *db, const float *S, const int64_t *L, int64_t nq, int m, int k, float *D,
                     int64_t *I, void *stream) {
    PF_HIP(hipSetDevice(db->device));
    return topk_merge(S, L, nq, m, k, D, I, (hipStream_t)stream);
}

int pfann_bound_reduce(pfann_db *db, const float *cands_dev, int n_ranks, int64_t nq, int m, int k, float *lb_dev, void *stream) {
    PF_HIP(hipSetDevice(db->device));
    return bound_reduce(cands_dev, n_ranks, nq, m, k, lb_dev, (hipStream_t)stream);
}

int pfann_topk_merge_lists(pfann_db *db, const float *D_lists_dev, const int64_t *I_lists_dev, int n_lists, int64_t nq, int k,
                           float *D_dev, int64_t *I_dev, void *stream) {
    PF_HIP(hipSetDevice(db->device));
    return merge_lists(D_lists_dev, I_lists_dev, n_lists, nq, k, D_dev, I_dev, (hipStream_t)stream);
}

int pfann_match(pfann_db *db, const float *q, const int64_t *labels, int k, const int64_t *qstart,
                const int32_t *qlen, int64_t nQ, int max_qlen, int frame_shift_mul, float score_alpha, int mode,
                int only_owned, pfann_match_result *results, float *song_scores, void *stream) {
    PF_HIP(hipSetDevice(db->device));
    RerankArgs a;
    a.db = db->emb; a.dbh = db->emb_h; a.n = db->n; a.d = db->d; a.label_base = db->label_base;
    a.song_pos = db->song_pos; a.n_songs = db->n_songs; a.song_lo = db->song_lo; a.song_hi = db->song_hi;
    a.q = q; a.labels = labels; a.k = k; a.qstart = qstart; a.qlen = qlen; a.nQ = nQ;
    a.fsm = frame_shift_mul; a.alpha = score_alpha; a.mode = mode; a.only_owned = only_owned & 1;
    // bit 1 of only_owned (PFANN_MATCH_OWNED_BLOCK): song_scores is [nQ][owned songs][2] instead of [nQ][n_songs][2]
    a.ss_lo = 0; a.ss_n = db->n_songs;
    if (only_owned & 2) {
        if (!(only_owned & 1)) { set_error("match: the owned-songs score block needs only_owned candidates"); return -1; }
        a.ss_lo = db->song_lo; a.ss_n = db->song_hi - db->song_lo;
    }
    int P = 1;
    while (P < (int64_t)max_qlen * k) P <<= 1;
    a.pmax = P;
    a.gkeys = nullptr; a.gscore = nullptr; a.ncand = nullptr; a.phase = 0;
    // a handful of queries: spread each one's candidate scoring over the GPU (three launches: candidates, scores on all CUs,
    // argmax).  One workgroup per query -- the single-launch form -- leaves most of the chip idle below ~128 queries.
    // (round 6, tools/ubench/match_mid.py: 32 queries 0.54 -> 0.21 ms, 64 queries 0.54 -> 0.33, same decisions and scores; at
    // 128 queries the single launch is level, 0.52 vs 0.56; up to round 5 the limit was 16)
    static const int64_t phased_max = getenv("PFANN_MATCH_PHASED_MAX") ? atoll(getenv("PFANN_MATCH_PHASED_MAX")) : 64;
    const bool phased = nQ <= phased_max;
    if (P > 8192 || phased) {     // longer than the LDS candidate buffer, or phased: per-query slabs in HBM
        if ((int64_t)max_qlen * k > (1 << 22)) { set_error("match: query of %d rows x top_k %d is too long", max_qlen, k); return -1; }
        const size_t need = (size_t)nQ * P * 12 + (size_t)nQ * sizeof(int);
        if (db->match_scratch_bytes < need) {
            PF_HIP(hipStreamSynchronize((hipStream_t)stream));
            if (db->match_scratch) (void)hipFree(db->match_scratch);
            db->match_scratch = nullptr; db->match_scratch_bytes = 0;
            PF_HIP(hipMalloc(&db->match_scratch, need));
            db->match_scratch_bytes = need;
        }
        a.gkeys = reinterpret_cast<unsigned long long *>(db->match_scratch);
        a.gscore = reinterpret_cast<float *>(a.gkeys + (size_t)nQ * P);
        if (phased) { a.ncand = reinterpret_cast<int *>(a.gscore + (size_t)nQ * P); a.phase = 1; }
    }
    a.results = results; a.song_scores = song_scores;
    return launch_match(a, (hipStream_t)stream);
}

int pfann_song_scores_to_seconds(pfann_db *db, float *song_scores_dev, int64_t n_pairs, int frame_shift_mul, double hop_size,
                                 int native_path, void *stream) {
    PF_HIP(hipSetDevice(db->device));
    return launch_song_scores_to_seconds(song_scores_dev, n_pairs, frame_shift_mul, hop_size, native_path, (hipStream_t)stream);
}

int pfann_db_set_owned_songs(pfann_db *db, int song_lo, int song_hi) {
    // the caller's own cut (pfann_amd/dist.py: shard_songs): songs without rows at a shard boundary belong to whichever side
    // the CUT says, which the row range alone cannot tell
    if (song_lo < 0 || song_hi < song_lo || song_hi > db->n_songs || (size_t)db->n_songs + 1 != db->song_pos_h.size()) {
        set_error("db_set_owned_songs: [%d,%d) outside 0..%d", song_lo, song_hi, db->n_songs);
        return -1;
    }
    if (db->song_pos_h[song_lo] != db->label_base || db->song_pos_h[song_hi] != db->label_base + db->n) {
        set_error("db_set_owned_songs: songs [%d,%d) are rows [%lld,%lld), the shard holds [%lld,%lld)", song_lo, song_hi,
                  (long long)db->song_pos_h[song_lo], (long long)db->song_pos_h[song_hi], (long long)db->label_base,
                  (long long)(db->label_base + db->n));
        return -3;
    }
    db->song_lo = song_lo;
    db->song_hi = song_hi;
    return 0;
}

int pfann_db_owned_songs(pfann_db *db, int *song_lo, int *song_hi) {
    if (song_lo) *song_lo = db->song_lo;
    if (song_hi) *song_hi = db->song_hi;
    return db->song_hi - db->song_lo;
}

int pfann_match_pack(pfann_db *db, const pfann_match_result *results_dev, int64_t nQ, uint64_t *keys_dev, void *stream) {
    PF_HIP(hipSetDevice(db->device));
    return launch_match_pack(results_dev, nQ, reinterpret_cast<unsigned long long *>(keys_dev), (hipStream_t)stream);
}

int pfann_match_pick(pfann_db *db, const uint64_t *keys_dev, int n_ranks, int64_t nQ, pfann_match_result *out_dev, void *stream) {
    PF_HIP(hipSetDevice(db->device));
    if (n_ranks < 1) { set_error("pfann_match_pick: n_ranks < 1"); return -1; }
    return launch_match_pick(reinterpret_cast<const unsigned long long *>(keys_dev), n_ranks, nQ, out_dev, (hipStream_t)stream);
}

int seq_score(void *index, const int64_t *song_pos, int n_songs, const float *query, int query_len,
              const int64_t *labels, int top_k, float *song_scores, int frame_shift_mul, float score_alpha) {
    pfann_db *db = (pfann_db *)index;
    if (!db) { set_error("seq_score: null index"); return -1; }
    if (hipSetDevice(db->device) != hipSuccess) { set_error("seq_score: hipSetDevice failed"); return -1; }
    if (n_songs != db->n_songs || memcmp(song_pos, db->song_pos_h.data(), sizeof(int64_t) * (n_songs + 1)) != 0) {
        set_error("seq_score: song_pos differs from the one the database handle was loaded with");
        return -1;
    }
    if (query_len <= 0 || top_k <= 0) return -1;
    // One device slab and one pinned host image, kept in the handle between calls (grown on demand), laid out
    //   [song_scores f32 n_songs*2 | result | qstart i64 | qlen i32 | query f32 | labels i64]
    // Per call: host memcpy of the inputs into the pinned image, then on the handle's private stream 1 memset + 1 H2D
    // + the match launches + 1 D2H, and ONE wait for that stream.  No allocation, no pageable staging copies, no
    // null-stream synchronisation with whatever else the process has in flight.
    std::lock_guard<std::mutex> lock(db->seq_mu);
    const size_t nq = (size_t)query_len;
    const size_t ss_bytes = (size_t)std::max(n_songs, 1) * 2 * sizeof(float);
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_res = up16(ss_bytes), o_qs = o_res + up16(sizeof(pfann_match_result)), o_ql = o_qs + 16;
    const size_t o_q = o_ql + 16, o_l = o_q + up16(nq * db->d * sizeof(float));
    const size_t total = o_l + nq * top_k * sizeof(int64_t);
    if (!db->seq_stream && hipStreamCreateWithFlags(&db->seq_stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("seq_score: stream creation failed");
        return -1;
    }
    hipStream_t st = db->seq_stream;
    if (db->seq_scratch_bytes < total) {
        if (db->seq_scratch) { (void)hipStreamSynchronize(st); (void)hipFree(db->seq_scratch); }
        db->seq_scratch = nullptr; db->seq_scratch_bytes = 0;
        if (hipMalloc(&db->seq_scratch, total + (total >> 2)) != hipSuccess) { set_error("seq_score: device allocation failed"); return -1; }
        db->seq_scratch_bytes = total + (total >> 2);
    }
    if (db->seq_host_bytes < total) {
        if (db->seq_host) { (void)hipStreamSynchronize(st); (void)hipHostFree(db->seq_host); }
        db->seq_host = nullptr; db->seq_host_bytes = 0;
        if (hipHostMalloc(reinterpret_cast<void **>(&db->seq_host), total + (total >> 2), hipHostMallocDefault) != hipSuccess) {
            set_error("seq_score: pinned host allocation failed");
            return -1;
        }
        db->seq_host_bytes = total + (total >> 2);
    }
    char *dev = reinterpret_cast<char *>(db->seq_scratch);
    char *host = db->seq_host;
    const int64_t zero = 0;
    const int32_t ql = query_len;
    memcpy(host + o_qs, &zero, sizeof(zero));
    memcpy(host + o_ql, &ql, sizeof(ql));
    memcpy(host + o_q, query, nq * db->d * sizeof(float));
    memcpy(host + o_l, labels, nq * top_k * sizeof(int64_t));
    if (hipMemsetAsync(dev, 0, ss_bytes, st) != hipSuccess ||
        hipMemcpyAsync(dev + o_qs, host + o_qs, total - o_qs, hipMemcpyHostToDevice, st) != hipSuccess) {
        set_error("seq_score: upload failed");
        return -1;
    }
    if (pfann_match(db, reinterpret_cast<float *>(dev + o_q), reinterpret_cast<int64_t *>(dev + o_l), top_k,
                    reinterpret_cast<int64_t *>(dev + o_qs), reinterpret_cast<int32_t *>(dev + o_ql), 1, query_len,
                    frame_shift_mul, score_alpha, 1, 0, reinterpret_cast<pfann_match_result *>(dev + o_res),
                    reinterpret_cast<float *>(dev), st) != 0) return -1;
    if (hipMemcpyAsync(host, dev, o_qs, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { set_error("seq_score: download failed"); return -1; }
    pfann_match_result res;
    memcpy(&res, host + o_res, sizeof(res));
    if (res.song == -2) { set_error("seq_score: query_len*top_k too large for the candidate buffer"); return -1; }
    const float *ss = reinterpret_cast<const float *>(host);
    for (int s = 0; s < n_songs; ++s)            // seqscore.cpp:126-133 against the caller's slots
        if (ss[2 * s] > song_scores[2 * s]) { song_scores[2 * s] = ss[2 * s]; song_scores[2 * s + 1] = ss[2 * s + 1]; }
    return res.song;
}

// ---- profiling ------------------------------------------------------------------------
void pfann_prof_marker(void *stream) {
    PF_LAUNCH(pfann_bench_region_marker, dim3(1), dim3(64), 0, (hipStream_t)stream);
}
void pfann_prof_enable(int on) { g_prof = on != 0; }
void pfann_prof_reset(void) {
    for (auto &r : g_recs) { g_pool.push_back(r.e0); g_pool.push_back(r.e1); }
    g_recs.clear();
}
double pfann_prof_elapsed_ms(const char *tag, int64_t *count) {
    double tot = 0;
    int64_t n = 0;
    for (auto &r : g_recs) {
        if (r.tag != tag) continue;
        (void)hipEventSynchronize(r.e1);
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { tot += ms; ++n; }
    }
    if (count) *count = n;
    return tot;
}
double pfann_prof_work(const char *tag) {
    double w = 0;
    for (auto &r : g_recs) if (r.tag == tag) w += r.work;
    return w;
}
int pfann_prof_tags(char *out, int cap) {
    std::map<std::string, int> seen;
    for (auto &r : g_recs) seen[r.tag]++;
    std::string s;
    for (auto &kv : seen) { if (!s.empty()) s += ","; s += kv.first; }
    if ((int)s.size() + 1 > cap) return -1;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)seen.size();
}

}  // extern "C"
