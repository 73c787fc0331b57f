// Pre-processing chain of the reference (image_preprocessor.py::preprocess_for_book_cover), host side.
#include "ctx.h"

// ------------------------------------------------------------------------------------------------ pre-processing chain (f2)
// Host-side constants of the chain, computed exactly like oracle/preprocess.py (float32 where the C sources use float).  Every step only
// ENQUEUES on the slot's stream: the data-dependent tables (enhancer LUT from the plane's mean, CLAHE tile LUTs from the tile histograms)
// are computed by device kernels, so one page costs one wait at the end of the chain instead of one per step.
// cv2.resize INTER_CUBIC through IPP (see preproc.hip::pp_resize_cubic_kernel and oracle/preprocess.py::_cubic_axis_exact): phase
// t = n / (2 dst) exactly, weights as exact integers over K = 4 (2 dst)^3 and as correctly rounded doubles (both < 2^53)
struct CubicAxis { std::vector<int> first; std::vector<double> wf; std::vector<long long> wi; unsigned long long K; };
static CubicAxis cubic_axis(int dst, int src) {
    if (dst <= 0 || dst >= 65536 || src <= 0 || src >= (1 << 24)) fail(BBOCR_ERR_ARG, "cubic resize: plane too large for the exact weight tables");
    CubicAxis a;
    a.first.resize(dst);
    a.wf.resize((size_t)dst * 4);
    a.wi.resize((size_t)dst * 4);
    const __int128 D = 2 * (__int128)dst;
    const __int128 K = 4 * D * D * D;
    a.K = (unsigned long long)K;
    auto inner = [&](__int128 x) { return 5 * x * x * x - 9 * D * x * x + 4 * D * D * D; };                       // 4 D^3 ((A+2) t^3 - (A+3) t^2 + 1)
    auto outer = [&](__int128 x) { return -3 * x * x * x + 15 * D * x * x - 24 * D * D * x + 12 * D * D * D; };    // 4 D^3 (A t^3 - 5A t^2 + 8A t - 4A)
    for (int d = 0; d < dst; ++d) {
        const long long num = (2LL * d + 1) * src - dst;       // source coordinate = num / (2 dst)
        long long s = num / (long long)D;
        if (num < 0 && s * (long long)D != num) --s;           // floor
        const __int128 n = num - s * (long long)D;
        const __int128 c[4] = {outer(n + D), inner(n), inner(D - n), outer(2 * D - n)};
        a.first[d] = (int)s - 1;
        for (int k = 0; k < 4; ++k) {
            a.wi[(size_t)d * 4 + k] = (long long)c[k];
            a.wf[(size_t)d * 4 + k] = (double)(long long)c[k] / (double)a.K;
        }
    }
    return a;
}
void gaussian_taps3(double sigma, int k[3]) {   // getGaussianKernelBitExact -> 8.8 fixed point, error diffusion (sum 256)
    double v[3], tot = 0;
    for (int i = 0; i < 3; ++i) { v[i] = std::exp(-((double)(i - 1) * (i - 1)) / (2.0 * sigma * sigma)); tot += v[i]; }
    double err = 0;
    for (int i = 0; i < 3; ++i) {
        const double t = v[i] / tot * 256.0;
        const int r = (int)std::floor(t + err + 0.5);
        err += t - r;
        k[i] = r;
    }
}
static float pil_box_radius(float radius, int passes) {   // libImaging/BoxBlur.c::_gaussian_blur_radius
    const float sigma2 = radius * radius / (float)passes;
    const float L = (float)std::sqrt(12.0 * (double)sigma2 + 1.0);
    const float l = (float)std::floor(((double)L - 1.0) / 2.0);
    float a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
    a = a / (6 * (sigma2 - (l + 1) * (l + 1)));
    return l + a;
}

// Small device tables of the chain, at fixed offsets of pp_tab (every step of a chain is enqueued without a host round trip, so
// no two steps may share an offset): pixel sum | folded enhancer LUT | 64 tile histograms | 64 tile LUTs
enum : size_t { PP_SUM = 0, PP_LUT = 256, PP_HIST = 512, PP_TLUT = PP_HIST + 64 * 256 * 4, PP_TAB_BYTES = PP_TLUT + 64 * 256 };
static unsigned char* pp_tab(bbocr_ctx* c) {
    c->pp_tab.ensure(PP_TAB_BYTES);
    return (unsigned char*)c->pp_tab.p;
}

// Enqueues the resize; the weight tables stay on the device while the geometry repeats (one page size per batch is the rule), so
// only a change of (W, dw, H, dh) costs their computation, six uploads and a wait.
void pp_resize(bbocr_ctx* c, const uint8_t* src, int H, int W, uint8_t* dst, int dh, int dw, bool src_bgr) {
    const size_t nx = (size_t)dw, ny = (size_t)dh;
    const int key[4] = {W, dw, H, dh};
    const bool hit = c->pp_cubic.p && std::memcmp(key, c->pp_cubic_key, sizeof key) == 0;
    if (!hit) {
        const size_t bytes = (nx + ny) * (4 * 8 + 4 * 8) + (nx + ny) * 4 + 64;
        c->pp_cubic_key[1] = 0;                            // invalid until the uploads below have landed
        slot_sync(c, c->stream);                           // an earlier resize may still read the tables about to be replaced
        c->pp_cubic.ensure(bytes);
    }
    unsigned char* t = (unsigned char*)c->pp_cubic.p;     // 8-byte tables first (alignment), then the int tables
    double* wx = (double*)t;        t += nx * 32;
    double* wy = (double*)t;        t += ny * 32;
    long long* ix = (long long*)t;  t += nx * 32;
    long long* iy = (long long*)t;  t += ny * 32;
    int* x0 = (int*)t;              t += nx * 4;
    int* y0 = (int*)t;
    if (!hit) {
        const CubicAxis ax = cubic_axis(dw, W), ay = cubic_axis(dh, H);
        HIPCHK(hipMemcpyAsync(wx, ax.wf.data(), nx * 32, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(wy, ay.wf.data(), ny * 32, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(ix, ax.wi.data(), nx * 32, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(iy, ay.wi.data(), ny * 32, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(x0, ax.first.data(), nx * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(y0, ay.first.data(), ny * 4, hipMemcpyHostToDevice, c->stream));
        slot_sync(c, c->stream);                           // the host tables must outlive the copies
        c->pp_cubic_K[0] = ax.K;
        c->pp_cubic_K[1] = ay.K;
        std::memcpy(c->pp_cubic_key, key, sizeof key);
    }
    HIPCHK(launch_pp_resize_cubic(src, H, W, dst, dh, dw, x0, wx, ix, y0, wy, iy, c->pp_cubic_K[0], c->pp_cubic_K[1], c->stream, src_bgr ? 1 : 0));
}
// GaussianBlur 3x3 (sigma <= 0: the identity taps); leaves the sum of the output pixels at PP_SUM for the following Contrast step.  Enqueues.
void pp_gauss(bbocr_ctx* c, const uint8_t* src, int H, int W, uint8_t* dst, double sigma) {
    int k[3] = {0, 256, 0};
    if (sigma > 0) gaussian_taps3(sigma, k);
    unsigned char* t = pp_tab(c);
    HIPCHK(hipMemsetAsync(t + PP_SUM, 0, 8, c->stream));
    HIPCHK(launch_pp_gauss3(src, H, W, dst, k[0], k[1], k[2], (unsigned long long*)(t + PP_SUM), c->stream));
}
// The two PIL enhancers as one LUT at PP_LUT, computed on the device from the sum at PP_SUM (a factor <= 0 skips its step; both <= 0
// give the identity).  Enqueues.
const uint8_t* pp_fold_lut(bbocr_ctx* c, size_t n, double contrast, double brightness) {
    unsigned char* t = pp_tab(c);
    HIPCHK(launch_pp_fold_lut((const unsigned long long*)(t + PP_SUM), (unsigned long long)n, (float)contrast, (float)brightness, t + PP_LUT,
                              c->stream));
    return t + PP_LUT;
}
// CLAHE of lut[src] (lut: device table from pp_fold_lut).  Histograms, clip / redistribute / cumulative LUTs and the interpolated
// lookup are three launches on the stream, nothing comes back to the host.  Enqueues.
void pp_clahe(bbocr_ctx* c, const uint8_t* src, int H, int W, const uint8_t* d_lut, uint8_t* dst, double clip_limit) {
    const int tx = 8, ty = 8;
    // clahe.cpp pads BOTH axes by tiles - (size % tiles) as soon as ONE of them is not a multiple of the grid -- a whole extra
    // 8 rows / columns on the axis that did divide (upstream quirk, restated as is)
    const bool pad = (H % ty) || (W % tx);
    const int EH = pad ? H + (ty - H % ty) : H, EW = pad ? W + (tx - W % tx) : W;
    if (EH - H >= H || EW - W >= W) fail(BBOCR_ERR_ARG, "image smaller than the CLAHE tile grid");
    const int th = EH / ty, tw = EW / tx;
    unsigned char* t = pp_tab(c);
    unsigned int* d_hist = (unsigned int*)(t + PP_HIST);
    uint8_t* d_tl = t + PP_TLUT;
    HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)tx * ty * 256 * 4, c->stream));
    HIPCHK(launch_pp_clahe_hist(src, H, W, d_lut, tw, th, tx, ty, d_hist, c->stream));
    const int area = th * tw;
    const float lut_scale = 255.0f / (float)area;
    int clip = 0;
    if (clip_limit > 0) clip = std::max((int)(clip_limit * area / 256), 1);
    HIPCHK(launch_pp_clahe_luts(d_hist, tx * ty, clip, lut_scale, d_tl, c->stream));
    HIPCHK(launch_pp_clahe_apply(src, H, W, d_lut, d_tl, tw, th, tx, ty, dst, c->stream));
}
// PIL UnsharpMask on src -> dst; tmp1/tmp2: two scratch planes of the same size
void pp_unsharp(bbocr_ctx* c, const uint8_t* src, int H, int W, uint8_t* dst, uint8_t* tmp1, uint8_t* tmp2, float radius, int percent,
                       int threshold) {
    const float fr = pil_box_radius(radius, 3);
    const int r = (int)fr;
    const unsigned int ww = (unsigned int)((float)(1 << 24) / (fr * 2.f + 1.f));
    const unsigned int fw = ((1u << 24) - (unsigned int)(r * 2 + 1) * ww) / 2;
    if (pp_unsharp_fused_ok(H, W, r, src, tmp1, dst)) {       // the chain's own case (radius 1 -> box radius 0): 2 launches instead of 7
        // (tmp = nullptr: the row passes applied on the fly inside the column kernel, ONE launch -- measured equal: 130 us against 67 + 63)
        HIPCHK(launch_pp_unsharp_fused(src, tmp1, dst, H, W, ww, fw, percent, threshold, c->stream));
        return;
    }
    const uint8_t* cur = src;
    uint8_t* bufs[2] = {tmp1, tmp2};
    int w = 0;
    for (int pass = 0; pass < 6; ++pass) {
        HIPCHK(launch_pp_box_pass(cur, bufs[w], H, W, pass >= 3, r, ww, fw, c->stream));
        cur = bufs[w];
        w ^= 1;
    }
    HIPCHK(launch_pp_unsharp(src, cur, dst, (size_t)H * W, percent, threshold, c->stream));
}

void preprocess_chain_impl(bbocr_ctx* c, const uint8_t* bgr, int H, int W, const bbocr_preproc_params& q, uint8_t* out, int dh, int dw) {
    const size_t n = (size_t)dh * dw;
    c->pp_gray.ensure((size_t)H * W);
    c->pp_a.ensure(n);
    c->pp_b.ensure(n);
    c->pp_c.ensure(n);
    uint8_t *g = (uint8_t*)c->pp_gray.p, *bufs[3] = {(uint8_t*)c->pp_a.p, (uint8_t*)c->pp_b.p, (uint8_t*)c->pp_c.p};
    // channels as given: B 3735, G 19235, R 9798 (>> 15) for cv2.imread's BGR.  When the LDS-tiled resize follows, the conversion rides in its
    // window load and the source-size gray plane is never written
    const bool gray_in_resize = q.scale > 0 && pp_resize_tile_rows(H, W, dh, dw) != 0;
    if (!gray_in_resize) HIPCHK(launch_gray(bgr, g, (size_t)H * W, c->stream));
    const uint8_t* cur = g;
    int nb = 0;                                                        // next free plane of the three
    auto next = [&]() { uint8_t* b = bufs[nb]; nb = (nb + 1) % 3; return b; };
    if (q.scale > 0) {
        uint8_t* d = next();
        pp_resize(c, gray_in_resize ? bgr : cur, H, W, d, dh, dw, gray_in_resize);
        cur = d;
    }
    // the mean ImageEnhance.Contrast needs: the blur kernel sums its own output; without a blur stage the identity taps (0, 256, 0) do
    bool have_sum = false;
    if (q.blur_sigma > 0) {
        uint8_t* d = next();
        pp_gauss(c, cur, dh, dw, d, q.blur_sigma);
        have_sum = true;
        cur = d;
    }
    if (q.contrast > 0 && !have_sum) {
        uint8_t* d = next();
        pp_gauss(c, cur, dh, dw, d, 0.0);
        cur = d;
    }
    // the two PIL enhancers are pointwise: folded into one LUT, applied in front of CLAHE (or on their own when CLAHE is skipped)
    const bool have_lut = q.contrast > 0 || q.brightness > 0;
    if (q.clahe_clip > 0) {
        uint8_t* d = next();
        pp_clahe(c, cur, dh, dw, pp_fold_lut(c, n, q.contrast, q.brightness), d, q.clahe_clip);
        cur = d;
    } else if (have_lut) {
        uint8_t* d = next();
        HIPCHK(launch_pp_lut(cur, d, pp_fold_lut(c, n, q.contrast, q.brightness), n, c->stream));
        cur = d;
    }
    if (q.unsharp_percent > 0 && q.unsharp_radius > 0) {
        uint8_t *t1 = next(), *t2 = next();                            // `cur` is the third plane
        pp_unsharp(c, cur, dh, dw, out, t1, t2, (float)q.unsharp_radius, q.unsharp_percent, q.unsharp_threshold);
    } else {
        HIPCHK(hipMemcpyAsync(out, cur, n, hipMemcpyDeviceToDevice, c->stream));
    }
    slot_sync(c, c->stream);
}
