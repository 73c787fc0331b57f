// Pre-processing chain of the reference (image_preprocessor.py::preprocess_for_book_cover), host side.
#include "ctx.h"

// ------------------------------------------------------------------------------------------------ pre-processing chain (f2)
// Host-side constants of the chain, computed exactly like oracle/preprocess.py (float32 where the C sources use float).
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
void pil_blend_lut(int in1, float alpha, uint8_t lut[256]) {   // libImaging/Blend.c with a constant first image
    for (int v = 0; v < 256; ++v) {
        const float t = (float)in1 + alpha * (float)(v - in1);
        if (alpha >= 0.f && alpha <= 1.f) lut[v] = (uint8_t)t;
        else lut[v] = t <= 0.f ? 0 : (t >= 255.f ? 255 : (uint8_t)t);
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

void pp_resize(bbocr_ctx* c, const uint8_t* src, int H, int W, uint8_t* dst, int dh, int dw) {
    const CubicAxis ax = cubic_axis(dw, W), ay = cubic_axis(dh, H);
    const size_t nx = (size_t)dw, ny = (size_t)dh;
    const size_t bytes = (nx + ny) * (4 * 8 + 4 * 8) + (nx + ny) * 4 + 64;
    c->pp_tab.ensure(bytes);
    unsigned char* t = (unsigned char*)c->pp_tab.p;       // 8-byte tables first (alignment), then the int tables
    double* wx = (double*)t;        t += nx * 32;
    double* wy = (double*)t;        t += ny * 32;
    long long* ix = (long long*)t;  t += nx * 32;
    long long* iy = (long long*)t;  t += ny * 32;
    int* x0 = (int*)t;              t += nx * 4;
    int* y0 = (int*)t;
    HIPCHK(hipMemcpyAsync(wx, ax.wf.data(), nx * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(wy, ay.wf.data(), ny * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(ix, ax.wi.data(), nx * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(iy, ay.wi.data(), ny * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(x0, ax.first.data(), nx * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(y0, ay.first.data(), ny * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_pp_resize_cubic(src, H, W, dst, dh, dw, x0, wx, ix, y0, wy, iy, ax.K, ay.K, c->stream));
    slot_sync(c, c->stream);   // the host tables must outlive the copies
}
// GaussianBlur 3x3; returns the sum of the output pixels (for the following Contrast step)
unsigned long long pp_gauss(bbocr_ctx* c, const uint8_t* src, int H, int W, uint8_t* dst, double sigma) {
    int k[3];
    gaussian_taps3(sigma, k);
    c->pp_tab.ensure(64);
    HIPCHK(hipMemsetAsync(c->pp_tab.p, 0, 8, c->stream));
    HIPCHK(launch_pp_gauss3(src, H, W, dst, k[0], k[1], k[2], (unsigned long long*)c->pp_tab.p, c->stream));
    unsigned long long sum = 0;
    HIPCHK(hipMemcpyAsync(&sum, c->pp_tab.p, 8, hipMemcpyDeviceToHost, c->stream));
    slot_sync(c, c->stream);
    return sum;
}
// CLAHE of lut[src] (lut = pointwise steps folded in front of it; identity if null)
void pp_clahe(bbocr_ctx* c, const uint8_t* src, int H, int W, const uint8_t* lut_host, uint8_t* dst, double clip_limit) {
    const int tx = 8, ty = 8;
    // clahe.cpp pads BOTH axes by tiles - (size % tiles) as soon as ONE of them is not a multiple of the grid -- a whole extra
    // 8 rows / columns on the axis that did divide (upstream quirk, restated as is)
    const bool pad = (H % ty) || (W % tx);
    const int EH = pad ? H + (ty - H % ty) : H, EW = pad ? W + (tx - W % tx) : W;
    if (EH - H >= H || EW - W >= W) fail(BBOCR_ERR_ARG, "image smaller than the CLAHE tile grid");
    const int th = EH / ty, tw = EW / tx;
    c->pp_tab.ensure(256 + (size_t)tx * ty * 256 * 4 + (size_t)tx * ty * 256);
    uint8_t* d_lut = (uint8_t*)c->pp_tab.p;
    unsigned int* d_hist = (unsigned int*)((unsigned char*)c->pp_tab.p + 256);
    uint8_t* d_tl = (uint8_t*)c->pp_tab.p + 256 + (size_t)tx * ty * 256 * 4;
    uint8_t ident[256];
    for (int i = 0; i < 256; ++i) ident[i] = (uint8_t)i;
    HIPCHK(hipMemcpyAsync(d_lut, lut_host ? lut_host : ident, 256, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)tx * ty * 256 * 4, c->stream));
    HIPCHK(launch_pp_clahe_hist(src, H, W, d_lut, tw, th, tx, ty, d_hist, c->stream));
    std::vector<unsigned int> hist((size_t)tx * ty * 256);
    HIPCHK(hipMemcpyAsync(hist.data(), d_hist, hist.size() * 4, hipMemcpyDeviceToHost, c->stream));
    slot_sync(c, c->stream);
    // imgproc clahe.cpp: clip + redistribute, LUT = cvRound(cumsum * 255 / tile_area) in float
    const int area = th * tw;
    const float lut_scale = 255.0f / (float)area;
    int clip = 0;
    if (clip_limit > 0) clip = std::max((int)(clip_limit * area / 256), 1);
    std::vector<uint8_t> tl((size_t)tx * ty * 256);
    for (int t = 0; t < tx * ty; ++t) {
        long long h[256];
        for (int i = 0; i < 256; ++i) h[i] = hist[(size_t)t * 256 + i];
        if (clip > 0) {
            long long clipped = 0;
            for (int i = 0; i < 256; ++i) if (h[i] > clip) { clipped += h[i] - clip; h[i] = clip; }
            const long long batch = clipped / 256;
            long long residual = clipped - batch * 256;
            for (int i = 0; i < 256; ++i) h[i] += batch;
            if (residual) {
                const int step = std::max((int)(256 / residual), 1);
                for (int i = 0; i < 256 && residual > 0; i += step, --residual) h[i] += 1;
            }
        }
        long long sum = 0;
        for (int i = 0; i < 256; ++i) {
            sum += h[i];
            const long v = std::lrintf((float)sum * lut_scale);
            tl[(size_t)t * 256 + i] = (uint8_t)std::max<long>(0, std::min<long>(255, v));
        }
    }
    HIPCHK(hipMemcpyAsync(d_tl, tl.data(), tl.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_pp_clahe_apply(src, H, W, d_lut, d_tl, tw, th, tx, ty, dst, c->stream));
    slot_sync(c, c->stream);
}
// PIL UnsharpMask on src -> dst; tmp1/tmp2: two scratch planes of the same size
void pp_unsharp(bbocr_ctx* c, const uint8_t* src, int H, int W, uint8_t* dst, uint8_t* tmp1, uint8_t* tmp2, float radius, int percent,
                       int threshold) {
    const float fr = pil_box_radius(radius, 3);
    const int r = (int)fr;
    const unsigned int ww = (unsigned int)((float)(1 << 24) / (fr * 2.f + 1.f));
    const unsigned int fw = ((1u << 24) - (unsigned int)(r * 2 + 1) * ww) / 2;
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
    HIPCHK(launch_gray(bgr, g, (size_t)H * W, c->stream));             // channels as given: B 3735, G 19235, R 9798 (>> 15) for cv2.imread's BGR
    const uint8_t* cur = g;
    int nb = 0;                                                        // next free plane of the three
    auto next = [&]() { uint8_t* b = bufs[nb]; nb = (nb + 1) % 3; return b; };
    if (q.scale > 0) {
        uint8_t* d = next();
        pp_resize(c, cur, H, W, d, dh, dw);
        cur = d;
    }
    // the mean ImageEnhance.Contrast needs: the blur kernel sums its own output; without a blur stage the identity taps (0, 256, 0) do
    unsigned long long sum = 0;
    bool have_sum = false;
    if (q.blur_sigma > 0) {
        uint8_t* d = next();
        sum = pp_gauss(c, cur, dh, dw, d, q.blur_sigma);
        have_sum = true;
        cur = d;
    }
    // the two PIL enhancers are pointwise: folded into one LUT, applied in front of CLAHE (or on their own when CLAHE is skipped)
    uint8_t lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)i;
    bool have_lut = false;
    if (q.contrast > 0) {
        if (!have_sum) {
            c->pp_tab.ensure(64);
            HIPCHK(hipMemsetAsync(c->pp_tab.p, 0, 8, c->stream));
            uint8_t* d = next();
            HIPCHK(launch_pp_gauss3(cur, dh, dw, d, 0, 256, 0, (unsigned long long*)c->pp_tab.p, c->stream));
            HIPCHK(hipMemcpyAsync(&sum, c->pp_tab.p, 8, hipMemcpyDeviceToHost, c->stream));
            slot_sync(c, c->stream);
            cur = d;
        }
        uint8_t l1[256];
        pil_blend_lut((int)((double)sum / (double)n + 0.5), (float)q.contrast, l1);
        for (int i = 0; i < 256; ++i) lut[i] = l1[lut[i]];
        have_lut = true;
    }
    if (q.brightness > 0) {
        uint8_t l2[256];
        pil_blend_lut(0, (float)q.brightness, l2);
        for (int i = 0; i < 256; ++i) lut[i] = l2[lut[i]];
        have_lut = true;
    }
    if (q.clahe_clip > 0) {
        uint8_t* d = next();
        pp_clahe(c, cur, dh, dw, have_lut ? lut : nullptr, d, q.clahe_clip);
        cur = d;
    } else if (have_lut) {
        uint8_t* d = next();
        c->pp_tab.ensure(512);
        HIPCHK(hipMemcpyAsync((unsigned char*)c->pp_tab.p + 256, lut, 256, hipMemcpyHostToDevice, c->stream));
        HIPCHK(launch_pp_lut(cur, d, (const uint8_t*)c->pp_tab.p + 256, n, c->stream));
        slot_sync(c, c->stream);
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
