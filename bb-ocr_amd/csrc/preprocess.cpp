// Pre-processing chain of the reference (image_preprocessor.py::preprocess_for_book_cover), host side.
#include "ctx.h"

// ------------------------------------------------------------------------------------------------ pre-processing chain (f2)
// Host-side constants of the chain, computed exactly like oracle/preprocess.py (float32 where the C sources use float).
struct CubicAxis { std::vector<int> first; std::vector<short> coef; };
static CubicAxis cubic_axis(int dst, int src) {   // imgproc resize.cpp: fx in float, cvFloor, interpolateCubic (A = -0.75), cvRound(c * 2048)
    CubicAxis a;
    a.first.resize(dst);
    a.coef.resize((size_t)dst * 4);
    const double scale = (double)src / (double)dst;
    const float A = -0.75f;
    for (int d = 0; d < dst; ++d) {
        float f = (float)(((double)d + 0.5) * scale - 0.5);
        const int s = (int)std::floor(f);
        f = f - (float)s;
        float c[4];
        const float x = f;
        c[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
        c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
        c[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
        c[3] = 1.f - c[0] - c[1] - c[2];
        a.first[d] = s - 1;
        for (int k = 0; k < 4; ++k) {
            long v = std::lrint((double)c[k] * 2048.0);        // cvRound: half to even
            a.coef[(size_t)d * 4 + k] = (short)std::max<long>(-32768, std::min<long>(32767, v));
        }
    }
    return a;
}
static void gaussian_taps3(double sigma, int k[3]) {   // getGaussianKernelBitExact -> 8.8 fixed point, error diffusion (sum 256)
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
    const size_t bytes = (size_t)dw * 4 + (size_t)dw * 8 + (size_t)dh * 4 + (size_t)dh * 8;
    c->pp_tab.ensure(bytes);
    unsigned char* t = (unsigned char*)c->pp_tab.p;
    int* x0 = (int*)t;                         t += (size_t)dw * 4;
    int* y0 = (int*)t;                         t += (size_t)dh * 4;
    short* cx = (short*)t;                     t += (size_t)dw * 8;
    short* cy = (short*)t;
    HIPCHK(hipMemcpyAsync(x0, ax.first.data(), (size_t)dw * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(y0, ay.first.data(), (size_t)dh * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(cx, ax.coef.data(), (size_t)dw * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(cy, ay.coef.data(), (size_t)dh * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_pp_resize_cubic(src, H, W, dst, dh, dw, x0, cx, y0, cy, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));   // the host tables must outlive the copies
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
    HIPCHK(hipStreamSynchronize(c->stream));
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
    HIPCHK(hipStreamSynchronize(c->stream));
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
    HIPCHK(hipStreamSynchronize(c->stream));
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

void preprocess_book_cover_impl(bbocr_ctx* c, const uint8_t* bgr, int H, int W, uint8_t* out, int dh, int dw) {
    const size_t n = (size_t)dh * dw;
    c->pp_gray.ensure((size_t)H * W);
    c->pp_a.ensure(n);
    c->pp_b.ensure(n);
    c->pp_c.ensure(n);
    uint8_t *g = (uint8_t*)c->pp_gray.p, *a = (uint8_t*)c->pp_a.p, *b = (uint8_t*)c->pp_b.p, *cc = (uint8_t*)c->pp_c.p;
    HIPCHK(launch_gray(bgr, g, (size_t)H * W, c->stream));             // channels as given: B 3735, G 19235, R 9798 (>> 15) for cv2.imread's BGR
    pp_resize(c, g, H, W, a, dh, dw);
    const unsigned long long sum = pp_gauss(c, a, dh, dw, b, 3.0);
    // ImageEnhance.Contrast(1.9) then Brightness(1.2): two pointwise maps, folded into one LUT in front of CLAHE
    const int mean = (int)((double)sum / (double)n + 0.5);
    uint8_t l1[256], l2[256], lut[256];
    pil_blend_lut(mean, 1.9f, l1);
    pil_blend_lut(0, 1.2f, l2);
    for (int i = 0; i < 256; ++i) lut[i] = l2[l1[i]];
    pp_clahe(c, b, dh, dw, lut, a, 2.5);
    pp_unsharp(c, a, dh, dw, out, b, cc, 1.0f, 30, 3);
    HIPCHK(hipStreamSynchronize(c->stream));
}
