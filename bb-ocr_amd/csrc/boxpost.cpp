// Host-side box geometry: component row-extremes -> dilated hull -> min-area rectangle -> text polygons -> grouped boxes.
//
// Restates the per-label tail of easyocr/craft_utils.py::getDetBoxes_core (cv2.dilate with a (1+niter)^2 rectangle,
// cv2.minAreaRect, cv2.boxPoints, the "diamond" fix and clockwise start), adjustResultCoordinates,
// detection.py::get_textbox (int32 cast), utils.py::group_text_box and the min_size filter of Reader.detect, for the
// reference call reader.readtext(...) (pipeline_demo/extractor/enhanced_extractor.py:520).  Work here is O(#rows of
// accepted components); the pixel passes live in ccl.hip.  Float steps are written operation by operation in float32
// (no contraction: this file is compiled with -ffp-contract=off) so they match the numpy restatement in oracle/boxes.py.
#include "boxpost.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace bbocr {

struct P2 { long long x, y; };

static long long cross(const P2& o, const P2& a, const P2& b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

// strictly convex hull, starting at the right-most point (max x, then max y), then max-y, min-x, min-y side
// (== cv2.convexHull(points, clockwise=False) order)
static void convex_hull(std::vector<P2>& pts, std::vector<P2>& hull) {
    std::sort(pts.begin(), pts.end(), [](const P2& a, const P2& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    pts.erase(std::unique(pts.begin(), pts.end(), [](const P2& a, const P2& b) { return a.x == b.x && a.y == b.y; }), pts.end());
    const size_t n = pts.size();
    hull.clear();
    if (n <= 1) { hull = pts; return; }
    if (n == 2) { hull.push_back(pts[1]); hull.push_back(pts[0]); return; }
    std::vector<P2> lower, upper;
    for (size_t i = 0; i < n; ++i) {
        while (lower.size() >= 2 && cross(lower[lower.size() - 2], lower.back(), pts[i]) <= 0) lower.pop_back();
        lower.push_back(pts[i]);
    }
    for (size_t i = n; i-- > 0;) {
        while (upper.size() >= 2 && cross(upper[upper.size() - 2], upper.back(), pts[i]) <= 0) upper.pop_back();
        upper.push_back(pts[i]);
    }
    for (size_t i = 0; i + 1 < upper.size(); ++i) hull.push_back(upper[i]);
    for (size_t i = 0; i + 1 < lower.size(); ++i) hull.push_back(lower[i]);
}

// rotcalipers.cpp::rotatingCalipers(CALIPERS_MINAREARECT) -> out[0]=corner, out[1], out[2] = side vectors
static bool rotating_calipers(const std::vector<P2>& hp, float out[6]) {
    const int n = (int)hp.size();
    std::vector<float> px(n), py(n), vx(n), vy(n), inv(n);
    for (int i = 0; i < n; ++i) { px[i] = (float)hp[i].x; py[i] = (float)hp[i].y; }
    int left = 0, bottom = 0, right = 0, top = 0;
    float left_x = px[0], right_x = px[0], top_y = py[0], bottom_y = py[0];
    for (int i = 0; i < n; ++i) {
        if (px[i] < left_x) { left_x = px[i]; left = i; }
        if (px[i] > right_x) { right_x = px[i]; right = i; }
        if (py[i] > top_y) { top_y = py[i]; top = i; }
        if (py[i] < bottom_y) { bottom_y = py[i]; bottom = i; }
        const int j = i + 1 < n ? i + 1 : 0;
        const double dx = (double)px[j] - (double)px[i], dy = (double)py[j] - (double)py[i];
        vx[i] = (float)dx;
        vy[i] = (float)dy;
        inv[i] = (float)(1. / std::sqrt(dx * dx + dy * dy));
    }
    float orientation = 0.f;
    {
        double ax = vx[n - 1], ay = vy[n - 1];
        for (int i = 0; i < n; ++i) {
            const double bx = vx[i], by = vy[i];
            const double convexity = ax * by - ay * bx;
            if (convexity != 0) { orientation = convexity > 0 ? 1.f : -1.f; break; }
            ax = bx; ay = by;
        }
    }
    if (orientation == 0.f) return false;
    float base_a = orientation, base_b = 0.f;
    int seq[4] = {bottom, right, top, left};
    float minarea = 3.402823466e+38F;
    int b_left = 0, b_bottom = 0;
    float b_a = 0, b_w = 0, b_b = 0, b_h = 0;
    for (int k = 0; k < n; ++k) {
        float dp[4];
        { const float t0 = base_a * vx[seq[0]], t1 = base_b * vy[seq[0]]; dp[0] = t0 + t1; }
        { const float t0 = -base_b * vx[seq[1]], t1 = base_a * vy[seq[1]]; dp[1] = t0 + t1; }
        { const float t0 = -base_a * vx[seq[2]], t1 = base_b * vy[seq[2]]; dp[2] = t0 - t1; }
        { const float t0 = base_b * vx[seq[3]], t1 = base_a * vy[seq[3]]; dp[3] = t0 - t1; }
        float maxcos = dp[0] * inv[seq[0]];
        int main_element = 0;
        for (int i = 1; i < 4; ++i) {
            const float cosalpha = dp[i] * inv[seq[i]];
            if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
        }
        {
            const int pindex = seq[main_element];
            const float lead_x = vx[pindex] * inv[pindex], lead_y = vy[pindex] * inv[pindex];
            switch (main_element) {
                case 0: base_a = lead_x; base_b = lead_y; break;
                case 1: base_a = lead_y; base_b = -lead_x; break;
                case 2: base_a = -lead_x; base_b = -lead_y; break;
                default: base_a = -lead_y; base_b = lead_x; break;
            }
        }
        seq[main_element] += 1;
        if (seq[main_element] == n) seq[main_element] = 0;
        float dx = px[seq[1]] - px[seq[3]], dy = py[seq[1]] - py[seq[3]];
        float width;
        { const float t0 = dx * base_a, t1 = dy * base_b; width = t0 + t1; }
        dx = px[seq[2]] - px[seq[0]];
        dy = py[seq[2]] - py[seq[0]];
        float height;
        { const float t0 = -dx * base_b, t1 = dy * base_a; height = t0 + t1; }
        const float area = width * height;
        if (area <= minarea) {
            minarea = area;
            b_left = seq[3]; b_a = base_a; b_w = width; b_b = base_b; b_h = height; b_bottom = seq[0];
        }
    }
    const float A1 = b_a, B1 = b_b, A2 = -b_b, B2 = b_a;
    float C1, C2;
    { const float t0 = A1 * px[b_left], t1 = py[b_left] * B1; C1 = t0 + t1; }
    { const float t0 = A2 * px[b_bottom], t1 = py[b_bottom] * B2; C2 = t0 + t1; }
    float idet;
    { const float t0 = A1 * B2, t1 = A2 * B1; idet = 1.f / (t0 - t1); }
    { const float t0 = C1 * B2, t1 = C2 * B1; out[0] = (t0 - t1) * idet; }
    { const float t0 = A1 * C2, t1 = A2 * C1; out[1] = (t0 - t1) * idet; }
    out[2] = A1 * b_w; out[3] = B1 * b_w;
    out[4] = A2 * b_h; out[5] = B2 * b_h;
    return true;
}

// cv2.minAreaRect + cv2.boxPoints on a point set given by its hull
static void min_area_box(std::vector<P2>& pts, float box[4][2]) {
    std::vector<P2> hull;
    convex_hull(pts, hull);
    const int n = (int)hull.size();
    float cx = 0, cy = 0, w = 0, h = 0, ang = 0;
    float o[6];
    if (n > 2 && rotating_calipers(hull, o)) {
        { const float t = o[2] + o[4]; cx = o[0] + t * 0.5f; }
        { const float t = o[3] + o[5]; cy = o[1] + t * 0.5f; }
        w = (float)std::sqrt((double)o[2] * o[2] + (double)o[3] * o[3]);
        h = (float)std::sqrt((double)o[4] * o[4] + (double)o[5] * o[5]);
        ang = (float)std::atan2((double)o[3], (double)o[2]);
    } else if (n == 2) {
        { const float t = (float)hull[0].x + (float)hull[1].x; cx = t * 0.5f; }
        { const float t = (float)hull[0].y + (float)hull[1].y; cy = t * 0.5f; }
        const double dx = (double)hull[1].x - (double)hull[0].x, dy = (double)hull[1].y - (double)hull[0].y;
        w = (float)std::sqrt(dx * dx + dy * dy);
        ang = (float)std::atan2(dy, dx);
    } else if (n >= 1) {
        cx = (float)hull[0].x; cy = (float)hull[0].y;
    }
    ang = (float)((double)ang * 180.0 / M_PI);
    const double _angle = (double)ang * M_PI / 180.0;
    const float b = (float)std::cos(_angle) * 0.5f;
    const float a = (float)std::sin(_angle) * 0.5f;
    { const float t0 = a * h, t1 = b * w; box[0][0] = (cx - t0) - t1; }
    { const float t0 = b * h, t1 = a * w; box[0][1] = (cy + t0) - t1; }
    { const float t0 = a * h, t1 = b * w; box[1][0] = (cx + t0) - t1; }
    { const float t0 = b * h, t1 = a * w; box[1][1] = (cy - t0) - t1; }
    box[2][0] = 2.f * cx - box[0][0];
    box[2][1] = 2.f * cy - box[0][1];
    box[3][0] = 2.f * cx - box[1][0];
    box[3][1] = 2.f * cy - box[1][1];
}

void component_box(const Component& c, const int* rowext, int img_w, int img_h, float box[4][2]) {
    const int w = c.right - c.left + 1, h = c.bottom - c.top + 1;
    const int niter = (int)(std::sqrt((double)((long long)c.area * std::min(w, h)) / (double)((long long)w * h)) * 2);
    const int lo = niter / 2, hi = (niter + 1) / 2;   // reach of the (1+niter)^2 rectangle, centre anchor
    std::vector<P2> pts;
    long long gl = 1LL << 40, gr = -1, gt = 1LL << 40, gb = -1;
    const int y_first = std::max(0, c.top - lo), y_last = std::min(img_h - 1, c.bottom + hi);
    for (int Y = y_first; Y <= y_last; ++Y) {
        int mn = 0x7fffffff, mx = -1;
        const int s0 = std::max(c.top, Y - hi), s1 = std::min(c.bottom, Y + lo);
        for (int y = s0; y <= s1; ++y) {
            const int a = rowext[2 * (y - c.top)], b = rowext[2 * (y - c.top) + 1];
            if (b < 0) continue;
            mn = std::min(mn, a);
            mx = std::max(mx, b);
        }
        if (mx < 0) continue;
        const int x0 = std::max(0, mn - lo), x1 = std::min(img_w - 1, mx + hi);
        pts.push_back({x0, Y});
        pts.push_back({x1, Y});
        gl = std::min<long long>(gl, x0); gr = std::max<long long>(gr, x1);
        gt = std::min<long long>(gt, Y); gb = std::max<long long>(gb, Y);
    }
    min_area_box(pts, box);
    // np.linalg.norm on float32 rows, then numpy-1.26 scalar promotion to float64 for the ratio
    float wn, hn;
    { const float dx = box[0][0] - box[1][0], dy = box[0][1] - box[1][1]; const float t0 = dx * dx, t1 = dy * dy; wn = std::sqrt(t0 + t1); }
    { const float dx = box[1][0] - box[2][0], dy = box[1][1] - box[2][1]; const float t0 = dx * dx, t1 = dy * dy; hn = std::sqrt(t0 + t1); }
    const double wd = wn, hd = hn;
    const double box_ratio = std::max(wd, hd) / (std::min(wd, hd) + 1e-5);
    if (std::fabs(1 - box_ratio) <= 0.1) {
        box[0][0] = (float)gl; box[0][1] = (float)gt;
        box[1][0] = (float)gr; box[1][1] = (float)gt;
        box[2][0] = (float)gr; box[2][1] = (float)gb;
        box[3][0] = (float)gl; box[3][1] = (float)gb;
    }
    int start = 0;
    float best = box[0][0] + box[0][1];
    for (int i = 1; i < 4; ++i) {
        const float s = box[i][0] + box[i][1];
        if (s < best) { best = s; start = i; }
    }
    float tmp[4][2];
    for (int i = 0; i < 4; ++i) { tmp[i][0] = box[(i + start) & 3][0]; tmp[i][1] = box[(i + start) & 3][1]; }
    std::memcpy(box, tmp, sizeof(tmp));
}

void box_to_poly(const float box[4][2], double ratio_w, double ratio_h, int poly[8]) {
    for (int i = 0; i < 4; ++i) {
        const float x = (float)((double)box[i][0] * (ratio_w * 2));
        const float y = (float)((double)box[i][1] * (ratio_h * 2));
        poly[2 * i] = (int)x;       // astype(np.int32): truncation toward zero
        poly[2 * i + 1] = (int)y;
    }
}

static double mean_of(const std::vector<double>& v) {
    double s = 0;
    for (double x : v) s += x;
    return s / (double)v.size();
}

void group_text_box(const std::vector<std::array<int, 8>>& polys, const GroupParams& gp, std::vector<std::array<int, 4>>& merged_list,
                    std::vector<std::array<double, 8>>& free_list) {
    struct HB { int x_min, x_max, y_min, y_max; double yc; int hgt; };
    std::vector<HB> horizontal_list;
    merged_list.clear();
    free_list.clear();
    for (const auto& poly : polys) {
        const double slope_up = (double)(poly[3] - poly[1]) / (double)std::max(10, poly[2] - poly[0]);
        const double slope_down = (double)(poly[5] - poly[7]) / (double)std::max(10, poly[4] - poly[6]);
        if (std::max(std::fabs(slope_up), std::fabs(slope_down)) < gp.slope_ths) {
            const int x_max = std::max(std::max(poly[0], poly[2]), std::max(poly[4], poly[6]));
            const int x_min = std::min(std::min(poly[0], poly[2]), std::min(poly[4], poly[6]));
            const int y_max = std::max(std::max(poly[1], poly[3]), std::max(poly[5], poly[7]));
            const int y_min = std::min(std::min(poly[1], poly[3]), std::min(poly[5], poly[7]));
            horizontal_list.push_back({x_min, x_max, y_min, y_max, 0.5 * (double)(y_min + y_max), y_max - y_min});
        } else {
            const double dxh = poly[6] - poly[0], dyh = poly[7] - poly[1], dxw = poly[2] - poly[0], dyw = poly[3] - poly[1];
            const double height = std::sqrt(dxh * dxh + dyh * dyh);
            const double width = std::sqrt(dxw * dxw + dyw * dyw);
            const int margin = (int)(1.44 * gp.add_margin * std::min(width, height));
            const double theta13 = std::fabs(std::atan((double)(poly[1] - poly[5]) / (double)std::max(10, poly[0] - poly[4])));
            const double theta24 = std::fabs(std::atan((double)(poly[3] - poly[7]) / (double)std::max(10, poly[2] - poly[6])));
            std::array<double, 8> f;
            f[0] = poly[0] - std::cos(theta13) * margin; f[1] = poly[1] - std::sin(theta13) * margin;
            f[2] = poly[2] + std::cos(theta24) * margin; f[3] = poly[3] - std::sin(theta24) * margin;
            f[4] = poly[4] + std::cos(theta13) * margin; f[5] = poly[5] + std::sin(theta13) * margin;
            f[6] = poly[6] - std::cos(theta24) * margin; f[7] = poly[7] + std::sin(theta24) * margin;
            free_list.push_back(f);
        }
    }
    std::stable_sort(horizontal_list.begin(), horizontal_list.end(), [](const HB& a, const HB& b) { return a.yc < b.yc; });
    std::vector<std::vector<HB>> combined_list;
    std::vector<HB> new_box;
    std::vector<double> b_height, b_ycenter;
    for (const HB& poly : horizontal_list) {
        if (new_box.empty()) {
            b_height.assign(1, poly.hgt);
            b_ycenter.assign(1, poly.yc);
            new_box.push_back(poly);
        } else if (std::fabs(mean_of(b_ycenter) - poly.yc) < gp.ycenter_ths * mean_of(b_height)) {
            b_height.push_back(poly.hgt);
            b_ycenter.push_back(poly.yc);
            new_box.push_back(poly);
        } else {
            b_height.assign(1, poly.hgt);
            b_ycenter.assign(1, poly.yc);
            combined_list.push_back(new_box);
            new_box.assign(1, poly);
        }
    }
    combined_list.push_back(new_box);
    for (auto& boxes : combined_list) {
        if (boxes.size() == 1) {
            const HB& box = boxes[0];
            const int margin = (int)(gp.add_margin * std::min(box.x_max - box.x_min, box.hgt));
            merged_list.push_back({box.x_min - margin, box.x_max + margin, box.y_min - margin, box.y_max + margin});
        } else {
            std::stable_sort(boxes.begin(), boxes.end(), [](const HB& a, const HB& b) { return a.x_min < b.x_min; });
            std::vector<std::vector<HB>> merged_box;
            std::vector<HB> nb;
            std::vector<double> bh;
            int x_max = 0;
            for (const HB& box : boxes) {
                if (nb.empty()) {
                    bh.assign(1, box.hgt);
                    x_max = box.x_max;
                    nb.push_back(box);
                } else if ((std::fabs(mean_of(bh) - box.hgt) < gp.height_ths * mean_of(bh)) &&
                           ((double)(box.x_min - x_max) < gp.width_ths * (double)(box.y_max - box.y_min))) {
                    bh.push_back(box.hgt);
                    x_max = box.x_max;
                    nb.push_back(box);
                } else {
                    bh.assign(1, box.hgt);
                    x_max = box.x_max;
                    merged_box.push_back(nb);
                    nb.assign(1, box);
                }
            }
            if (!nb.empty()) merged_box.push_back(nb);
            for (const auto& mbox : merged_box) {
                if (mbox.size() != 1) {
                    int xmn = mbox[0].x_min, xmx = mbox[0].x_max, ymn = mbox[0].y_min, ymx = mbox[0].y_max;
                    for (const HB& b : mbox) {
                        xmn = std::min(xmn, b.x_min); xmx = std::max(xmx, b.x_max);
                        ymn = std::min(ymn, b.y_min); ymx = std::max(ymx, b.y_max);
                    }
                    const int margin = (int)(gp.add_margin * std::min(xmx - xmn, ymx - ymn));
                    merged_list.push_back({xmn - margin, xmx + margin, ymn - margin, ymx + margin});
                } else {
                    const HB& box = mbox[0];
                    const int margin = (int)(gp.add_margin * std::min(box.x_max - box.x_min, box.y_max - box.y_min));
                    merged_list.push_back({box.x_min - margin, box.x_max + margin, box.y_min - margin, box.y_max + margin});
                }
            }
        }
    }
    if (gp.min_size) {
        std::vector<std::array<int, 4>> h2;
        for (const auto& b : merged_list)
            if (std::max(b[1] - b[0], b[3] - b[2]) > gp.min_size) h2.push_back(b);
        merged_list.swap(h2);
        std::vector<std::array<double, 8>> f2;
        for (const auto& f : free_list) {
            const double dx = std::max(std::max(f[0], f[2]), std::max(f[4], f[6])) - std::min(std::min(f[0], f[2]), std::min(f[4], f[6]));
            const double dy = std::max(std::max(f[1], f[3]), std::max(f[5], f[7])) - std::min(std::min(f[1], f[3]), std::min(f[5], f[7]));
            if (std::max(dx, dy) > gp.min_size) f2.push_back(f);
        }
        free_list.swap(f2);
    }
}

// ------------------------------------------------------------------------------------------------ perspective helpers
bool solve8(double a[8][8], double b[8], double x[8]) {
    double A[8][9];
    for (int i = 0; i < 8; ++i) { for (int j = 0; j < 8; ++j) A[i][j] = a[i][j]; A[i][8] = b[i]; }
    for (int i = 0; i < 8; ++i) {
        int k = i;
        for (int j = i + 1; j < 8; ++j)
            if (std::fabs(A[j][i]) > std::fabs(A[k][i])) k = j;
        if (std::fabs(A[k][i]) < 2.220446049250313e-16 * 100) return false;
        if (k != i)
            for (int c = 0; c < 9; ++c) std::swap(A[i][c], A[k][c]);
        const double d = -1.0 / A[i][i];
        for (int j = i + 1; j < 8; ++j) {
            const double alpha = A[j][i] * d;
            for (int c = i + 1; c < 9; ++c) A[j][c] += alpha * A[i][c];
        }
    }
    for (int i = 7; i >= 0; --i) {
        double s = A[i][8];
        for (int c = i + 1; c < 8; ++c) s -= A[i][c] * x[c];
        x[i] = s / A[i][i];
    }
    return true;
}

void perspective_inverse(const float src[4][2], int max_w, int max_h, double Minv[9]) {
    const float dst[4][2] = {{0.f, 0.f}, {(float)(max_w - 1), 0.f}, {(float)(max_w - 1), (float)(max_h - 1)}, {0.f, (float)(max_h - 1)}};
    double a[8][8] = {}, b[8], x[8] = {};
    for (int i = 0; i < 4; ++i) {
        const double sx = src[i][0], sy = src[i][1], dx = dst[i][0], dy = dst[i][1];
        a[i][0] = a[i + 4][3] = sx;
        a[i][1] = a[i + 4][4] = sy;
        a[i][2] = a[i + 4][5] = 1.0;
        a[i][6] = -sx * dx; a[i][7] = -sy * dx;
        a[i + 4][6] = -sx * dy; a[i + 4][7] = -sy * dy;
        b[i] = dx; b[i + 4] = dy;
    }
    if (!solve8(a, b, x))
        for (int i = 0; i < 8; ++i) x[i] = 0.0;
    const double m[3][3] = {{x[0], x[1], x[2]}, {x[3], x[4], x[5]}, {x[6], x[7], 1.0}};
    double d = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
               m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    for (int i = 0; i < 9; ++i) Minv[i] = 0.0;
    if (d != 0.0) {
        d = 1.0 / d;
        Minv[0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) * d;
        Minv[1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * d;
        Minv[2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * d;
        Minv[3] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * d;
        Minv[4] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * d;
        Minv[5] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * d;
        Minv[6] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * d;
        Minv[7] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * d;
        Minv[8] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * d;
    }
}

}  // namespace bbocr
