// Host-side box geometry (see boxpost.cpp).  Internal.
#pragma once
#include <array>
#include <vector>

namespace bbocr {

struct Component {   // one accepted connected component (heat-map coordinates)
    int root, left, top, right, bottom, area, row_off;
};

struct GroupParams {
    double slope_ths, ycenter_ths, height_ths, width_ths, add_margin;
    int min_size;
};

// rowext: [bottom-top+1][2] min/max x of TEXT pixels per component row (max < 0: none)
void component_box(const Component& c, const int* rowext, int img_w, int img_h, float box[4][2]);
void box_to_poly(const float box[4][2], double ratio_w, double ratio_h, int poly[8]);
void group_text_box(const std::vector<std::array<int, 8>>& polys, const GroupParams& gp, std::vector<std::array<int, 4>>& merged_list,
                    std::vector<std::array<double, 8>>& free_list);
void perspective_inverse(const float src[4][2], int max_w, int max_h, double Minv[9]);

}  // namespace bbocr
