// Native Flex-DSL placer (see flex_place.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace mic {

enum { kFlexOk = 0, kFlexUnsupported = 1, kFlexMalformed = 2 };

// layout JSON text ({"root": {...}}) + cutout sizes + canvas -> object ids and clamped boxes
// (x1, y1, x2, y2) in depth-first order.  kFlexUnsupported: the tree uses something only the
// Python placer mirrors (the caller falls back to it); kFlexMalformed: not JSON.
int flex_place(const char *json, size_t len, int n_obj, const int32_t *obj_ids, const int32_t *obj_w,
               const int32_t *obj_h, int W, int H, std::vector<int32_t> *out_ids, std::vector<int32_t> *out_boxes,
               std::string *err);

}  // namespace mic
