// Native Flex-DSL placer: layout JSON text -> clamped placement boxes, for callers that place many
// variants per second (a 64-variant batch costs 11 ms of Python box maths against 0.5 ms of GPU time).
//
// Mirrors, for the well-formed subset of the DSL, exactly what image_transformation_amd/flex.py does
// (which mirrors macro_placement_test.py:637-964 of the reference and is pinned by fixtures):
//   measure  (:637-686)   object = cutout size + padding; container = sum/max of children + gaps + 2*padding
//   place    (:689-951)   justify start/center/end/space_between/space_around (floor division), align
//                         start/center/end on the cross axis, children placed with their measured size
//   clamp    (:954-964)   push boxes back inside the canvas, size preserved
// Anything whose behaviour depends on Python's type rules or on the object-level validators
// (pin / offset_px / stick_to, non-integer or string numbers, booleans used as integers, ids that
// are not plain integers, non-object children, ...) is NOT reimplemented here: the function reports
// kFlexUnsupported and the Python binding falls back to flex.py, which then produces the result or
// raises the reference's exact error.  So a result from here is always the reference's result.
#include "flex_place.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace mic {
namespace {

// ---------------------------------------------------------------------------------------- JSON
struct JValue;
using JPtr = std::unique_ptr<JValue>;
struct JValue {
    enum Kind { Null, Bool, Int, Float, String, Array, Object } kind = Null;
    bool b = false;
    long long i = 0;
    double d = 0.0;  // Float: the value Python's float() would hold (strtod is correctly rounded, like CPython's parser)
    std::string s;
    std::vector<JPtr> arr;
    std::vector<std::pair<std::string, JPtr>> obj;  // insertion order; duplicate keys: last wins on lookup
    const JValue *get(const char *key) const {
        const JValue *hit = nullptr;
        for (const auto &kv : obj)
            if (kv.first == key) hit = kv.second.get();
        return hit;
    }
};

struct Parser {
    const char *p, *end;
    bool ok = true;
    bool saw_unicode_escape = false;  // \uXXXX could spell a DSL keyword: leave those documents to Python
    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool lit(const char *t) {
        const size_t n = strlen(t);
        if ((size_t)(end - p) >= n && memcmp(p, t, n) == 0) {
            p += n;
            return true;
        }
        return false;
    }
    JPtr fail() {
        ok = false;
        return nullptr;
    }
    bool parse_string(std::string *out) {
        if (p >= end || *p != '"') return false;
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return false;
                switch (*p) {
                    case '"': out->push_back('"'); break;
                    case '\\': out->push_back('\\'); break;
                    case '/': out->push_back('/'); break;
                    case 'b': out->push_back('\b'); break;
                    case 'f': out->push_back('\f'); break;
                    case 'n': out->push_back('\n'); break;
                    case 'r': out->push_back('\r'); break;
                    case 't': out->push_back('\t'); break;
                    case 'u':
                        if (end - p < 5) return false;
                        saw_unicode_escape = true;
                        out->append("\\u");
                        out->append(p + 1, 4);
                        p += 4;
                        break;
                    default: return false;
                }
                ++p;
            } else {
                out->push_back(*p++);
            }
        }
        if (p >= end) return false;
        ++p;
        return true;
    }
    JPtr value(int depth) {
        if (depth > 64) return fail();
        ws();
        if (p >= end) return fail();
        JPtr v(new JValue());
        if (*p == '{') {
            v->kind = JValue::Object;
            ++p;
            ws();
            if (p < end && *p == '}') { ++p; return v; }
            for (;;) {
                ws();
                std::string key;
                if (!parse_string(&key)) return fail();
                ws();
                if (p >= end || *p != ':') return fail();
                ++p;
                JPtr child = value(depth + 1);
                if (!ok) return nullptr;
                v->obj.emplace_back(std::move(key), std::move(child));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return v; }
                return fail();
            }
        }
        if (*p == '[') {
            v->kind = JValue::Array;
            ++p;
            ws();
            if (p < end && *p == ']') { ++p; return v; }
            for (;;) {
                JPtr child = value(depth + 1);
                if (!ok) return nullptr;
                v->arr.push_back(std::move(child));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return v; }
                return fail();
            }
        }
        if (*p == '"') {
            v->kind = JValue::String;
            if (!parse_string(&v->s)) return fail();
            return v;
        }
        if (lit("true")) { v->kind = JValue::Bool; v->b = true; return v; }
        if (lit("false")) { v->kind = JValue::Bool; return v; }
        if (lit("null")) return v;
        // number
        const char *s0 = p;
        if (p < end && *p == '-') ++p;
        if (p >= end || *p < '0' || *p > '9') return fail();
        while (p < end && *p >= '0' && *p <= '9') ++p;
        bool is_float = false;
        if (p < end && *p == '.') { is_float = true; ++p; while (p < end && *p >= '0' && *p <= '9') ++p; }
        if (p < end && (*p == 'e' || *p == 'E')) {
            is_float = true;
            ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        if (is_float || p - s0 > 15) {
            v->kind = JValue::Float;
            v->d = strtod(std::string(s0, p).c_str(), nullptr);
            // (a 16+ digit integer lands here too: far outside every range accepted below)
        } else {
            v->kind = JValue::Int;
            v->i = strtoll(std::string(s0, p).c_str(), nullptr, 10);
        }
        return v;
    }
};

// ---------------------------------------------------------------------------------------- placer
struct Unsupported {};

struct Placer {
    const std::map<long long, std::pair<int, int>> &sizes;
    std::vector<int32_t> &ids, &boxes;

    // int("12") / int(" -7 "): an optional sign and plain ASCII decimal digits between ASCII blanks.  Everything
    // else int() accepts for a str (underscores, other Unicode digits and spaces) stays Python's business.
    static long long int_of_string(const std::string &t) {
        size_t a = 0, b = t.size();
        while (a < b && (t[a] == ' ' || t[a] == '\t' || t[a] == '\n')) ++a;
        while (b > a && (t[b - 1] == ' ' || t[b - 1] == '\t' || t[b - 1] == '\n')) --b;
        size_t d = a;
        if (d < b && (t[d] == '+' || t[d] == '-')) ++d;
        if (d == b || b - d > 9) throw Unsupported{};
        for (size_t k = d; k < b; ++k)
            if (t[k] < '0' || t[k] > '9') throw Unsupported{};
        return strtoll(t.substr(a, b - a).c_str(), nullptr, 10);
    }
    // int(node.get(key, dflt)) as the reference applies it to a container's gap_px / padding_px
    // (macro_placement_test.py:661-662, :696-697): an int as it is, a float truncated toward zero, a bool as 0 / 1,
    // a plain decimal string parsed; null / lists / objects raise TypeError there -> the Python placer's business.
    static long long int_field(const JValue &node, const char *key, long long dflt) {
        const JValue *v = node.get(key);
        if (!v) return dflt;
        long long r;
        switch (v->kind) {
            case JValue::Int: r = v->i; break;
            case JValue::Bool: r = v->b ? 1 : 0; break;
            case JValue::Float:
                if (!(v->d > -16777216.0 && v->d < 16777216.0)) throw Unsupported{};  // (also NaN)
                r = (long long)v->d;  // C++ truncates toward zero, like int(float)
                break;
            case JValue::String: r = int_of_string(v->s); break;
            default: throw Unsupported{};
        }
        if (r < -(1 << 24) || r > (1 << 24)) throw Unsupported{};
        return r;
    }
    static std::string str_field(const JValue &node, const char *key, const char *dflt) {
        const JValue *v = node.get(key);
        if (!v) return dflt;
        if (v->kind != JValue::String) throw Unsupported{};  // compared with == against strings in Python
        return v->s;
    }
    static const std::vector<JPtr> &children_of(const JValue &node) {
        static const std::vector<JPtr> empty;
        const JValue *v = node.get("children");
        if (!v) return empty;
        if (v->kind != JValue::Array) throw Unsupported{};
        return v->arr;
    }
    static bool is_object_node(const JValue &n) { return n.get("object_id") != nullptr; }

    static bool keys_within(const JValue &o, std::initializer_list<const char *> allowed) {
        for (size_t a = 0; a < o.obj.size(); ++a) {
            bool known = false;
            for (const char *k : allowed) known = known || o.obj[a].first == k;
            if (!known) return false;
            for (size_t b = a + 1; b < o.obj.size(); ++b)
                if (o.obj[a].first == o.obj[b].first) return false;  // duplicate keys: Python keeps the last
        }
        return true;
    }
    static void validate_hints(const JValue &node) {
        if (const JValue *pin = node.get("pin")) {
            if (pin->kind != JValue::Null) {
                if (pin->kind != JValue::Object || !keys_within(*pin, {"horizontal", "vertical"})) throw Unsupported{};
                for (const auto &kv : pin->obj) {
                    const JValue &v = *kv.second;
                    if (v.kind == JValue::Null) continue;
                    if (v.kind != JValue::String || (v.s != "start" && v.s != "center" && v.s != "end")) throw Unsupported{};
                }
            }
        }
        if (const JValue *off = node.get("offset_px")) {
            if (off->kind != JValue::Null) {
                if (off->kind != JValue::Object || !keys_within(*off, {"x", "y"})) throw Unsupported{};
                for (const auto &kv : off->obj)
                    if (kv.second->kind != JValue::Int) throw Unsupported{};
            }
        }
        if (const JValue *st = node.get("stick_to")) {
            if (st->kind != JValue::Null) {
                if (st->kind != JValue::Object || !keys_within(*st, {"edges", "margin_px"})) throw Unsupported{};
                const JValue *edges = st->get("edges");
                if (!edges || edges->kind != JValue::Array || edges->arr.empty()) throw Unsupported{};
                bool seen[4] = {false, false, false, false};  // left, right, top, bottom
                for (const JPtr &e : edges->arr) {
                    if (e->kind != JValue::String) throw Unsupported{};
                    std::string low = e->s;
                    for (char &ch : low) {
                        if ((unsigned char)ch >= 0x80) throw Unsupported{};  // str.lower() on non-ASCII: Python's business
                        if (ch >= 'A' && ch <= 'Z') ch = (char)(ch - 'A' + 'a');
                    }
                    const int idx = low == "left" ? 0 : low == "right" ? 1 : low == "top" ? 2 : low == "bottom" ? 3 : -1;
                    if (idx < 0 || seen[idx]) throw Unsupported{};
                    seen[idx] = true;
                }
                if ((seen[0] && seen[1]) || (seen[2] && seen[3])) throw Unsupported{};
                if (const JValue *m = st->get("margin_px"))
                    if (m->kind != JValue::Int || m->i < 0) throw Unsupported{};
            }
        }
    }

    struct Pad { long long l = 0, r = 0, t = 0, b = 0; };
    static Pad object_padding(const JValue &node) {
        // pin / offset_px / stick_to never move a box (macro_placement_test.py:767-811 nets out to
        // slot + padding), but the reference validates them (:286-372) and raises on anything odd.
        // Plainly valid values are accepted here; everything else goes to the Python placer, which
        // raises the reference's errors.
        validate_hints(node);
        Pad pad;
        const JValue *v = node.get("padding_px");
        if (!v || v->kind == JValue::Null) return pad;
        if (v->kind == JValue::Int) {
            if (v->i < 0 || v->i > (1 << 24)) throw Unsupported{};
            pad.l = pad.r = pad.t = pad.b = v->i;
            return pad;
        }
        if (v->kind != JValue::Object) throw Unsupported{};
        for (const auto &kv : v->obj) {
            if (kv.first != "left" && kv.first != "right" && kv.first != "top" && kv.first != "bottom") throw Unsupported{};
            if (kv.second->kind != JValue::Int || kv.second->i < 0 || kv.second->i > (1 << 24)) throw Unsupported{};
        }
        auto side = [&](const char *k) { const JValue *s = v->get(k); return s ? s->i : 0LL; };
        pad.l = side("left"); pad.r = side("right"); pad.t = side("top"); pad.b = side("bottom");
        return pad;
    }
    long long object_id(const JValue &node) const {
        const JValue *v = node.get("object_id");
        if (v->kind == JValue::String) return int_of_string(v->s);  // int("3")
        if (v->kind != JValue::Int) throw Unsupported{};  // int(3.7), int(True), None: Python's business
        return v->i;
    }
    std::pair<long long, long long> measure(const JValue &node) const {
        if (node.kind != JValue::Object) throw Unsupported{};
        if (is_object_node(node)) {
            const Pad pad = object_padding(node);
            auto it = sizes.find(object_id(node));
            const long long w = it == sizes.end() ? 0 : it->second.first, h = it == sizes.end() ? 0 : it->second.second;
            return {std::max(0LL, w + pad.l + pad.r), std::max(0LL, h + pad.t + pad.b)};
        }
        const long long gap = int_field(node, "gap_px", 0), pad = int_field(node, "padding_px", 0);
        const auto &kids = children_of(node);
        if (kids.empty()) return {std::max(0LL, 2 * pad), std::max(0LL, 2 * pad)};
        const bool row = str_field(node, "direction", "row") == "row";
        long long sum = 0, mx = 0;
        for (const auto &k : kids) {
            const auto s = measure(*k);
            sum += row ? s.first : s.second;
            mx = std::max(mx, row ? s.second : s.first);
        }
        sum += kids.size() > 1 ? gap * (long long)(kids.size() - 1) : 0;
        const long long grow = 2 * std::max(0LL, pad);
        return row ? std::make_pair(std::max(0LL, sum + grow), std::max(0LL, mx + grow))
                   : std::make_pair(std::max(0LL, mx + grow), std::max(0LL, sum + grow));
    }
    static long long floordiv(long long a, long long b) {  // Python's //, b > 0
        long long q = a / b;
        if ((a % b != 0) && (a < 0)) --q;
        return q;
    }
    void place(const JValue &node, long long x0, long long y0, long long cw, long long ch) {
        const bool row = str_field(node, "direction", "row") == "row";
        const std::string justify = str_field(node, "justify", "center"), align = str_field(node, "align", "center");
        const long long gap = int_field(node, "gap_px", 0), pad = int_field(node, "padding_px", 0);
        const long long ix = x0 + pad, iy = y0 + pad;
        const long long iw = std::max(0LL, cw - 2 * pad), ih = std::max(0LL, ch - 2 * pad);
        const auto &kids = children_of(node);
        const long long n = (long long)kids.size();
        std::vector<std::pair<long long, long long>> sz;
        sz.reserve(kids.size());
        long long sum_main = 0;
        for (const auto &k : kids) {
            sz.push_back(measure(*k));
            sum_main += row ? sz.back().first : sz.back().second;
        }
        const long long inner_main = row ? iw : ih, inner_cross = row ? ih : iw;
        const long long total = sum_main + gap * (n > 0 ? n - 1 : 0);
        long long start = 0, step = gap;
        if (justify == "start") {
        } else if (justify == "center") {
            start = std::max(0LL, floordiv(inner_main - total, 2));
        } else if (justify == "end") {
            start = std::max(0LL, inner_main - total);
        } else if (justify == "space_between" && n > 1) {
            step = std::max(0LL, floordiv(inner_main - sum_main, n - 1));
        } else if (justify == "space_around" && n > 0) {
            step = std::max(0LL, floordiv(inner_main - sum_main, n));
            start = floordiv(step, 2);
        }
        long long cur = (row ? ix : iy) + start;
        for (size_t c = 0; c < kids.size(); ++c) {
            const long long sm = row ? sz[c].first : sz[c].second, sc = row ? sz[c].second : sz[c].first;
            const long long org = row ? iy : ix;
            long long cross;
            if (align == "start") cross = org;
            else if (align == "end") cross = org + (inner_cross - sc);
            else cross = org + floordiv(inner_cross - sc, 2);
            const long long px = row ? cur : cross, py = row ? cross : cur;
            const JValue &k = *kids[c];
            if (is_object_node(k)) {
                // slot == measured size, so the object sits at the slot's padded corner at scale 1
                // (pins / offsets / sticks are arithmetic no-ops there and are routed to Python anyway)
                const Pad op = object_padding(k);
                const long long oid = object_id(k);
                auto it = sizes.find(oid);
                const long long w = it == sizes.end() ? 0 : it->second.first, h = it == sizes.end() ? 0 : it->second.second;
                if (oid < INT32_MIN || oid > INT32_MAX) throw Unsupported{};
                ids.push_back((int32_t)oid);
                const long long bx = px + op.l, by = py + op.t;
                for (long long v : {bx, by, bx + w, by + h}) {
                    if (v < INT32_MIN / 2 || v > INT32_MAX / 2) throw Unsupported{};
                    boxes.push_back((int32_t)v);
                }
            } else {
                place(k, px, py, sz[c].first, sz[c].second);
            }
            cur += sm + step;
        }
    }
};

}  // namespace

int flex_place(const char *json, size_t len, int n_obj, const int32_t *obj_ids, const int32_t *obj_w,
               const int32_t *obj_h, int W, int H, std::vector<int32_t> *out_ids, std::vector<int32_t> *out_boxes,
               std::string *err) {
    Parser ps{json, json + len};
    JPtr doc = ps.value(0);
    if (ps.ok) {
        ps.ws();
        if (ps.p != ps.end) ps.ok = false;
    }
    if (!ps.ok || !doc) {
        *err = "malformed layout JSON";
        return kFlexMalformed;
    }
    if (doc->kind != JValue::Object || ps.saw_unicode_escape) return kFlexUnsupported;
    const JValue *root = doc->get("root");
    if (!root || root->kind != JValue::Object || root->get("object_id")) return kFlexUnsupported;
    std::map<long long, std::pair<int, int>> sizes;
    for (int i = 0; i < n_obj; ++i) sizes.emplace(obj_ids[i], std::make_pair(obj_w[i], obj_h[i]));
    out_ids->clear();
    out_boxes->clear();
    try {
        Placer pl{sizes, *out_ids, *out_boxes};
        pl.place(*root, 0, 0, W, H);
    } catch (const Unsupported &) {
        return kFlexUnsupported;
    }
    // _clamp_boxes_to_canvas (macro_placement_test.py:954-964)
    for (size_t i = 0; i < out_ids->size(); ++i) {
        int32_t *b = &(*out_boxes)[4 * i];
        const long long w = (long long)b[2] - b[0], h = (long long)b[3] - b[1];
        const long long x1 = std::max(0LL, std::min<long long>(b[0], W - w));
        const long long y1 = std::max(0LL, std::min<long long>(b[1], H - h));
        b[0] = (int32_t)x1; b[1] = (int32_t)y1; b[2] = (int32_t)(x1 + w); b[3] = (int32_t)(y1 + h);
    }
    return kFlexOk;
}

}  // namespace mic
