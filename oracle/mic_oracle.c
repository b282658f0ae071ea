/*
 * oracle/mic_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the arithmetic the reference's pixel hot path runs.
 * It is the parity checker for the HIP kernels and the "port" CPU baseline of
 * bench.py; the product package never imports, links or calls it.
 *
 * The reference (/root/reference, Python) delegates the arithmetic to Pillow
 * (requirements.txt:29 pins pillow==11.3.0; the oracle was pinned here against
 * the installed Pillow 12.2.0) and NumPy's median.  Each function below cites the
 * reference call site it follows and the Pillow routine whose published
 * algorithm it restates:
 *
 *   compositor.py:6-22      composite()            -> orc_composite
 *   compositor.py:20        Image.resize(LANCZOS)  -> orc_resize       (Pillow Image.resize,
 *                                                     Resample.c precompute_coeffs /
 *                                                     normalize_coeffs_8bpc / Horizontal_8bpc /
 *                                                     Vertical_8bpc, Convert.c rgbA2rgba / rgba2rgbA)
 *   compositor.py:21        Image.alpha_composite  -> orc_alpha_over_at (AlphaComposite.c, Crop/Paste clip)
 *   background_resizing.py:11-22  median colour    -> orc_median_rgb   (np.median + int())
 *   background_resizing.py:25-33  fill_solid       -> orc_fill_solid
 *   macro_placement_test.py:194   Image.thumbnail  -> orc_thumbnail_size (Pillow Image.thumbnail size rule)
 *
 * Pinning: tests/test_oracle_golden.py checks every function against fixtures
 * produced by importing the reference in the build container
 * (tests/golden/make_golden.py), and tests/test_oracle_vs_pillow.py checks it
 * against live Pillow when Pillow is importable.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_FILTER_LANCZOS 0
#define ORC_FILTER_BILINEAR 1

static inline uint32_t div255_shift(uint32_t t) { return ((t >> 8) + t) >> 8; }

/* Pillow AlphaComposite.c, one pixel: src over dst.  (compositor.py:21) */
void orc_alpha_over_px(const uint8_t *dst, const uint8_t *src, uint8_t *out) {
    if (src[3] == 0) {
        memcpy(out, dst, 4);
        return;
    }
    uint32_t sa = src[3], da = dst[3];
    uint32_t blend = da * (255 - sa);
    uint32_t outa255 = sa * 255 + blend;
    uint32_t coef1 = sa * 255 * 255 * 128 / outa255; /* 7 precision bits */
    uint32_t coef2 = 255 * 128 - coef1;
    for (int c = 0; c < 3; c++) {
        uint32_t t = src[c] * coef1 + dst[c] * coef2 + (0x80u << 7);
        out[c] = (uint8_t)(div255_shift(t) >> 7);
    }
    out[3] = (uint8_t)div255_shift(outa255 + 0x80);
}

/*
 * In-place Image.alpha_composite(overlay, dest=(dx,dy)) as the reference calls it
 * (compositor.py:21, macro_placement_test.py:218): Pillow crops the destination
 * box (transparent padding outside the canvas), blends, and pastes back clipped,
 * so only the in-canvas part of the overlay has any effect.
 */
void orc_alpha_over_at(uint8_t *canvas, int W, int H, const uint8_t *ov, int ow, int oh,
                       int dx, int dy) {
    for (int y = 0; y < oh; y++) {
        int cy = dy + y;
        if (cy < 0 || cy >= H) continue;
        for (int x = 0; x < ow; x++) {
            int cx = dx + x;
            if (cx < 0 || cx >= W) continue;
            uint8_t *d = canvas + ((size_t)cy * W + cx) * 4;
            orc_alpha_over_px(d, ov + ((size_t)y * ow + x) * 4, d);
        }
    }
}

/* Pillow Convert.c rgbA2rgba: RGBA -> premultiplied RGBa. */
void orc_premultiply(const uint8_t *in, uint8_t *out, size_t npx) {
    for (size_t i = 0; i < npx; i++, in += 4, out += 4) {
        uint32_t a = in[3];
        for (int c = 0; c < 3; c++) out[c] = (uint8_t)div255_shift(in[c] * a + 128);
        out[3] = (uint8_t)a;
    }
}

/* Pillow Convert.c rgba2rgbA: premultiplied RGBa -> RGBA. */
void orc_unpremultiply(const uint8_t *in, uint8_t *out, size_t npx) {
    for (size_t i = 0; i < npx; i++, in += 4, out += 4) {
        uint32_t a = in[3];
        if (a == 255 || a == 0) {
            out[0] = in[0]; out[1] = in[1]; out[2] = in[2];
        } else {
            for (int c = 0; c < 3; c++) {
                uint32_t v = (255u * in[c]) / a;
                out[c] = (uint8_t)(v > 255 ? 255 : v);
            }
        }
        out[3] = (uint8_t)a;
    }
}

static double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
static double lanczos_filter(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}
static double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}

/*
 * Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for one axis.
 * Returns ksize; *bounds_out = [xmin, n] pairs, *kk_out = out_size*ksize int32
 * fixed-point (22 bit) coefficients.  Caller frees both.
 */
int orc_resample_coeffs(int in_size, int out_size, int filter, int32_t **bounds_out,
                        int32_t **kk_out) {
    double (*f)(double) = filter == ORC_FILTER_BILINEAR ? bilinear_filter : lanczos_filter;
    double fsupport = filter == ORC_FILTER_BILINEAR ? 1.0 : 3.0;
    double scale = (double)in_size / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    double support = fsupport * filterscale;
    int ksize = (int)ceil(support) * 2 + 1;
    int32_t *bounds = (int32_t *)malloc(sizeof(int32_t) * 2 * out_size);
    int32_t *kk = (int32_t *)calloc((size_t)out_size * ksize, sizeof(int32_t));
    double *k = (double *)malloc(sizeof(double) * ksize);
    double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; xx++) {
        double center = (xx + 0.5) * scale, ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; x++) {
            double w = f((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; x++) {
            if (ww != 0.0) k[x] /= ww;
            kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int32_t)(-0.5 + k[x] * (1 << 22))
                                                  : (int32_t)(0.5 + k[x] * (1 << 22));
        }
        bounds[xx * 2] = xmin;
        bounds[xx * 2 + 1] = xmax;
    }
    free(k);
    *bounds_out = bounds;
    *kk_out = kk;
    return ksize;
}

static inline uint8_t clip8(int32_t v) {
    v >>= 22; /* arithmetic shift, as Pillow's clip8 lookup index */
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/*
 * Image.resize((dw,dh), LANCZOS) on an RGBA image exactly as Pillow runs it for
 * the reference (compositor.py:20): identity -> copy; otherwise premultiply,
 * horizontal pass, 8-bit intermediate, vertical pass, unpremultiply.
 */
int orc_resize(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh, int filter) {
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return -1;
    if (sw == dw && sh == dh) {
        memcpy(dst, src, (size_t)sw * sh * 4);
        return 0;
    }
    uint8_t *cur = (uint8_t *)malloc((size_t)sw * sh * 4);
    orc_premultiply(src, cur, (size_t)sw * sh);
    int cw = sw, chh = sh;
    if (dw != sw) { /* horizontal pass over every source row */
        int32_t *bounds, *kk;
        int ksize = orc_resample_coeffs(sw, dw, filter, &bounds, &kk);
        uint8_t *nxt = (uint8_t *)malloc((size_t)dw * chh * 4);
        for (int y = 0; y < chh; y++) {
            const uint8_t *row = cur + (size_t)y * cw * 4;
            for (int xx = 0; xx < dw; xx++) {
                int xmin = bounds[xx * 2], n = bounds[xx * 2 + 1];
                const int32_t *k = kk + (size_t)xx * ksize;
                int32_t s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21, s3 = 1 << 21;
                for (int x = 0; x < n; x++) {
                    const uint8_t *p = row + (size_t)(xmin + x) * 4;
                    s0 += p[0] * k[x]; s1 += p[1] * k[x]; s2 += p[2] * k[x]; s3 += p[3] * k[x];
                }
                uint8_t *o = nxt + ((size_t)y * dw + xx) * 4;
                o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2); o[3] = clip8(s3);
            }
        }
        free(bounds); free(kk); free(cur);
        cur = nxt; cw = dw;
    }
    if (dh != sh) { /* vertical pass */
        int32_t *bounds, *kk;
        int ksize = orc_resample_coeffs(sh, dh, filter, &bounds, &kk);
        uint8_t *nxt = (uint8_t *)malloc((size_t)cw * dh * 4);
        for (int yy = 0; yy < dh; yy++) {
            int ymin = bounds[yy * 2], n = bounds[yy * 2 + 1];
            const int32_t *k = kk + (size_t)yy * ksize;
            for (int x = 0; x < cw; x++) {
                int32_t s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21, s3 = 1 << 21;
                for (int y = 0; y < n; y++) {
                    const uint8_t *p = cur + ((size_t)(ymin + y) * cw + x) * 4;
                    s0 += p[0] * k[y]; s1 += p[1] * k[y]; s2 += p[2] * k[y]; s3 += p[3] * k[y];
                }
                uint8_t *o = nxt + ((size_t)yy * cw + x) * 4;
                o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2); o[3] = clip8(s3);
            }
        }
        free(bounds); free(kk); free(cur);
        cur = nxt; chh = dh;
    }
    orc_unpremultiply(cur, dst, (size_t)dw * dh);
    free(cur);
    return 0;
}

/*
 * compositor.py:6-22.  Placements arrive already coerced to ints by the caller
 * (int() truncation and the unknown-id skip are host-side Python semantics);
 * place_obj[i] indexes the object arrays, or is < 0 for "unknown id: skip".
 * out starts as a copy of bg (compositor.py:11) and never aliases it.
 */
int orc_composite(const uint8_t *bg, int W, int H, int n_obj, const int32_t *obj_w,
                  const int32_t *obj_h, const uint8_t *const *obj_rgba, int n_place,
                  const int32_t *place_obj, const int32_t *boxes_xyxy, int filter, uint8_t *out) {
    memcpy(out, bg, (size_t)W * H * 4);
    for (int i = 0; i < n_place; i++) {
        int o = place_obj[i];
        if (o < 0 || o >= n_obj) continue;
        const int32_t *b = boxes_xyxy + 4 * i;
        int w = b[2] - b[0], h = b[3] - b[1];
        if (w < 1) w = 1;
        if (h < 1) h = 1;
        if (w == obj_w[o] && h == obj_h[o]) {
            orc_alpha_over_at(out, W, H, obj_rgba[o], w, h, b[0], b[1]);
        } else {
            uint8_t *tmp = (uint8_t *)malloc((size_t)w * h * 4);
            if (!tmp) return -2;
            orc_resize(obj_rgba[o], obj_w[o], obj_h[o], tmp, w, h, filter);
            orc_alpha_over_at(out, W, H, tmp, w, h, b[0], b[1]);
            free(tmp);
        }
    }
    return 0;
}

/*
 * background_resizing.py:11-22: per-channel np.median over pixels with alpha > 0
 * (all pixels when none qualifies), truncated by int().  np.median of n values is
 * the mean of order statistics (n-1)//2 and n//2.
 */
void orc_median_rgb(const uint8_t *rgba, size_t npx, uint8_t out_rgb[3]) {
    uint64_t hist[3][256];
    for (int pass = 0; pass < 2; pass++) {
        memset(hist, 0, sizeof hist);
        uint64_t n = 0;
        for (size_t i = 0; i < npx; i++) {
            const uint8_t *p = rgba + i * 4;
            if (pass == 0 && p[3] == 0) continue;
            hist[0][p[0]]++; hist[1][p[1]]++; hist[2][p[2]]++;
            n++;
        }
        if (n == 0 && pass == 0) continue; /* fully transparent: fall back to all pixels */
        if (n == 0) { out_rgb[0] = out_rgb[1] = out_rgb[2] = 0; return; }
        uint64_t klo = (n - 1) / 2, khi = n / 2;
        for (int c = 0; c < 3; c++) {
            uint64_t cum = 0;
            int lo = -1, hi = -1;
            for (int v = 0; v < 256; v++) {
                cum += hist[c][v];
                if (lo < 0 && cum > klo) lo = v;
                if (hi < 0 && cum > khi) { hi = v; break; }
            }
            out_rgb[c] = (uint8_t)((lo + hi) / 2); /* int((lo+hi)/2.0) for non-negative ints */
        }
        return;
    }
}

/* background_resizing.py:32: Image.new("RGBA", (W,H), color + (255,)). */
void orc_fill_solid(uint8_t *out, int W, int H, const uint8_t rgba[4]) {
    for (size_t i = 0, n = (size_t)W * H; i < n; i++) memcpy(out + i * 4, rgba, 4);
}

/*
 * Pillow Image.thumbnail size rule (call site macro_placement_test.py:194):
 * unchanged if the image already fits, else aspect-preserving fit with
 * round_aspect = max(min(floor(v), ceil(v), key), 1) (ties -> floor).
 */
void orc_thumbnail_size(int w, int h, int req_w, int req_h, int *out_w, int *out_h) {
    int x = req_w, y = req_h;
    if (x >= w && y >= h) { *out_w = w; *out_h = h; return; }
    double aspect = (double)w / (double)h;
    if ((double)x / (double)y >= aspect) {
        double v = y * aspect;
        long fl = (long)floor(v), ce = (long)ceil(v);
        double kf = fabs(aspect - (double)fl / y), kc = fabs(aspect - (double)ce / y);
        long r = (kc < kf) ? ce : fl;
        x = (int)(r < 1 ? 1 : r);
    } else {
        double v = x / aspect;
        long fl = (long)floor(v), ce = (long)ceil(v);
        double kf = fl == 0 ? 0.0 : fabs(aspect - (double)x / fl);
        double kc = ce == 0 ? 0.0 : fabs(aspect - (double)x / ce);
        long r = (kc < kf) ? ce : fl;
        y = (int)(r < 1 ? 1 : r);
    }
    *out_w = x;
    *out_h = y;
}

/* macro_placement_test.py:967-983 _save_overlay_debug: a transparent RGBA overlay on which
 * ImageDraw.rectangle([x1, y1, x2, y2], outline=colour, width=3) is called per placement, in order
 * (ImageDraw on an RGBA image stores the ink, it does not blend: later outlines overwrite).
 * Restates Pillow Draw.c ImagingDrawRectangle's outline branch as the installed Pillow behaves
 * (probed in the build container, pinned by tests/golden/overlay.npz): per i < width two clipped
 * horizontal lines at y0 + i and y1 - i over x0..x1, and two vertical lines at x1 - i and x0 + i that
 * start at y0 + width and run |dy| pixels TOWARDS y1 - width + 1 without reaching it -- for boxes
 * thinner than 2 width the "towards" is upwards and the outline spills outside the box.
 * Callers guarantee x1 >= x0 and y1 >= y0 (ImageDraw raises ValueError otherwise).
 */
static void orc_put(uint8_t *out, int W, int H, int x, int y, const uint8_t *ink) {
    if (x >= 0 && x < W && y >= 0 && y < H) memcpy(out + ((size_t)y * W + x) * 4, ink, 4);
}

void orc_rect_outlines(uint8_t *out, int W, int H, int n, const int32_t *boxes, const uint8_t *rgba, int width) {
    memset(out, 0, (size_t)W * H * 4);
    if (width <= 0) width = 1;
    for (int r = 0; r < n; ++r) {
        const int x0 = boxes[4 * r], y0 = boxes[4 * r + 1], x1 = boxes[4 * r + 2], y1 = boxes[4 * r + 3];
        const uint8_t *ink = rgba + 4 * r;
        for (int i = 0; i < width; ++i) {
            for (int x = x0; x <= x1; ++x) {
                orc_put(out, W, H, x, y0 + i, ink);
                orc_put(out, W, H, x, y1 - i, ink);
            }
            const int ya = y0 + width, yb = y1 - width + 1;
            const int dy = yb > ya ? yb - ya : ya - yb, ys = yb > ya ? 1 : -1;
            for (int k = 0, y = ya; k < dy; ++k, y += ys) {
                orc_put(out, W, H, x1 - i, y, ink);
                orc_put(out, W, H, x0 + i, y, ink);
            }
        }
    }
}
