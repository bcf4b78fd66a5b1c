/*
 * stereonet_oracle.c — CPU-float ORACLE (see stereonet_oracle.h for status).
 * TEST INFRASTRUCTURE ONLY — never linked into the product library.
 *
 * Host-side integer functions restate the reference line ranges cited at each
 * function.  The network follows DESIGN.md §2 ("SN-K4"); it has no counterpart
 * source in the reference (the BPU blob is opaque), so each stage cites the
 * reference line that pins its *contract* instead.
 */
#include "stereonet_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int so_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ======================================================================== */
/* Reference host code, restated                                            */
/* ======================================================================== */

/* stereonet_node.cpp:705-738: the message is one 2w-wide NV12 image; each eye
 * gets h luma rows followed by h/2 chroma rows, left = first w bytes of every
 * source row, right = last w bytes. */
void so_split_sbs_nv12(const uint8_t *sbs, int w, int h, uint8_t *left, uint8_t *right) {
  const int rows = h + h / 2;
  for (int r = 0; r < rows; ++r) {
    const uint8_t *src = sbs + (size_t)r * 2 * w;
    memcpy(left + (size_t)r * w, src, (size_t)w);
    memcpy(right + (size_t)r * w, src + w, (size_t)w);
  }
}

/* preprocess.h:128-155.  srcU = in + w*h, srcV = srcU + w*h/4 (planar I420
 * interpretation of what is really interleaved NV12 chroma — reproduced on
 * purpose); the chroma sample for (i,j) is
 * src[(i/2)*w/2 + j/2], evaluated left to right in integers as the reference
 * does: ((i/2)*w)/2 + j/2. */
void so_yuv420_to_yuv444(const uint8_t *in, uint8_t *out, int w, int h) {
  const uint8_t *src_u = in + (size_t)w * h;
  const uint8_t *src_v = src_u + (size_t)w * h / 4;
  uint8_t *dst_u = out + (size_t)w * h;
  uint8_t *dst_v = dst_u + (size_t)w * h;
  memcpy(out, in, (size_t)w * h);
  for (int i = 0; i < h; ++i) {
    for (int j = 0; j < w; ++j) {
      const int s = (i / 2) * w / 2 + j / 2;
      dst_u[(size_t)i * w + j] = src_u[s];
      dst_v[(size_t)i * w + j] = src_v[s];
    }
  }
}

/* preprocess.cpp:1131-1136 */
int8_t so_quantize(float value, float scale, float zero_point, float mn, float mx) {
  value = floorf(value / scale + zero_point);
  value = fminf(fmaxf(value, mn), mx);
  return (int8_t)value;
}

/* preprocess.cpp:975-1056: YUV444 planes of the left eye, then of the right eye
 * (:999-1003), every byte through Quantize(((float)b - 128)/128) (:1033-1040). */
void so_preprocess_nv12(const uint8_t *img_l, const uint8_t *img_r, int w, int h, int8_t *out6) {
  const size_t plane3 = (size_t)3 * w * h;
  uint8_t *tmp = (uint8_t *)malloc(2 * plane3);
  so_yuv420_to_yuv444(img_l, tmp, w, h);
  so_yuv420_to_yuv444(img_r, tmp + plane3, w, h);
  for (size_t i = 0; i < 2 * plane3; ++i) {
    out6[i] = so_quantize(((float)tmp[i] - 128.0f) / 128.0f, 0.0078125f, 0.5f, -128.0f, 127.0f);
  }
  free(tmp);
}

/* preprocess.h:56-96: step 1 = planar I420 the way OpenCV's BGR2YUV_I420 fills it, step 2 = the reference's
 * own interleave loop (:91-94). */
static uint8_t so_sat8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

int so_bgr_to_nv12(const uint8_t *bgr, int w, int h, uint8_t *nv12) {
  if ((w % 2) || (h % 2)) return -1;
  const int SH = 20;
  const int RND = 1 << (SH - 1);
  const size_t ysz = (size_t)w * h, csz = ysz / 4;
  uint8_t *i420 = (uint8_t *)malloc(ysz + 2 * csz);
  uint8_t *yp = i420, *up = i420 + ysz, *vp = up + csz;
  for (int by = 0; by < h / 2; ++by) {
    for (int bx = 0; bx < w / 2; ++bx) {
      for (int dy = 0; dy < 2; ++dy) {
        for (int dx = 0; dx < 2; ++dx) {
          const uint8_t *px = bgr + ((size_t)(2 * by + dy) * w + (2 * bx + dx)) * 3;
          const int B = px[0], G = px[1], R = px[2];
          yp[(size_t)(2 * by + dy) * w + 2 * bx + dx] =
              so_sat8((269484 * R + 528482 * G + 102760 * B + RND + (16 << SH)) >> SH);
          if (dy == 0 && dx == 0) {
            up[(size_t)by * (w / 2) + bx] = so_sat8((-155188 * R - 305135 * G + 460324 * B + RND + (128 << SH)) >> SH);
            vp[(size_t)by * (w / 2) + bx] = so_sat8((460324 * R - 385875 * G - 74448 * B + RND + (128 << SH)) >> SH);
          }
        }
      }
    }
  }
  memcpy(nv12, yp, ysz);
  uint8_t *o = nv12 + ysz;
  for (size_t i = 0; i < csz; ++i) {
    *o++ = up[i];
    *o++ = vp[i];
  }
  free(i420);
  return 0;
}

/* parser.cpp:84-86 (C++, int32 view) == render.py:65-81 (uint32 view; equal
 * because the output is non-negative).  f, B: parser.cpp:70-71. */
void so_dequant_depth(const int32_t *raw, int n, float scale, float dmax,
                      float *disp_px, float *depth_m) {
  const float f = 527.1931762695312f;
  const float B = 119.89382172f;
  for (int i = 0; i < n; ++i) {
    const float dis = (float)raw[i] * scale;
    if (disp_px) disp_px[i] = dis * dmax;
    if (depth_m) depth_m[i] = (float)(f * B / (dis * dmax) / 1000.0);
  }
}

/* ======================================================================== */
/* Weight blob layout                                                       */
/* ======================================================================== */

typedef struct {
  const float *w, *b;
} conv_t;

typedef struct {
  conv_t down[SO_NDOWN];
  conv_t fres[SO_NFRES][2];
  conv_t fout;
  conv_t agg[SO_NAGG];
  conv_t aout;
  conv_t rin;
  conv_t rres[SO_NRRES][2];
  conv_t rout;
} net_t;

static const int k_ref_dil[SO_NRRES] = {1, 2, 4, 8, 1, 1};

#define SO_MAX_LAYERS 96
typedef struct {
  char name[24];
  long w_off, w_n, b_off, b_n;
} layer_t;

static layer_t g_layers[SO_MAX_LAYERS];
static int g_nlayers = 0;
static long g_total = 0;
static long g_total_single = 0;   /* parameters of the single-scale network = offset of the first coarse tower */

static void add_layer(const char *name, long nw, long nb) {
  layer_t *l = &g_layers[g_nlayers++];
  snprintf(l->name, sizeof l->name, "%s", name);
  l->w_off = g_total;
  l->w_n = nw;
  g_total += nw;
  l->b_off = g_total;
  l->b_n = nb;
  g_total += nb;
}

static void build_table(void) {
  if (g_nlayers) return;
  const long C = SO_C;
  char nm[24];
  for (int i = 0; i < SO_NDOWN; ++i) {
    snprintf(nm, sizeof nm, "feat.down%d", i);
    add_layer(nm, C * (i == 0 ? 3 : C) * 25, C);
  }
  for (int i = 0; i < SO_NFRES; ++i)
    for (int j = 1; j <= 2; ++j) {
      snprintf(nm, sizeof nm, "feat.res%d.%d", i, j);
      add_layer(nm, C * C * 9, C);
    }
  add_layer("feat.out", C * C * 9, C);
  for (int i = 0; i < SO_NAGG; ++i) {
    snprintf(nm, sizeof nm, "agg.conv%d", i);
    add_layer(nm, C * C * 27, C);
  }
  add_layer("agg.out", C * 27, 1);
  /* refinement towers: level 0 ("ref", full resolution = the `single` tower), then — only present in a `multi`
   * blob — the towers of the coarser levels 1 .. SO_MULTI_LEVELS-1 ("ref1" = 1/2, "ref2" = 1/4, "ref3" = 1/8) */
  for (int lv = 0; lv < SO_MULTI_LEVELS; ++lv) {
    char pre[8];
    if (lv == 0) snprintf(pre, sizeof pre, "ref");
    else snprintf(pre, sizeof pre, "ref%d", lv);
    snprintf(nm, sizeof nm, "%s.in", pre);
    add_layer(nm, C * 4 * 9, C);
    for (int i = 0; i < SO_NRRES; ++i)
      for (int j = 1; j <= 2; ++j) {
        snprintf(nm, sizeof nm, "%s.res%d.%d", pre, i, j);
        add_layer(nm, C * C * 9, C);
      }
    snprintf(nm, sizeof nm, "%s.out", pre);
    add_layer(nm, C * 9, 1);
    if (lv == 0) g_total_single = g_total;
  }
}

long so_weight_count(void) {
  build_table();
  return g_total_single;
}

long so_weight_count_levels(int levels) {
  build_table();
  if (levels <= 1) return g_total_single;
  return levels == SO_MULTI_LEVELS ? g_total : -1;
}

/* name is "<layer>.w" or "<layer>.b", e.g. "feat.res3.2.w" */
long so_weight_offset(const char *name) {
  build_table();
  size_t n = strlen(name);
  if (n < 3 || name[n - 2] != '.') return -1;
  for (int i = 0; i < g_nlayers; ++i) {
    if (strlen(g_layers[i].name) == n - 2 && strncmp(g_layers[i].name, name, n - 2) == 0) {
      if (name[n - 1] == 'w') return g_layers[i].w_off;
      if (name[n - 1] == 'b') return g_layers[i].b_off;
    }
  }
  return -1;
}

static conv_t get_conv(const float *base, const char *layer) {
  build_table();
  conv_t c = {NULL, NULL};
  for (int i = 0; i < g_nlayers; ++i)
    if (strcmp(g_layers[i].name, layer) == 0) {
      c.w = base + g_layers[i].w_off;
      c.b = base + g_layers[i].b_off;
    }
  return c;
}

static void bind_net(const float *base, net_t *net) {
  char nm[24];
  for (int i = 0; i < SO_NDOWN; ++i) {
    snprintf(nm, sizeof nm, "feat.down%d", i);
    net->down[i] = get_conv(base, nm);
  }
  for (int i = 0; i < SO_NFRES; ++i)
    for (int j = 0; j < 2; ++j) {
      snprintf(nm, sizeof nm, "feat.res%d.%d", i, j + 1);
      net->fres[i][j] = get_conv(base, nm);
    }
  net->fout = get_conv(base, "feat.out");
  for (int i = 0; i < SO_NAGG; ++i) {
    snprintf(nm, sizeof nm, "agg.conv%d", i);
    net->agg[i] = get_conv(base, nm);
  }
  net->aout = get_conv(base, "agg.out");
  net->rin = get_conv(base, "ref.in");
  for (int i = 0; i < SO_NRRES; ++i)
    for (int j = 0; j < 2; ++j) {
      snprintf(nm, sizeof nm, "ref.res%d.%d", i, j + 1);
      net->rres[i][j] = get_conv(base, nm);
    }
  net->rout = get_conv(base, "ref.out");
}

/* the tower of refinement level `level` (0 = "ref") bound into the rin / rres / rout members */
static void bind_tower(const float *base, int level, net_t *net) {
  char pre[8], nm[24];
  if (level == 0) snprintf(pre, sizeof pre, "ref");
  else snprintf(pre, sizeof pre, "ref%d", level);
  snprintf(nm, sizeof nm, "%s.in", pre);
  net->rin = get_conv(base, nm);
  for (int i = 0; i < SO_NRRES; ++i)
    for (int j = 0; j < 2; ++j) {
      snprintf(nm, sizeof nm, "%s.res%d.%d", pre, i, j + 1);
      net->rres[i][j] = get_conv(base, nm);
    }
  snprintf(nm, sizeof nm, "%s.out", pre);
  net->rout = get_conv(base, nm);
}

/* ======================================================================== */
/* Primitive ops                                                            */
/* ======================================================================== */

/* out[co][y][x] = bias[co] + sum_{ci,ky,kx} wt[co][ci][ky][kx] *
 *                 in[ci][y*stride + ky*dil - pad][x*stride + kx*dil - pad]   (0 outside) */
void so_conv2d(const float *in, int cin, int h, int w,
               const float *wt, const float *bias, int cout,
               int k, int stride, int pad, int dil, float *out) {
  const int ho = (h + 2 * pad - dil * (k - 1) - 1) / stride + 1;
  const int wo = (w + 2 * pad - dil * (k - 1) - 1) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int co = 0; co < cout; ++co) {
    for (int y = 0; y < ho; ++y) {
      float *orow = out + ((size_t)co * ho + y) * wo;
      const float b = bias ? bias[co] : 0.0f;
      for (int x = 0; x < wo; ++x) orow[x] = b;
      for (int ci = 0; ci < cin; ++ci) {
        for (int ky = 0; ky < k; ++ky) {
          const int iy = y * stride + ky * dil - pad;
          if (iy < 0 || iy >= h) continue;
          const float *irow = in + ((size_t)ci * h + iy) * w;
          for (int kx = 0; kx < k; ++kx) {
            const float wv = wt[(((size_t)co * cin + ci) * k + ky) * k + kx];
            const int off = kx * dil - pad;
            /* x*stride + off in [0, w) */
            int x0 = off < 0 ? (-off + stride - 1) / stride : 0;
            int x1 = (w - 1 - off) / stride;  /* inclusive */
            if (w - 1 - off < 0) continue;
            if (x1 > wo - 1) x1 = wo - 1;
            if (stride == 1) {
              const float *ip = irow + off;
              for (int x = x0; x <= x1; ++x) orow[x] += wv * ip[x];
            } else {
              for (int x = x0; x <= x1; ++x) orow[x] += wv * irow[x * stride + off];
            }
          }
        }
      }
    }
  }
}

/* 3x3x3, stride 1, pad 1; in [ci][d][h][w], wt [co][ci][kd][kh][kw] */
void so_conv3d(const float *in, int cin, int d, int h, int w,
               const float *wt, const float *bias, int cout, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int co = 0; co < cout; ++co) {
    for (int z = 0; z < d; ++z) {
      for (int y = 0; y < h; ++y) {
        float *orow = out + (((size_t)co * d + z) * h + y) * w;
        const float b = bias ? bias[co] : 0.0f;
        for (int x = 0; x < w; ++x) orow[x] = b;
        for (int ci = 0; ci < cin; ++ci) {
          for (int kz = 0; kz < 3; ++kz) {
            const int iz = z + kz - 1;
            if (iz < 0 || iz >= d) continue;
            for (int ky = 0; ky < 3; ++ky) {
              const int iy = y + ky - 1;
              if (iy < 0 || iy >= h) continue;
              const float *irow = in + (((size_t)ci * d + iz) * h + iy) * w;
              for (int kx = 0; kx < 3; ++kx) {
                const float wv = wt[((((size_t)co * cin + ci) * 3 + kz) * 3 + ky) * 3 + kx];
                const int off = kx - 1;
                const int x0 = off < 0 ? 1 : 0;
                const int x1 = off > 0 ? w - 2 : w - 1;
                const float *ip = irow + off;
                for (int x = x0; x <= x1; ++x) orow[x] += wv * ip[x];
              }
            }
          }
        }
      }
    }
  }
}

void so_lrelu(float *x, long n, float slope) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) x[i] = x[i] > 0.0f ? x[i] : x[i] * slope;
}

/* cv[c][d][y][x] = fl[c][y][x] - fr[c][y][x-d], 0 where x-d < 0 */
void so_cost_volume(const float *fl, const float *fr, int c, int dl, int h, int w, float *cv) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int ch = 0; ch < c; ++ch)
    for (int d = 0; d < dl; ++d)
      for (int y = 0; y < h; ++y) {
        const float *l = fl + ((size_t)ch * h + y) * w;
        const float *r = fr + ((size_t)ch * h + y) * w;
        float *o = cv + (((size_t)ch * dl + d) * h + y) * w;
        for (int x = 0; x < w; ++x) o[x] = x >= d ? l[x] - r[x - d] : 0.0f;
      }
}

/* p = softmax_d(-cost); disp = sum_d d * p_d  (max-subtracted) */
void so_soft_argmin(const float *cost, int dl, int h, int w, float *disp) {
  const size_t plane = (size_t)h * w;
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)plane; ++i) {
    float m = -cost[i];
    for (int d = 1; d < dl; ++d) m = fmaxf(m, -cost[d * plane + i]);
    float s = 0.0f, acc = 0.0f;
    for (int d = 0; d < dl; ++d) {
      const float e = expf(-cost[d * plane + i] - m);
      s += e;
      acc += (float)d * e;
    }
    disp[i] = acc / s;
  }
}

/* bilinear, align_corners=False (half-pixel centres, source index clamped at 0,
 * right/bottom neighbour clamped to the edge) — the torch.nn.functional.interpolate
 * convention; output multiplied by `mul`. */
void so_upsample_bilinear(const float *in, int h, int w, int factor, float mul, float *out) {
  const int ho = h * factor, wo = w * factor;
  const float rs = 1.0f / (float)factor;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < ho; ++y) {
    float sy = ((float)y + 0.5f) * rs - 0.5f;
    if (sy < 0.0f) sy = 0.0f;
    const int y0 = (int)sy;
    const int y1 = y0 < h - 1 ? y0 + 1 : y0;
    const float ly = sy - (float)y0, hy = 1.0f - ly;
    for (int x = 0; x < wo; ++x) {
      float sx = ((float)x + 0.5f) * rs - 0.5f;
      if (sx < 0.0f) sx = 0.0f;
      const int x0 = (int)sx;
      const int x1 = x0 < w - 1 ? x0 + 1 : x0;
      const float lx = sx - (float)x0, hx = 1.0f - lx;
      const float v = hy * (hx * in[(size_t)y0 * w + x0] + lx * in[(size_t)y0 * w + x1]) +
                      ly * (hx * in[(size_t)y1 * w + x0] + lx * in[(size_t)y1 * w + x1]);
      out[(size_t)y * wo + x] = v * mul;
    }
  }
}

/* ======================================================================== */
/* Stages                                                                   */
/* ======================================================================== */

static float *falloc(size_t n) { return (float *)malloc(n * sizeof(float)); }

/* y = lrelu(x + conv2(lrelu(conv1(x)))) with both convs 3x3, pad = dil */
static void res_block(const conv_t *c, float *x, float *tmp, float *tmp2, int h, int w, int dil) {
  const long n = (long)SO_C * h * w;
  so_conv2d(x, SO_C, h, w, c[0].w, c[0].b, SO_C, 3, 1, dil, dil, tmp);
  so_lrelu(tmp, n, SO_LRELU);
  so_conv2d(tmp, SO_C, h, w, c[1].w, c[1].b, SO_C, 3, 1, dil, dil, tmp2);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) {
    const float v = x[i] + tmp2[i];
    x[i] = v > 0.0f ? v : v * SO_LRELU;
  }
}

void so_features(const float *weights, const float *planes, int hp, int wp, float *feat) {
  net_t net;
  bind_net(weights, &net);
  int h = hp, w = wp;
  const float *cur = planes;
  float *bufs[2] = {falloc((size_t)SO_C * (hp / 2) * (wp / 2)), falloc((size_t)SO_C * (hp / 4) * (wp / 4))};
  int cin = 3;
  for (int i = 0; i < SO_NDOWN; ++i) {
    float *dst = bufs[i & 1];
    so_conv2d(cur, cin, h, w, net.down[i].w, net.down[i].b, SO_C, 5, 2, 2, 1, dst);
    h /= 2;
    w /= 2;
    cin = SO_C;
    cur = dst;
  }
  /* cur = bufs[1] holds C x hl x wl (SO_NDOWN even) */
  const size_t n = (size_t)SO_C * h * w;
  float *x = falloc(n), *t1 = falloc(n), *t2 = falloc(n);
  memcpy(x, cur, n * sizeof(float));
  for (int i = 0; i < SO_NFRES; ++i) res_block(net.fres[i], x, t1, t2, h, w, 1);
  so_conv2d(x, SO_C, h, w, net.fout.w, net.fout.b, SO_C, 3, 1, 1, 1, feat);
  free(x); free(t1); free(t2); free(bufs[0]); free(bufs[1]);
}

void so_aggregate(const float *weights, const float *fl, const float *fr,
                  int dl, int hl, int wl, float *cost) {
  net_t net;
  bind_net(weights, &net);
  const size_t n = (size_t)SO_C * dl * hl * wl;
  float *a = falloc(n), *b = falloc(n);
  so_cost_volume(fl, fr, SO_C, dl, hl, wl, a);
  for (int i = 0; i < SO_NAGG; ++i) {
    so_conv3d(a, SO_C, dl, hl, wl, net.agg[i].w, net.agg[i].b, SO_C, b);
    so_lrelu(b, (long)n, SO_LRELU);
    float *t = a; a = b; b = t;
  }
  so_conv3d(a, SO_C, dl, hl, wl, net.aout.w, net.aout.b, 1, cost);
  free(a); free(b);
}

void so_refine(const float *weights, const float *disp_up, const float *img,
               int hp, int wp, int dmax, float *disp) {
  so_refine_level(weights, 0, disp_up, img, hp, wp, (float)dmax, disp);
}

/* 2x2 average pooling of c planes (h, w even): the image pyramid of the hierarchical refinement */
void so_avgpool2(const float *in, int c, int h, int w, float *out) {
  const int ho = h / 2, wo = w / 2;
#pragma omp parallel for schedule(static)
  for (int p = 0; p < c * ho; ++p) {
    const int ch = p / ho, y = p - ch * ho;
    const float *r0 = in + ((size_t)ch * h + 2 * y) * w, *r1 = r0 + w;
    float *o = out + ((size_t)ch * ho + y) * wo;
    for (int x = 0; x < wo; ++x) o[x] = ((r0[2 * x] + r0[2 * x + 1]) + (r1[2 * x] + r1[2 * x + 1])) * 0.25f;
  }
}

void so_refine_level(const float *weights, int level, const float *disp_up, const float *img,
                     int hp, int wp, float dnorm, float *disp) {
  net_t net;
  bind_tower(weights, level, &net);
  const size_t plane = (size_t)hp * wp;
  const size_t n = (size_t)SO_C * plane;
  float *in4 = falloc(4 * plane);
  const float inv_d = 1.0f / dnorm;
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)plane; ++i) in4[i] = disp_up[i] * inv_d;
  memcpy(in4 + plane, img, 3 * plane * sizeof(float));
  float *x = falloc(n), *t1 = falloc(n), *t2 = falloc(n);
  so_conv2d(in4, 4, hp, wp, net.rin.w, net.rin.b, SO_C, 3, 1, 1, 1, x);
  so_lrelu(x, (long)n, SO_LRELU);
  for (int i = 0; i < SO_NRRES; ++i) res_block(net.rres[i], x, t1, t2, hp, wp, k_ref_dil[i]);
  so_conv2d(x, SO_C, hp, wp, net.rout.w, net.rout.b, 1, 3, 1, 1, 1, t1);
  const float fd = dnorm;
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)plane; ++i) {
    const float v = disp_up[i] + fd * t1[i];
    disp[i] = v > 0.0f ? v : 0.0f;
  }
  free(in4); free(x); free(t1); free(t2);
}

/* Tensor contract: stereonet_node.cpp:63-72,682-683 (input 1x6xHxW int8 NCHW:
 * L-Y,L-U,L-V,R-Y,R-U,R-V), stereonet_node.cpp:282-288 + parser.cpp:84-86
 * (output 1x1xHxW int32, value*scale*16*12 = disparity in px). */
int so_forward(const float *weights, const int8_t *in6, int w, int h, int dmax,
               float *disp, int32_t *raw, float *disp_low) {
  return so_forward_levels(weights, 1, in6, w, h, dmax, disp, raw, disp_low, NULL);
}

/* levels = 1: single-scale refinement (x16 upsample, one tower).  levels = SO_MULTI_LEVELS: hierarchical refinement
 * (SURVEY.md appendix A `multi`): for k = levels-1 .. 0, at 1/2^k resolution,
 *     up_k = bilinear_x2(d_{k+1}) * 2,   img_k = avgpool_{2^k}(left planes),   D_k = D / 2^k,
 *     d_k  = relu(up_k + D_k * tower_k(cat[up_k / D_k, img_k])),               d_levels = soft-argmin map.
 * level_maps (nullable): levels-1 pointers, level_maps[k-1] receives d_k (hp/2^k x wp/2^k) for k >= 1. */
int so_forward_levels(const float *weights, int levels, const int8_t *in6, int w, int h, int dmax,
                      float *disp, int32_t *raw, float *disp_low, float *const *level_maps) {
  if (levels != 1 && levels != SO_MULTI_LEVELS) return -1;
  if (!weights || !in6 || w <= 0 || h <= 0 || dmax < 16 || dmax % 16) return -1;
  const int wp = (w + 15) / 16 * 16, hp = (h + 15) / 16 * 16;
  const int wl = wp / 16, hl = hp / 16, dl = dmax / 16;
  const size_t pp = (size_t)hp * wp;
  float *planes = (float *)calloc(6 * pp, sizeof(float));
  for (int c = 0; c < 6; ++c)
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x)
        planes[c * pp + (size_t)y * wp + x] = (float)in6[((size_t)c * h + y) * w + x] * (1.0f / 128.0f);
  const size_t nf = (size_t)SO_C * hl * wl;
  float *fl = falloc(nf), *fr = falloc(nf);
  so_features(weights, planes, hp, wp, fl);
  so_features(weights, planes + 3 * pp, hp, wp, fr);
  float *cost = falloc((size_t)dl * hl * wl);
  so_aggregate(weights, fl, fr, dl, hl, wl, cost);
  float *dlow = falloc((size_t)hl * wl);
  so_soft_argmin(cost, dl, hl, wl, dlow);
  if (disp_low) memcpy(disp_low, dlow, (size_t)hl * wl * sizeof(float));
  float *dup = falloc(pp), *dfull = falloc(pp);
  if (levels == 1) {
    so_upsample_bilinear(dlow, hl, wl, 16, 16.0f, dup);
    so_refine(weights, dup, planes, hp, wp, dmax, dfull);
  } else {
    /* image pyramid of the left eye: pyr[k] = 3 x hp/2^k x wp/2^k */
    float *pyr[SO_MULTI_LEVELS];
    pyr[0] = planes;
    for (int k = 1; k < levels; ++k) {
      pyr[k] = falloc((size_t)3 * (hp >> k) * (wp >> k));
      so_avgpool2(pyr[k - 1], 3, hp >> (k - 1), wp >> (k - 1), pyr[k]);
    }
    float *prev = falloc(pp), *cur = falloc(pp);
    memcpy(prev, dlow, (size_t)hl * wl * sizeof(float));
    for (int k = levels - 1; k >= 0; --k) {
      const int hk = hp >> k, wk = wp >> k;
      so_upsample_bilinear(prev, hk / 2, wk / 2, 2, 2.0f, dup);
      so_refine_level(weights, k, dup, pyr[k], hk, wk, (float)dmax / (float)(1 << k), cur);
      if (k >= 1 && level_maps && level_maps[k - 1]) memcpy(level_maps[k - 1], cur, (size_t)hk * wk * sizeof(float));
      float *t = prev; prev = cur; cur = t;
    }
    memcpy(dfull, prev, pp * sizeof(float));
    for (int k = 1; k < levels; ++k) free(pyr[k]);
    free(prev); free(cur);
  }
  /* wire format: raw = lrintf(disp * inv_q), inv_q = float(1 / (16 * 12 * scale)) for EVERY dmax: every consumer of
   * the tensor multiplies by the literal 16 * 12 (parser.cpp:86, stereonet_node.cpp:288, publisher_member_function.py:75) */
  const float inv_q = (float)(1.0 / (16.0 * 12.0 * (double)SO_OUT_SCALE));
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const float v = dfull[(size_t)y * wp + x];
      if (disp) disp[(size_t)y * w + x] = v;
      if (raw) raw[(size_t)y * w + x] = (int32_t)lrintf(v * inv_q);
    }
  free(planes); free(fl); free(fr); free(cost); free(dlow); free(dup); free(dfull);
  return 0;
}
