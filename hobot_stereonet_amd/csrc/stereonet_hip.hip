// stereonet_hip.hip — host engine + C ABI of libstereonet_hip.so (see include/stereonet_hip.h).
//
// Replaces, for the StereoNet hot path, what the reference obtains from the closed dnn_node /
// libdnn runtime: model load (DnnNode::Init, stereonet_infer/src/stereonet_node.cpp:44), tensor
// introspection (:57-103) and DnnNode::Run (:812 async, :968 sync).  No CPU fallback exists: if
// there is no gfx950 device every entry point fails with SN_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#ifndef SN_DIAGNOSTICS
#define SN_DIAGNOSTICS 0      // 1: the precision-ablation switches of scripts/lowres_ablation.py (never in the shipping library)
#endif

#include "../../include/stereonet_hip.h"
#include "sn_internal.h"
#include "sn_kernels.hpp"

namespace {

using namespace sn;

constexpr int kNDown = 4, kNFeatRes = 6, kNAgg = 4, kNRefRes = 6;
constexpr int kRefDil[kNRefRes] = {1, 2, 4, 8, 1, 1};
constexpr float kOutScale = 2.60443857769133e-6f;   // stereonet_node.cpp:282
constexpr double kWireFactor = 16.0 * 12.0;         // parser.cpp:86
constexpr double kAutoEnvelopeSingle = 1.0, kAutoEnvelopeMulti = 2.9;   // sn_auto_envelope_px
constexpr int kMaxPieceEvents = 64;
constexpr int kMaxTowerStreams = 2;

#define HIP_TRY(h, expr)                                                              \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      set_err(h, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
      return SN_ERR_DEVICE;                                                           \
    }                                                                                 \
  } while (0)

// hipMemset runs on the legacy default stream and may return before the device has finished; the engine's streams are
// created hipStreamNonBlocking and do NOT order themselves behind it.  A kernel launched on one of them right after a
// plain hipMemset of its output can therefore be overtaken by the memset (seen once as a parity-hook flake in round 4:
// zeros in a freshly written tensor).  Every memset of a buffer that another stream touches next goes through this.
inline hipError_t memset_now(void* p, int v, size_t bytes) {
  hipError_t e = hipMemset(p, v, bytes);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(nullptr);
}

// Device buffers of a parity hook (sn_dbg_*): freed on EVERY return path, error paths included.
struct DevScope {
  std::vector<void*> ptrs;
  void track(const void* p) {
    void* q = const_cast<void*>(p);
    if (q && std::find(ptrs.begin(), ptrs.end(), q) == ptrs.end()) ptrs.push_back(q);
  }
  ~DevScope() {
    for (void* q : ptrs) (void)hipFree(q);
  }
};

struct ConvLayer {
  uint4* wx3 = nullptr;    // device, split fp16 A-fragments [cin_pad/16][9][hi|lo][64 lanes] (fp16 modes, 3x3 layers)
  float* wpk = nullptr;    // device, packed [cin_pad][taps][32]
  float* bias = nullptr;   // device [32]
  int cin = 0, cin_pad = 0, taps = 0;
};

struct Down0F16 {           // first down-conv on the fp16 MFMA (k_down0_f16): 8 K-steps x (hi, lo) A-fragments
  uint4* wfrag = nullptr;   // device [8][2][64] slots
};

struct Down01W {            // down-convs 0 and 1 folded into one 13x13 stride-4 conv (sn_down01.hpp): nine weight classes
  uint4* wfrag = nullptr;   // device [9][39][2][64] slots
  float* bias = nullptr;    // device [9][32]
};

struct RefLayerF16 {        // fp16 tower layer: 18 MFMA A-fragments + fp32 bias
  uint4* wfrag = nullptr;   // device [9][2][64] slots
  float* bias = nullptr;
};

struct HeadLayer {          // C -> 1 layers (VALU kernels)
  float* w = nullptr;       // device [32][taps]
  float bias = 0.f;
  uint4* pfrag = nullptr;   // agg.out only, fp16 modes: split A fragments of the taps-as-M contraction [2][hi|lo][64] (k_agg_x3s_dma HEADP)
};

constexpr int kMaxLevels = 4, kMultiLevels = 4;              // hierarchical refinement: 1/8, 1/4, 1/2, 1
constexpr int kStatWords = 8;                                // refinement statistic: [level 0..3] sum |D r|, [4] self-check sum |a - b|
constexpr size_t kStatU64 = (size_t)kStatWords * kStatWordStride;   // each word = kStatSlots partial sums in separate 128-byte lines
constexpr int kTileCtrStride = 8 * 16;                       // uints per tower launch (one 64-B line per XCD)
constexpr size_t kTileCtrBytes = (size_t)2 * 6 * kTileCtrStride * sizeof(unsigned);   // 2 * kNRefRes launches

struct Workspace {          // activations for up to `nb` pairs
  int nb = 0, rb = 0, pb = 0;   // batch capacity, pairs per tower launch, pairs per low-res piece
  int rb_x3 = 0;                // SN_PREC_AUTO: pairs per tower launch while the handle runs in SN_PREC_F16X3 (same buffers)
  int tower_cu = 0;             // > 0: workgroups of a streamed tower launch (an async slot that shares the GPU, submit_common)
  int rbk_x3[4] = {};           // ... and per coarse level
  // refinement statistic: one 64-bit fixed-point sum of |D r| per level (refine_stat_commit) + the self-check's sum at [4];
  // copied to the pinned twin at the end of every forward()
  unsigned long long* stats = nullptr;
  unsigned long long* stats_host = nullptr;
  int8_t* in6 = nullptr;
  float* down[3] = {nullptr, nullptr, nullptr};
  float* low[3] = {nullptr, nullptr, nullptr};
  float* feat = nullptr;
  float* vol[2] = {nullptr, nullptr};
  uint4* volp[2] = {nullptr, nullptr};
  uint4* lowp[2] = {nullptr, nullptr};            // zero-bordered (x, t) of the 3x3 feature layers (fp16 modes, FeatPad)
  uint4* downp[3] = {nullptr, nullptr, nullptr};   // zero-bordered inputs of down-convs 1..3 (fp16 modes, DownDma)   // zero-bordered split-slot volumes of the aggregation layers (fp16 modes, VolPad)
  float* cost = nullptr;     // [nb][Dl][hl][wl] (debug / parity)
  float* disp_low = nullptr;
  int ns = 1;                 // tower streams this workspace serves: one (x, t) activation pair per stream
  float* ref[2 * kMaxTowerStreams] = {};
  uint4* ref16[2 * kMaxTowerStreams] = {};   // fp16 NCHW8c padded (fp16 modes): [2 * stream + {x, t}]
  uint4* ref16_raw[2 * kMaxTowerStreams] = {};            // the allocations behind them (alloc_ref16)
  // hierarchical refinement, levels 1..: the coarse levels run once per low-resolution PIECE (pb pairs), in chunks
  // of rbk[level] = min(pb, rb * 4^level) pairs (the same activation footprint per launch as level 0).  Activation
  // pairs (their own zero borders) for rbk pairs; image pyramid [pb][3][Hk][Wk] and level maps [pb][Hk][Wk].
  int rbk[kMaxLevels] = {};
  float* ref_lv[kMaxLevels][2] = {};
  uint4* ref16_lv[kMaxLevels][2] = {};
  uint4* ref16_lv_raw[kMaxLevels][2] = {};
  float* pyr[kMaxLevels] = {};
  float* lvl_disp[kMaxLevels] = {};
  int n_chunks = 0;
  unsigned* tile_ctr = nullptr;           // dynamic tile queues of the fp16 tower: [12 launches][8 XCDs][16] uints
  float* out_disp = nullptr;
  int32_t* out_raw = nullptr;
  uint8_t* nv12 = nullptr;   // staging for NV12 inputs (2 eyes or one side-by-side frame)
};

struct Tower {               // one refinement level: weights (in the forms the precision mode needs) + geometry
  ConvLayer rin, rres[kNRefRes][2];
  Down0F16 refin;
  RefLayerF16 rres16[kNRefRes][2];      // SN_PREC_F16 (and AUTO): plain fp16 A fragments
  RefLayerF16 rres16x3[kNRefRes][2];    // SN_PREC_F16X3 (and AUTO): hi / lo split A fragments
  HeadLayer rout;
  RefGeom rg{};
  int Hk = 0, Wk = 0;        // padded size of this level: Hp >> k, Wp >> k
};

struct Slot {                // async request slot (sn_submit / sn_wait)
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  Workspace ws;
  int8_t* pin_in = nullptr;
  int32_t* pin_raw = nullptr;
  float* pin_disp = nullptr;
  int32_t* user_raw = nullptr;
  float* user_disp = nullptr;
  uint64_t ticket = 0;       // 0 = free
  // hipGraph of {H2D, forward, D2H} per output mask (1 = int32, 2 = float, 3 = both): the second request with a
  // given mask is captured, later ones replay it (the ~45 launches of a single-pair forward are launch-bound)
  // the first index is the input kind: 0 = int8 model tensor (sn_submit), 1 = side-by-side NV12 frame (sn_submit_nv12)
  // the last index is the arithmetic the request runs in (0 = SN_PREC_F16X3, 1 = anything else): SN_PREC_AUTO may change it
  hipGraphExec_t gexec[2][4][4] = {};       // [input kind][output mask][arithmetic x (alone | sharing the GPU)]
  int uses[2][4][4] = {};
  int mode_run = 0;          // arithmetic of the request in flight (SN_PREC_*)
};

// SN_PREC_AUTO (include/stereonet_hip.h): the handle starts in SN_PREC_F16 and moves to SN_PREC_F16X3 when the refinement
// statistic leaves the envelope inside which the fp16 tower keeps EPE <= 1e-3 px, or when the self-check says so.
struct AutoCtl {
  sn_auto_state st{};
  bool calibrated = false;       // the self-check (one pair in both arithmetics) has run since the handle last entered F16
  bool pending = false;          // a stream-enqueued call's statistic has not been folded in yet (ev_stats marks it)
  int pending_mode = 0, pending_n = 0;
  double selfcheck_epe = -1.0, selfcheck_res = -1.0;
  double last_level[4] = {}, last_res = 0.0;
  int last_mode = 0;
  uint64_t calls = 0, pairs = 0, reruns = 0;
};

}  // namespace

struct sn_handle {
  int device = 0;
  int W = 0, H = 0, D = 0, Wp = 0, Hp = 0, wl = 0, hl = 0, Dl = 0;
  int max_batch = 1, precision = SN_PREC_F16, task_num = 4, refine_chunk = 1, piece = 16;
  // `precision` is what the caller configured; SN_PREC_AUTO runs in actl.st.mode (SN_PREC_F16 or SN_PREC_F16X3)
  AutoCtl actl;
  std::mutex mu_cal;         // the self-check's scratch maps (chk) are shared by every slot
  float* chk[2] = {nullptr, nullptr};
  hipEvent_t ev_stats = nullptr;
  int refine_chunk_x3 = 1;   // SN_PREC_AUTO: pairs per tower launch in SN_PREC_F16X3
  hipStream_t stream = nullptr;
  // piece pipeline: the low-resolution branch of piece k+1 runs on s_low while the refinement towers of piece k run
  // on s_tow[]; consecutive tower chunks alternate between the tower streams so that the ramp-up / tail of one
  // chunk's launches is filled by the other chunk's workgroups
  hipStream_t s_low = nullptr, s_tow[kMaxTowerStreams] = {};
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_tow_join[kMaxTowerStreams] = {}, ev_piece[kMaxPieceEvents] = {};
  bool overlap = true;
  int tower_streams = kMaxTowerStreams;
#if SN_DIAGNOSTICS
  unsigned ablate_x = 0;     // SN_ABLATE_X mask (diagnostic build only): layers whose input tensor gets its lo slots zeroed
#else
  static constexpr unsigned ablate_x = 0;      // the shipping library has no ablation code: every test of it folds away
#endif
  bool tail_fuse = true;     // the streamed last block carries the head (tail form); SN_TAIL_FUSE=0: block + k_head_final_f16
  int fuse_mode = 4;         // SN_FUSE: 4 = streaming fused blocks (default), 0 = two launches per block
  unsigned* dump = nullptr;  // 2 KB device scratch: where lanes without an output pixel store (fused head)
  bool use_graphs = true;    // hipGraph replay for the async single-pair path (SN_NO_GRAPH disables)
  bool stream_prio = false;  // the pipeline streams were created with the device's highest priority (sn_create_prio)
  ConvLayer down[kNDown], fres[kNFeatRes][2], fout, agg[kNAgg];
  Down0F16 down0;
  Down01W down01;            // fp16 modes, unless SN_DOWN01=0
  bool fold_down01 = false;
  HeadLayer aout;
  // refinement towers: tw[0] = full resolution (the only one of a single-scale model); a hierarchical ("multi") model
  // has levels = kMultiLevels towers, tw[k] working at 1/2^k resolution
  int levels = 1;
  Tower tw[kMaxLevels];
  int num_cu = 256;
  Workspace ws;
  std::vector<Slot> slots;
  std::mutex mu;
  std::condition_variable cv;
  uint64_t next_ticket = 1;
  // profiling
  bool profiling = false;
  hipEvent_t ev[8] = {};
  hipEvent_t ev_dom[2 * 6] = {};   // profiling: one pair around every streamed block of the first chunk (the dominant kernel)
  int dom_pairs = 0;
  float stage_ms[SN_STAGE_COUNT] = {};
  mutable std::string err;
};

namespace {

void set_err(const sn_handle* h, const std::string& s) {
  if (h) h->err = s;
}
void set_err(std::nullptr_t, const std::string&) {}

template <class T>
hipError_t dalloc(T** p, size_t count) {
  return hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T) + 256);
}

// ---- weight file (hobot_stereonet_amd/weights.py documents the layout) -------------------------
struct SnwHeader {
  char magic[4];
  uint32_t version, width, height, dmax, channels, n_down, n_fres, n_agg, n_rres;
  uint32_t dil[6];
  uint64_t n_params, reserved;
};
static_assert(sizeof(SnwHeader) == 80, "SNW1 header is 80 bytes");

struct HostLayer {
  const float* w;
  const float* b;
  int cout, cin, taps;
};

// Walks the canonical tensor order (spec.layers()).
struct BlobWalker {
  const float* base;
  size_t off = 0;
  HostLayer next(int cout, int cin, int taps) {
    HostLayer l{base + off, nullptr, cout, cin, taps};
    off += (size_t)cout * cin * taps;
    l.b = base + off;
    off += cout;
    return l;
  }
};

size_t tower_param_count() {
  return (size_t)kC * 4 * 9 + kC + (size_t)2 * kNRefRes * (kC * kC * 9 + kC) + kC * 9 + 1;
}

size_t param_count(int levels = 1) {
  size_t n = (size_t)(levels - 1) * tower_param_count();
  for (int i = 0; i < kNDown; ++i) n += (size_t)kC * (i == 0 ? 3 : kC) * 25 + kC;
  n += (size_t)(2 * kNFeatRes + 1) * (kC * kC * 9 + kC);
  n += (size_t)kNAgg * (kC * kC * 27 + kC) + kC * 27 + 1;
  n += (size_t)kC * 4 * 9 + kC + (size_t)2 * kNRefRes * (kC * kC * 9 + kC) + kC * 9 + 1;
  return n;
}

// 2-D conv weights [co][ci][ky][kx] -> packed [ci_pad][tap][co]
int upload_conv2d(sn_handle* h, const HostLayer& l, int ch_multiple, ConvLayer* out) {
  const int cin_pad = (l.cin + ch_multiple - 1) / ch_multiple * ch_multiple;
  std::vector<float> pk((size_t)cin_pad * l.taps * kC, 0.f);
  for (int co = 0; co < kC; ++co)
    for (int ci = 0; ci < l.cin; ++ci)
      for (int t = 0; t < l.taps; ++t)
        pk[((size_t)ci * l.taps + t) * kC + co] = l.w[((size_t)co * l.cin + ci) * l.taps + t];
  out->cin = l.cin;
  out->cin_pad = cin_pad;
  out->taps = l.taps;
  HIP_TRY(h, dalloc(&out->wpk, pk.size()));
  HIP_TRY(h, dalloc(&out->bias, kC));
  HIP_TRY(h, hipMemcpy(out->wpk, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(out->bias, l.b, kC * sizeof(float), hipMemcpyHostToDevice));
  return SN_OK;
}

// 3-D conv weights [co][ci][kz][ky][kx] -> packed [c' = kz*32+ci][tap = ky*3+kx][co]  (96 virtual channels)
int upload_conv3d(sn_handle* h, const HostLayer& l, ConvLayer* out) {
  std::vector<float> pk((size_t)96 * 9 * kC, 0.f);
  for (int co = 0; co < kC; ++co)
    for (int ci = 0; ci < kC; ++ci)
      for (int kz = 0; kz < 3; ++kz)
        for (int t = 0; t < 9; ++t)
          pk[((size_t)(kz * kC + ci) * 9 + t) * kC + co] = l.w[(((size_t)co * kC + ci) * 3 + kz) * 9 + t];
  out->cin = 96;
  out->cin_pad = 96;
  out->taps = 9;
  HIP_TRY(h, dalloc(&out->wpk, pk.size()));
  HIP_TRY(h, dalloc(&out->bias, kC));
  HIP_TRY(h, hipMemcpy(out->wpk, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(out->bias, l.b, kC * sizeof(float), hipMemcpyHostToDevice));
  return SN_OK;
}

// split fp16 A-fragments for k_conv3x3_c32_x3: wv(co, c', tap) is the weight of virtual input channel c'
template <class WV>
int upload_x3(sn_handle* h, int cin_virtual, WV wv, ConvLayer* out, int taps = 9, bool zero_lo = false) {
  const int nchunk = cin_virtual / 16;
  std::vector<_Float16> pk((size_t)nchunk * taps * 2 * 64 * 8);
  for (int ch = 0; ch < nchunk; ++ch)
    for (int tap = 0; tap < taps; ++tap)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int co = lane & 31, c = ch * 16 + 8 * (lane >> 5) + e;
          const float w = wv(co, c, tap);
          const _Float16 hi = (_Float16)w;
          const size_t base = (((size_t)ch * taps + tap) * 2) * 64 * 8 + (size_t)lane * 8 + e;
          pk[base] = hi;
          pk[base + 64 * 8] = zero_lo ? (_Float16)0.f : (_Float16)((w - (float)hi) * kSplitScale);
        }
  HIP_TRY(h, dalloc(&out->wx3, pk.size() / 8));
  HIP_TRY(h, hipMemcpy(out->wx3, pk.data(), pk.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  return SN_OK;
}

// A-fragments of k_down0_f16: K = 8 * rho + kx, rho = ci * 5 + ky (row 15 and kx >= 5 are zero)
int upload_down0_f16(sn_handle* h, const HostLayer& l, Down0F16* out) {
  std::vector<_Float16> pk((size_t)8 * 2 * 64 * 8);
  for (int t = 0; t < 8; ++t)
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 8; ++e) {
        const int co = lane & 31, rho = 2 * t + (lane >> 5);
        const float w = (rho < 15 && e < 5) ? l.w[((size_t)co * 3 + rho / 5) * 25 + (rho % 5) * 5 + e] : 0.f;
        const _Float16 hi = (_Float16)w;
        const size_t base = ((size_t)(2 * t) * 64 + lane) * 8 + e;
        pk[base] = hi;
        pk[base + 64 * 8] = (_Float16)((w - (float)hi) * kSplitScale);
      }
  HIP_TRY(h, dalloc(&out->wfrag, pk.size() / 8));
  HIP_TRY(h, hipMemcpy(out->wfrag, pk.data(), pk.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  return SN_OK;
}

// A-fragments of k_refin_f16: K = 8 * tap + e over the pixel slot [d_hi, Y_hi, U_hi, V_hi, d_lo, Y_lo, U_lo, V_lo]
// (the image lo parts are zero when the source is the int8 model input, whose values are exact in fp16);
// fragment a = hi weights on entries 0..3; fragment b = lo weights on entries 0..3 + hi weights on entries 4..7
int upload_refin_f16(sn_handle* h, const HostLayer& l, Down0F16* out) {
  std::vector<_Float16> pk((size_t)5 * 2 * 64 * 8, (_Float16)0.f);
  for (int t = 0; t < 5; ++t)
    for (int lane = 0; lane < 64; ++lane) {
      const int co = lane & 31, tap = 2 * t + (lane >> 5);
      if (tap >= 9) continue;
      _Float16* a = &pk[((size_t)(2 * t) * 64 + lane) * 8];
      _Float16* b = a + 64 * 8;
      for (int c = 0; c < 4; ++c) {
        const float w = l.w[((size_t)co * 4 + c) * 9 + tap];
        const _Float16 hi = (_Float16)w;
        a[c] = hi;
        b[c] = (_Float16)((w - (float)hi) * kSplitScale);
        b[4 + c] = hi;
      }
    }
  HIP_TRY(h, dalloc(&out->wfrag, pk.size() / 8));
  HIP_TRY(h, hipMemcpy(out->wfrag, pk.data(), pk.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  return SN_OK;
}

// img_src: the int8 model input [n][6][H][W] (pyr = false) or the float image pyramid level [n][3][g.H][g.W].
template <int TW>
hipError_t launch_refin_f16_tw(hipStream_t st, const Down0F16& L, const float* bias, const float* disp_low,
                               const void* img_src, bool pyr, int hl, int wl, int H, int W, float inv_d, UpScale ups,
                               RefGeom g, int nimg, uint4* out, bool split, size_t lo_off_bytes, int num_cu) {
  using T = RefInTile<TW>;
  g.tiles_x = (g.W + TW - 1) / TW;
  g.tiles_y = (g.H + T::TH - 1) / T::TH;
  const int total = g.tiles_x * g.tiles_y * nimg;
  int blocks = 2 * num_cu;
  if (blocks > total) blocks = total;
  const int al4 = !pyr && (W % 4 == 0) && (reinterpret_cast<uintptr_t>(img_src) % 4 == 0);
#define SN_REFIN(S, P)                                                                                             \
  hipLaunchKernelGGL((k_refin_f16<S, P, TW>), dim3(blocks), dim3(256), T::LDS_BYTES, st, disp_low, img_src, hl, wl, H, \
                     W, inv_d, ups, L.wfrag, bias, out, lo_off_bytes, g, nimg, al4)
  if (split && pyr) SN_REFIN(true, true);
  else if (split) SN_REFIN(true, false);
  else if (pyr) SN_REFIN(false, true);
  else SN_REFIN(false, false);
#undef SN_REFIN
  return hipGetLastError();
}

// (8x32 tiles, which pay off for the tower's dilation-1 / -2 launches, were measured for this kernel too: 54.8 us
// against 41.5 us per two pairs at 1280x720 — only 100 of the 256 threads have a staging unit then.)
hipError_t launch_refin_f16(hipStream_t st, const Down0F16& L, const float* bias, const float* disp_low,
                            const void* img_src, bool pyr, int hl, int wl, int H, int W, float inv_d, UpScale ups,
                            const RefGeom& g, int nimg, uint4* out, bool split, size_t lo_off_bytes, int num_cu) {
  return launch_refin_f16_tw<64>(st, L, bias, disp_low, img_src, pyr, hl, wl, H, W, inv_d, ups, g, nimg, out, split,
                                 lo_off_bytes, num_cu);
}

hipError_t launch_down0_f16(hipStream_t st, const Down0F16& L, const float* bias, const int8_t* in6, int H, int W,
                            int nimg, int Ho, int Wo, float* out, int num_cu, const SlotGeom* og = nullptr) {
  constexpr int TC = 32;
  using T = Down0Tile<TC>;
  const int tiles_x = (Wo + TC - 1) / TC, tiles_y = (Ho + T::TR - 1) / T::TR;
  const int total = tiles_x * tiles_y * nimg;
  int blocks = 2 * num_cu;                        // register budget: two workgroups per CU
  if (blocks > total) blocks = total;
  const int al4 = (W % 4 == 0) && (reinterpret_cast<uintptr_t>(in6) % 4 == 0);
  hipLaunchKernelGGL((k_down0_f16<TC>), dim3(blocks), dim3(256), T::LDS_BYTES, st, in6, H, W, L.wfrag, bias, out, Ho, Wo,
                     tiles_x, tiles_y, nimg, 0, al4, og ? og->PH : Ho, og ? og->PW : Wo, og ? og->py : 0, og ? og->px : 0);
  return hipGetLastError();
}

template <class K>
hipError_t ensure_lds_attr(K kern, int bytes);

// SN_DOWN01=0 restores k_down0_f16 + k_down_x3s_dma for the first two down-convs (parity switch)
bool down01_enabled() {
  static const bool on = !(getenv("SN_DOWN01") != nullptr && atoi(getenv("SN_DOWN01")) == 0);
  return on;
}

int upload_down01(sn_handle* h, const HostLayer& l0, const HostLayer& l1, Down01W* out) {
  std::vector<double> weff, beff;
  compose_down01(l0.w, l0.b, l1.w, l1.b, weff, beff);
  std::vector<_Float16> pk;
  pack_down01(weff, pk);
  std::vector<float> bf(beff.begin(), beff.end());
  HIP_TRY(h, dalloc(&out->wfrag, pk.size() / 8));
  HIP_TRY(h, dalloc(&out->bias, bf.size()));
  HIP_TRY(h, hipMemcpy(out->wfrag, pk.data(), pk.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(out->bias, bf.data(), bf.size() * sizeof(float), hipMemcpyHostToDevice));
  return SN_OK;
}

// in6: int8 model input of the piece; out: split-slot tensor of the quarter-resolution map in geometry `og`
hipError_t launch_down01(hipStream_t st, const Down01W& L, const int8_t* in6, int H, int W, int nimg, int Ho, int Wo,
                         uint4* out, const SlotGeom& og, int num_cu) {
  using T = Down01;
  const bool w4 = (W % 4) == 0;
  hipError_t e = w4 ? ensure_lds_attr(k_down01_f16<true>, T::LDS_BYTES) : ensure_lds_attr(k_down01_f16<false>, T::LDS_BYTES);
  if (e != hipSuccess) return e;
  const int tiles_x = (Wo + T::TC - 1) / T::TC, tiles_y = (Ho + T::TR - 1) / T::TR;
  const int total = tiles_x * tiles_y * nimg;
  int blocks = num_cu / 8 * 8;                    // one workgroup per CU (register budget), whole XCD bands
  while (blocks > 8 && blocks / 8 > (total + 7) / 8) blocks -= 8;
  // (dword loads at any byte address: the int8 planes need no alignment, only W % 4 decides which instance runs)
  if (w4)
    hipLaunchKernelGGL(k_down01_f16<true>, dim3(blocks), dim3(256), T::LDS_BYTES, st, in6, H, W,
                       L.wfrag + (size_t)T::INNER * T::NK * 2 * 64, L.bias + T::INNER * kC, out, Ho, Wo, tiles_x, tiles_y, nimg,
                       og.PH, og.PW, og.py, og.px);
  else
    hipLaunchKernelGGL(k_down01_f16<false>, dim3(blocks), dim3(256), T::LDS_BYTES, st, in6, H, W,
                       L.wfrag + (size_t)T::INNER * T::NK * 2 * 64, L.bias + T::INNER * kC, out, Ho, Wo, tiles_x, tiles_y, nimg,
                       og.PH, og.PW, og.py, og.px);
  const int per_img = 4 + 2 * ((Wo - 2 + 31) / 32) + 2 * ((Ho - 2 + 31) / 32);
  if (w4)
    hipLaunchKernelGGL(k_down01_border<true>, dim3(per_img * (nimg / 2)), dim3(192), 0, st, in6, H, W, L.wfrag, L.bias, out,
                       Ho, Wo, nimg, og.PH, og.PW, og.py, og.px);
  else
    hipLaunchKernelGGL(k_down01_border<false>, dim3(per_img * (nimg / 2)), dim3(192), 0, st, in6, H, W, L.wfrag, L.bias, out,
                       Ho, Wo, nimg, og.PH, og.PW, og.py, og.px);
  return hipGetLastError();
}

// weights-stationary split-operand conv on split-slot tensors: persistent grid of MINB workgroups per CU
template <int KS, int STRIDE, int VCH, int TR, int TC, int SEGW, int MINB, bool OUTSLOT, class Loader, bool HASRES = false>
hipError_t launch_conv_x3s(hipStream_t st, const ConvLayer& L, const Loader& ld, int nimg, int Ho, int Wo, float* out,
                           const float* res, bool lrelu, int num_cu) {
  using T = X3sTile<KS, STRIDE, VCH, TR, TC, SEGW>;
  ConvArgs a{};
  a.wpk = reinterpret_cast<const float*>(L.wx3);
  a.bias = L.bias;
  a.out = out;
  a.res = res;
  a.nimg = nimg;
  a.cin_pad = L.cin_pad;
  a.Ho = Ho;
  a.Wo = Wo;
  a.dil = 1;
  a.pad = KS / 2;
  a.lrelu = lrelu ? 1 : 0;
  a.tiles_x = (Wo + TC - 1) / TC;
  a.tiles_y = (Ho + TR - 1) / TR;
  if (res != nullptr && !HASRES)      // residual layers use their own instantiation (16 more registers)
    return launch_conv_x3s<KS, STRIDE, VCH, TR, TC, SEGW, MINB, OUTSLOT, Loader, true>(st, L, ld, nimg, Ho, Wo, out, res, lrelu, num_cu);
  auto kern = k_conv_x3s<KS, STRIDE, VCH, TR, TC, SEGW, MINB, OUTSLOT, HASRES, Loader>;
  static_assert(T::LDS_BYTES <= 160 * 1024, "x3s tile does not fit the LDS");
  if (T::LDS_BYTES > 64 * 1024) {
    hipError_t e = ensure_lds_attr(kern, (int)T::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  const int total = a.tiles_x * a.tiles_y * nimg;
  int blocks = num_cu * MINB;                       // a multiple of 8: one band of tiles per XCD
  if (blocks > (total + 7) / 8 * 8) blocks = (total + 7) / 8 * 8;
  blocks = (blocks + 7) / 8 * 8;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), T::LDS_BYTES, st, a, ld);
  return hipGetLastError();
}

// ---- precision ablation of the low-resolution branch (scripts/lowres_ablation.py) -------------------------------------
// DIAGNOSTIC BUILD ONLY (-DSN_DIAGNOSTICS=1: `python -m hobot_stereonet_amd.build --diag` -> libstereonet_hip_diag.so, which the
// script loads through STEREONET_HIP_LIB); the shipping library ignores both variables.
// The split-operand layers evaluate x*w as xh*wh + (xh*wl + xl*wh) / 2048 (three fp16 MFMAs).  What a cheaper form of a
// layer would compute is reproduced exactly with zeroed operands (an MFMA with a zero operand adds exact zeros):
//   SN_ABLATE_W=<layers>  the layer's weights rounded to fp16: its lo A-fragments are uploaded as zeros  (drops xh*wl)
//   SN_ABLATE_X=<layers>  the layer's input rounded to fp16: the lo slots of its input tensor are zeroed in front of the
//                         launch (drops xl*wh; runs the plain split-slot layouts, which are bit-identical to the
//                         zero-bordered ones; for the first conv of a residual block the rounded tensor is also the
//                         block's residual input, so that entry is an upper bound)
// <layers>: comma-separated names out of down1..down3, f0..f12 (the thirteen 3x3 feature convs), agg0..agg3, or "all".
enum { kAblDown = 0, kAblFeat = 3, kAblAgg = 16, kAblCount = 20 };
#if !SN_DIAGNOSTICS
inline unsigned ablate_mask(const char*) { return 0u; }
inline hipError_t zero_lo_slots(hipStream_t, float*, int, size_t) { return hipSuccess; }
#else
unsigned ablate_mask(const char* var) {
  const char* e = getenv(var);
  if (!e || !*e) return 0u;
  if (!strcmp(e, "all")) return (1u << kAblCount) - 1u;
  unsigned m = 0;
  std::string str(e);
  size_t pos = 0;
  while (pos <= str.size()) {
    size_t c = str.find(',', pos);
    if (c == std::string::npos) c = str.size();
    const std::string t = str.substr(pos, c - pos);
    int idx = -1;
    if (t.rfind("down", 0) == 0 && t.size() == 5 && t[4] >= '1' && t[4] <= '3') idx = kAblDown + (t[4] - '1');
    else if (t.rfind("agg", 0) == 0 && t.size() == 4 && t[3] >= '0' && t[3] <= '3') idx = kAblAgg + (t[3] - '0');
    else if (t.size() >= 2 && t[0] == 'f' && atoi(t.c_str() + 1) >= 0 && atoi(t.c_str() + 1) <= 12 && isdigit((unsigned char)t[1])) idx = kAblFeat + atoi(t.c_str() + 1);
    if (idx >= 0) m |= 1u << idx;
    pos = c + 1;
  }
  return m;
}

// split-slot tensor [nblk][hi | lo][hw] (nblk = images x 4 channel blocks): zero the lo halves
__global__ void k_zero_lo_slots(uint4* t, size_t hw, size_t nblk) {
  const size_t total = nblk * hw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t b = i / hw, r = i - b * hw;
    t[(2 * b + 1) * hw + r] = uint4{0u, 0u, 0u, 0u};
  }
}
inline hipError_t zero_lo_slots(hipStream_t st, float* tensor, int nimg, size_t hw) {
  hipLaunchKernelGGL(k_zero_lo_slots, dim3(1024), dim3(256), 0, st, reinterpret_cast<uint4*>(tensor), hw, (size_t)nimg * 4);
  return hipGetLastError();
}
#endif      // SN_DIAGNOSTICS

// SN_AGG_DMA=0: aggregation layers on the plain split-slot volumes (k_conv_x3s) instead of the zero-bordered ones
bool agg_dma_enabled() {
  static const bool on = [] {
    const char* e = getenv("SN_AGG_DMA");
    return !(e && *e == '0') && ablate_mask("SN_ABLATE_X") == 0;
  }();
  return on;
}

VolPad vol_pad(int Dl, int hl, int wl) { return VolPad{Dl, hl, wl, VolPad::ph(hl), VolPad::pw(wl)}; }

// 3x3x3 aggregation layer on zero-bordered split-slot volumes (sn_agg_dma.hpp): one persistent workgroup per CU
// head_frag != nullptr (OUTSLOT = false): the layer ends in the output conv's taps-as-M contraction and writes its partial
// sums P [npairs Dl][27][H][W] instead of the activated volume (k_agg_x3s_dma HEADP)
template <bool OUTSLOT, bool HEADP = false>
hipError_t launch_agg_dma(hipStream_t st, const ConvLayer& L, const uint4* vin, const VolPad& g, int npairs, void* out,
                          bool lrelu, int num_cu, const uint4* head_frag = nullptr) {
  ConvArgs a{};
  a.wpk = reinterpret_cast<const float*>(L.wx3);
  a.bias = L.bias;
  a.out = reinterpret_cast<float*>(out);
  a.res = reinterpret_cast<const float*>(head_frag);
  if (HEADP != (head_frag != nullptr)) return hipErrorInvalidValue;
  a.nimg = npairs * g.Dl;
  a.cin_pad = L.cin_pad;
  a.Ho = g.H;
  a.Wo = g.W;
  a.dil = 1;
  a.pad = 1;
  a.lrelu = lrelu ? 1 : 0;
  a.tiles_x = (g.W + 15) / 16;
  a.tiles_y = (g.H + 7) / 8;
  auto kern = k_agg_x3s_dma<OUTSLOT, HEADP>;
  hipError_t e = ensure_lds_attr(kern, (int)AggDma::LDS_BYTES);
  if (e != hipSuccess) return e;
  const int total = a.tiles_x * a.tiles_y * a.nimg;
  int blocks = num_cu;
  if (blocks > (total + 7) / 8 * 8) blocks = (total + 7) / 8 * 8;
  blocks = (blocks + 7) / 8 * 8;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), AggDma::LDS_BYTES, st, a, vin, g);
  return hipGetLastError();
}

// SN_FEAT_DMA=0: the 3x3 feature layers on the plain split-slot tensors (k_conv_x3s) instead of the zero-bordered ones
bool feat_dma_enabled() {
  static const bool on = [] {
    const char* e = getenv("SN_FEAT_DMA");
    return !(e && *e == '0') && ablate_mask("SN_ABLATE_X") == 0;
  }();
  return on;
}

FeatPad feat_pad(int H, int W) { return FeatPad{H, W, FeatPad::ph(H), FeatPad::pw(W)}; }

// 3x3 32->32 feature layer on zero-bordered split-slot tensors (sn_feat_dma.hpp): two persistent workgroups per CU.
// out / res: FeatPad tensors (OUTSLOT) or fp32 NCHW.
template <bool OUTSLOT, bool HASRES>
hipError_t launch_feat_dma(hipStream_t st, const ConvLayer& L, const uint4* vin, const FeatPad& g, int nimg, void* out,
                           const void* res, bool lrelu, int num_cu) {
  ConvArgs a{};
  a.wpk = reinterpret_cast<const float*>(L.wx3);
  a.bias = L.bias;
  a.out = reinterpret_cast<float*>(out);
  a.res = reinterpret_cast<const float*>(res);
  a.nimg = nimg;
  a.cin_pad = L.cin_pad;
  a.Ho = g.H;
  a.Wo = g.W;
  a.dil = 1;
  a.pad = 1;
  a.lrelu = lrelu ? 1 : 0;
  a.tiles_x = (g.W + 15) / 16;
  a.tiles_y = (g.H + 7) / 8;
  auto kern = k_feat_x3s_dma<OUTSLOT, HASRES>;
  hipError_t e = ensure_lds_attr(kern, (int)FeatDma::LDS_BYTES);
  if (e != hipSuccess) return e;
  const int total = a.tiles_x * a.tiles_y * a.nimg;
  int blocks = 2 * num_cu;
  if (blocks > (total + 7) / 8 * 8) blocks = (total + 7) / 8 * 8;
  blocks = (blocks + 7) / 8 * 8;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), FeatDma::LDS_BYTES, st, a, vin, g);
  return hipGetLastError();
}

// SN_DOWN_DMA=0: down-convs 1..3 on the plain split-slot tensors (k_conv_x3s) instead of the zero-bordered ones
bool down_dma_enabled() {
  static const bool on = [] {
    const char* e = getenv("SN_DOWN_DMA");
    return !(e && *e == '0') && ablate_mask("SN_ABLATE_X") == 0;
  }();
  return on;
}

// zero-bordered input grid of a 5x5 stride-2 down-conv with an Ho x Wo output
SlotGeom down_in_geom(int Ho, int Wo) { return SlotGeom{DownDma::ph(Ho), DownDma::pw(Wo), DownDma::PADY, DownDma::PADX}; }

// 5x5 stride-2 32->32 down-conv on a zero-bordered split-slot input (sn_agg_dma.hpp); go = the output tensor's grid
hipError_t launch_down_dma(hipStream_t st, const ConvLayer& L, const uint4* vin, int nimg, int Ho, int Wo, void* out,
                           const SlotGeom& go, bool lrelu, int num_cu) {
  ConvArgs a{};
  a.wpk = reinterpret_cast<const float*>(L.wx3);
  a.bias = L.bias;
  a.out = reinterpret_cast<float*>(out);
  a.res = nullptr;
  a.nimg = nimg;
  a.cin_pad = L.cin_pad;
  a.Ho = Ho;
  a.Wo = Wo;
  a.dil = 1;
  a.pad = 2;
  a.lrelu = lrelu ? 1 : 0;
  a.tiles_x = (Wo + DownDma::TC - 1) / DownDma::TC;
  a.tiles_y = (Ho + DownDma::TR - 1) / DownDma::TR;
  hipError_t e = ensure_lds_attr(k_down_x3s_dma, (int)DownDma::LDS_BYTES);
  if (e != hipSuccess) return e;
  const int total = a.tiles_x * a.tiles_y * nimg;
  int blocks = num_cu;
  if (blocks > (total + 7) / 8 * 8) blocks = (total + 7) / 8 * 8;
  blocks = (blocks + 7) / 8 * 8;
  hipLaunchKernelGGL(k_down_x3s_dma, dim3(blocks), dim3(256), DownDma::LDS_BYTES, st, a, vin, down_in_geom(Ho, Wo), go);
  return hipGetLastError();
}

int upload_head(sn_handle* h, const HostLayer& l, HeadLayer* out) {   // [1][32][taps] as-is
  HIP_TRY(h, dalloc(&out->w, (size_t)kC * l.taps));
  HIP_TRY(h, hipMemcpy(out->w, l.w, (size_t)kC * l.taps * sizeof(float), hipMemcpyHostToDevice));
  out->bias = l.b[0];
  return SN_OK;
}

// agg.out as the A operand of P[tap][pixel] = sum_c w[c][tap] y[c][pixel] (k_agg_x3s_dma<false, true>): row m = tap
// (27 of 32 rows), K-step kk = channels 16 kk .. 16 kk + 15, lane (m, g) holds channels 16 kk + 8 g + e; hi / lo split
int upload_agg_head_frag(sn_handle* h, const HostLayer& l, HeadLayer* out) {
  std::vector<_Float16> pk((size_t)2 * 2 * 64 * 8, (_Float16)0.f);
  for (int kk = 0; kk < 2; ++kk)
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 8; ++e) {
        const int m = lane & 31, c = 16 * kk + 8 * (lane >> 5) + e;
        if (m >= 27) continue;
        const float w = l.w[(size_t)c * 27 + m];
        const _Float16 hi = (_Float16)w;
        const size_t base = ((size_t)(2 * kk) * 64 + lane) * 8 + e;
        pk[base] = hi;
        pk[base + 64 * 8] = (_Float16)((w - (float)hi) * kSplitScale);
      }
  HIP_TRY(h, dalloc(&out->pfrag, pk.size() / 8));
  HIP_TRY(h, hipMemcpy(out->pfrag, pk.data(), pk.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  return SN_OK;
}

// SN_HEAD_FOLD=0: the last aggregation layer writes its 32-channel volume and k_head_softargmin contracts it (parity switch)
bool head_fold_enabled() {
  static const bool on = !(getenv("SN_HEAD_FOLD") != nullptr && atoi(getenv("SN_HEAD_FOLD")) == 0);
  return on;
}

// Raise a kernel's dynamic-LDS limit once per (kernel, device) — not per launch: launches may happen inside a
// stream capture.  Keyed by the kernel's address (different instantiations can share one function type).
template <class K>
hipError_t ensure_lds_attr(K kern, int bytes) {
  static std::mutex mu;
  static std::unordered_map<const void*, unsigned long long> done;     // kernel -> bitmask of device ordinals
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  const void* key = reinterpret_cast<const void*>(kern);
  std::lock_guard<std::mutex> lk(mu);
  unsigned long long& m = done[key];
  if (m & bit) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) m |= bit;
  return e;
}

// ---- convolution launcher ------------------------------------------------------------------------
template <int KS, int STRIDE, int DIL, int CH, int TR, int TC, class Loader, bool PF = true, int MINW = 1>
hipError_t launch_conv(hipStream_t st, const ConvLayer& L, const Loader& ld, int nimg, int Ho, int Wo,
                       float* out, const float* res, bool lrelu) {
  constexpr int dil = DIL;
  ConvArgs a{};
  a.wpk = L.wpk;
  a.bias = L.bias;
  a.out = out;
  a.res = res;
  a.nimg = nimg;
  a.cin_pad = L.cin_pad;
  a.Ho = Ho;
  a.Wo = Wo;
  a.dil = dil;
  a.pad = (KS / 2) * dil;
  a.lrelu = lrelu ? 1 : 0;
  a.tiles_x = (Wo + TC - 1) / TC;
  a.tiles_y = (Ho + TR - 1) / TR;
  const int rows_in = (TR - 1) * STRIDE + (KS - 1) * dil + 1;
  const int cols_in = (TC - 1) * STRIDE + (KS - 1) * dil + 1;
  const int pitch = STRIDE == 1 ? cols_in : 2 * ((cols_in + 1) / 2);
  const size_t lds = ((size_t)CH * KS * KS * 32 + (size_t)CH * rows_in * pitch) * sizeof(float);
  auto kern = k_conv_c32_mfma<KS, STRIDE, DIL, CH, TR, TC, Loader, PF, MINW>;
  if (lds > 64 * 1024) {
    hipError_t e = ensure_lds_attr(kern, (int)lds);
    if (e != hipSuccess) return e;
  }
  const int nwg = a.tiles_x * a.tiles_y * nimg;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, st, a, ld);
  return hipGetLastError();
}

// 3x3 C->C conv on a plain NCHW fp32 tensor; tile shape chosen from image size and dilation
template <int DIL>
hipError_t conv3x3_d(hipStream_t st, const ConvLayer& L, const float* in, int nimg, int H, int W, float* out,
                     const float* res, bool lrelu) {
  LoadF32 ld{in, kC, H, W};
  // (chunk sizes 8/16 and tile heights 4/8 measured equal within noise on the 45x80 low-resolution maps)
  if (H * W <= 64 * 128) return launch_conv<3, 1, DIL, 8, 4, 32>(st, L, ld, nimg, H, W, out, res, lrelu);
  if (DIL >= 4) return launch_conv<3, 1, DIL, 4, 16, 64>(st, L, ld, nimg, H, W, out, res, lrelu);
  return launch_conv<3, 1, DIL, 8, 8, 64>(st, L, ld, nimg, H, W, out, res, lrelu);
}

// Tower layers of SN_PREC_FP32 (plain fp32 NCHW, 32 -> 32, 3x3 dilated): the weights-stationary kernel of sn_tower_f32.hpp.
template <int DIL, int CPH, int NW = 8>
hipError_t launch_ref_conv_f32(hipStream_t st, const ConvLayer& L, const float* in, int nimg, int H, int W, float* out,
                               const float* res, bool lrelu, int num_cu) {
  using T = F32Tile<DIL, CPH, 64, NW>;
  auto kern = res ? k_ref_conv_f32<DIL, CPH, true, NW> : k_ref_conv_f32<DIL, CPH, false, NW>;
  if (T::LDS_BYTES > 64 * 1024 - 1024) {
    hipError_t e = ensure_lds_attr(kern, T::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  const int total = ((W + T::TW - 1) / T::TW) * ((H + T::TH - 1) / T::TH) * nimg;
  // persistent: ONE workgroup per CU (it double-buffers its own staging, sn_tower_f32.hpp), each walks a contiguous share of
  // the launch's HALF tiles (1280x720, one pair = 3600 halves on 256 CUs = 14.06 per workgroup instead of 8 whole tiles for
  // 7.03 tiles of work)
  int grid = 2 * total < num_cu ? 2 * total : num_cu;
  static const int grid_env = getenv("SN_F32_GRID") ? atoi(getenv("SN_F32_GRID")) : 0;          // probe: workgroups per launch
  if (grid_env > 0) grid = grid_env < 2 * total ? grid_env : 2 * total;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), T::LDS_BYTES, st, in, out, res, L.wpk, L.bias, nimg, H, W, lrelu ? 1 : 0);
  return hipGetLastError();
}

// SN_F32_TOWER=0 keeps the generic kernel for the tower layers too (A/B)
inline bool f32_tower_env() {
  static const bool on = !(getenv("SN_F32_TOWER") != nullptr && atoi(getenv("SN_F32_TOWER")) == 0);
  return on;
}

hipError_t conv3x3(hipStream_t st, const ConvLayer& L, const float* in, int nimg, int H, int W, int dil,
                   float* out, const float* res, bool lrelu, int tower_cu = 0) {
  // (the 16-byte staging wants rows that start 16-byte aligned: W % 4 == 0 — every level-0 geometry, not every coarse
  // level of a hierarchical model)
  if (tower_cu > 0 && L.cin == kC && L.cin_pad == kC && (W & 3) == 0 && f32_tower_env()) {
    switch (dil) {
      case 1: return launch_ref_conv_f32<1, 8>(st, L, in, nimg, H, W, out, res, lrelu, tower_cu);
      case 2: return launch_ref_conv_f32<2, 8>(st, L, in, nimg, H, W, out, res, lrelu, tower_cu);
      case 4: return launch_ref_conv_f32<4, 4, 16>(st, L, in, nimg, H, W, out, res, lrelu, tower_cu);
      // dilation 4 / 8: sixteen rows per tile (1024 threads) — 16 / 24 halo rows per 8 would be 2 - 3x the staging of the tile itself
      case 8: return launch_ref_conv_f32<8, 4, 16>(st, L, in, nimg, H, W, out, res, lrelu, tower_cu);
      default: return hipErrorInvalidValue;
    }
  }
  switch (dil) {
    case 1: return conv3x3_d<1>(st, L, in, nimg, H, W, out, res, lrelu);
    case 2: return conv3x3_d<2>(st, L, in, nimg, H, W, out, res, lrelu);
    case 4: return conv3x3_d<4>(st, L, in, nimg, H, W, out, res, lrelu);
    case 8: return conv3x3_d<8>(st, L, in, nimg, H, W, out, res, lrelu);
    default: return hipErrorInvalidValue;
  }
}

hipError_t conv5x5s2(hipStream_t st, const ConvLayer& L, const float* in, int nimg, int Hin, int Win,
                     float* out) {
  LoadF32 ld{in, kC, Hin, Win};
  const int Ho = Hin / 2, Wo = Win / 2;
  // 8 x 64 tiles are the efficient shape, but a launch needs workgroups: a single pair's second down-conv is 56 of them on
  // 256 CUs (130 us for a quarter of the first one's work, profiles/r05_fp32_b1_kernel_summary.txt); below two workgroups
  // per CU the 4 x 32 shape (same K order, same sums) fills the chip instead
  const long big_tiles = (long)((Wo + 63) / 64) * ((Ho + 7) / 8) * nimg;
  if (Ho * Wo <= 64 * 128 || big_tiles < 512) return launch_conv<5, 2, 1, 4, 4, 32>(st, L, ld, nimg, Ho, Wo, out, nullptr, false);
  return launch_conv<5, 2, 1, 4, 8, 64>(st, L, ld, nimg, Ho, Wo, out, nullptr, false);
}

// ---- fp16 refinement tower -------------------------------------------------------------------------
RefGeom make_ref_geom(int Hp, int Wp) {
  RefGeom g{};
  g.tiles_x = (Wp + 63) / 64;
  g.tiles_y = (Hp + 7) / 8;
  g.H = Hp;
  g.W = Wp;
  g.Hs = (Hp + 15) / 16 * 16 + 2 * kRefPad;      // whole 16-row tiles (tall-tile variants of the dilated layers)
  g.Ws = g.tiles_x * 64 + 2 * kRefPad;
  return g;
}

size_t ref16_slots(const RefGeom& g, int nimg) { return (size_t)nimg * 4 * g.Hs * g.Ws; }
// The streaming block kernels the pipeline instantiates (ref_block_stream below): ONE list, from which the zero rows
// around a tensor are derived.
using StreamTile1 = StreamTile<1, 64, 4, 6, 4>;
using StreamTile2 = StreamTile<2, 64, 4, 6, 4>;
using StreamTile4 = StreamTile<4, 128, 2, 6, 4>;
using StreamTile8 = StreamTile<8, 128, 2, 6, 4>;
using StreamTileTail = StreamTile<1, 64, 4, 5, 4, true>;      // last block + refinement head (x ring of 5 groups: early residual fetch)
static_assert(StreamTileTail::ROWS_ABOVE <= kRefPad && StreamTileTail::ROWS_BELOW <= StreamTile8::ROWS_BELOW, "tail form stays inside the zero rows");
constexpr int cmax4(int a, int b, int c, int d) { return (a > b ? a : b) > (c > d ? c : d) ? (a > b ? a : b) : (c > d ? c : d); }
constexpr int kStreamRowsAbove = cmax4(StreamTile1::ROWS_ABOVE, StreamTile2::ROWS_ABOVE, StreamTile4::ROWS_ABOVE, StreamTile8::ROWS_ABOVE);
constexpr int kStreamRowsBelow = cmax4(StreamTile1::ROWS_BELOW, StreamTile2::ROWS_BELOW, StreamTile4::ROWS_BELOW, StreamTile8::ROWS_BELOW);
// Slots behind a tensor that kernels may over-read (never written, zero).  Streaming kernel: a DMA group reaches up to
// ROWS_BELOW image rows below the last image row, of which the tensor itself holds Hs - kRefPad - H >= kRefPad; a group
// whose columns run past Ws wraps into the next row (+1).  The per-layer kernels over-read < 4096 slots.
constexpr int kRefSlackRows = kStreamRowsBelow - kRefPad + 1;
// Slots IN FRONT of a tensor (zero, never written): a strip's first group starts ROWS_ABOVE image rows above row 0
// (16 at dilation 8) and up to 2 DIL columns left of column 0, where the tensor's own border is kRefPad rows / columns
// (a column underrun wraps into the previous row: +1).  An fp16 activation tensor is allocated as
// [front | tensor | slack] and handed around by the address of `tensor`.
constexpr int kRefFrontRows = (kStreamRowsAbove > kRefPad ? kStreamRowsAbove - kRefPad : 0) + 1;
static_assert(kRefSlackRows == 24 && kRefFrontRows == 9, "zero rows around the fp16 tower tensors follow the StreamTile list");
size_t ref_slack(const RefGeom& g) {
  const size_t rows = (size_t)kRefSlackRows * g.Ws;
  return rows > 4096 ? rows : 4096;
}
size_t ref_front(const RefGeom& g) { return (size_t)kRefFrontRows * g.Ws; }
hipError_t alloc_ref16(const RefGeom& g, size_t tensor_and_slack_slots, uint4** raw, uint4** base) {
  const size_t front = ref_front(g), all = front + tensor_and_slack_slots;
  hipError_t e = dalloc(raw, all);
  if (e != hipSuccess) return e;
  e = memset_now(*raw, 0, all * sizeof(uint4));        // the zero borders are never written again
  *base = *raw + front;
  return e;
}

// [co][ci][ky][kx] fp32 -> wfrag[tap][kk][lane][e] fp16 = w[co = lane&31][ci = 16kk + 8(lane>>5) + e][tap]
// ---- fp16 weights of the tower (SN_PREC_F16): sum-preserving rounding of every 3x3 kernel ----------------------------
// Rounding each weight to nearest leaves every (cout, cin) kernel with a sum error of ~sqrt(9) half-ulps.  The tower's
// activations are LeakyReLU outputs: positive mean, and smooth wherever the image is — so a kernel's response to them is
// mostly (sum of its taps) x (local mean), and the sum errors of the 32 x 32 x 12 kernels add up COHERENTLY over the whole
// image into an offset of the refinement residual: 2.4e-4 ... 1.1e-3 px at D = 192 depending on the weight draw and the
// image content, the largest single term of the mode's error (scripts/f16_error_sources.py,
// profiles/r05_f16_error_sources.txt).  Here each kernel's nine taps are rounded down or up (never further than the two
// neighbouring fp16 numbers) in the combination, out of the 512, whose SUM of errors is smallest: the offset disappears
// (< 2e-5 px in the same experiment), at the price of individual tap errors of up to one ulp instead of half — which only
// the high-frequency part of the activations sees.  Costs nothing at run time, needs no calibration data; weights that
// are exact in fp16 stay as they are.  SN_W_ROUND=rne restores round-to-nearest (A/B switch).
inline _Float16 f16_neighbour(_Float16 hval, bool up) {
  uint16_t b;
  memcpy(&b, &hval, 2);
  if (up) {
    if (b == 0x8000) b = 0x0001;
    else if (b & 0x8000) b -= 1;
    else b += 1;
  } else {
    if (b == 0x0000) b = 0x8001;
    else if (b & 0x8000) b += 1;
    else b -= 1;
  }
  _Float16 r;
  memcpy(&r, &b, 2);
  return r;
}

// w[9] -> q[9]: q[t] is one of the two fp16 numbers enclosing w[t]
void round_kernel_sum_preserving(const float* w, _Float16* q) {
  double lo[9], hi[9];
  _Float16 hlo[9], hhi[9];
  for (int t = 0; t < 9; ++t) {
    const _Float16 n = (_Float16)w[t];
    const double nd = (double)n, wd = (double)w[t];
    hlo[t] = nd <= wd ? n : f16_neighbour(n, false);
    hhi[t] = nd >= wd ? n : f16_neighbour(n, true);
    lo[t] = (double)hlo[t] - wd;        // <= 0
    hi[t] = (double)hhi[t] - wd;        // >= 0
  }
  int best = 0;
  double best_score = 1e300;
  for (int m = 0; m < 512; ++m) {
    double sum = 0, sq = 0;
    for (int t = 0; t < 9; ++t) {
      const double e = (m >> t) & 1 ? hi[t] : lo[t];
      sum += e;
      sq += e * e;
    }
    const double score = std::fabs(sum) + 1e-3 * std::sqrt(sq);      // sum first; among (near-)ties the smallest errors
    if (score < best_score) {
      best_score = score;
      best = m;
    }
  }
  for (int t = 0; t < 9; ++t) q[t] = (best >> t) & 1 ? hhi[t] : hlo[t];
}

bool w_round_sum_preserving() {      // read at every sn_create (not cached): scripts/epe_sensitivity.py compares the two in one process
  const char* e = getenv("SN_W_ROUND");
  return !(e && strcmp(e, "rne") == 0);
}

int upload_ref_f16(sn_handle* h, const HostLayer& l, RefLayerF16* out) {
  std::vector<_Float16> q((size_t)kC * kC * 9);
  const bool sp = w_round_sum_preserving();
  for (size_t k = 0; k < (size_t)kC * kC; ++k) {
    if (sp) {
      round_kernel_sum_preserving(l.w + k * 9, &q[k * 9]);
    } else {
      for (int t = 0; t < 9; ++t) q[k * 9 + t] = (_Float16)l.w[k * 9 + t];
    }
  }
  std::vector<_Float16> pk((size_t)18 * 64 * 8);
  for (int tap = 0; tap < 9; ++tap)
    for (int kk = 0; kk < 2; ++kk)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int co = lane & 31, ci = 16 * kk + 8 * (lane >> 5) + e;
          pk[(((size_t)tap * 2 + kk) * 64 + lane) * 8 + e] = q[((size_t)co * kC + ci) * 9 + tap];
        }
  HIP_TRY(h, dalloc(&out->wfrag, (size_t)18 * 64));
  HIP_TRY(h, dalloc(&out->bias, kC));
  HIP_TRY(h, hipMemcpy(out->wfrag, pk.data(), pk.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(out->bias, l.b, kC * sizeof(float), hipMemcpyHostToDevice));
  return SN_OK;
}

// F16X3: [co][ci][ky][kx] fp32 -> hi fragments (18 x 64 slots) followed by lo fragments, lo = fp16((w - hi) * 2^11)
int upload_ref_f16x3(sn_handle* h, const HostLayer& l, RefLayerF16* out) {
  std::vector<_Float16> pk((size_t)36 * 64 * 8);
  for (int tap = 0; tap < 9; ++tap)
    for (int kk = 0; kk < 2; ++kk)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int co = lane & 31, ci = 16 * kk + 8 * (lane >> 5) + e;
          const float w = l.w[((size_t)co * kC + ci) * 9 + tap];
          const _Float16 hi = (_Float16)w;
          const size_t i = (((size_t)tap * 2 + kk) * 64 + lane) * 8 + e;
          pk[i] = hi;
          pk[(size_t)18 * 64 * 8 + i] = (_Float16)((w - (float)hi) * kSplitScale);
        }
  HIP_TRY(h, dalloc(&out->wfrag, (size_t)36 * 64));
  HIP_TRY(h, dalloc(&out->bias, kC));
  HIP_TRY(h, hipMemcpy(out->wfrag, pk.data(), pk.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(out->bias, l.b, kC * sizeof(float), hipMemcpyHostToDevice));
  return SN_OK;
}

template <int DIL, int TW, int NBUF>
hipError_t launch_ref_f16x3(hipStream_t st, const RefLayerF16& L, const RefGeom& g, int num_cu, const uint4* in,
                            uint4* out, const uint4* res, size_t lo_slots, int nimg, bool lrelu) {
  using T = RefTile2<DIL, TW>;
  constexpr int lds_bytes = NBUF * 2 * T::BUF * 16;
  auto kern = res ? k_ref_conv_f16x3<DIL, TW, NBUF, true> : k_ref_conv_f16x3<DIL, TW, NBUF, false>;
  hipError_t e = ensure_lds_attr(kern, lds_bytes);
  if (e != hipSuccess) return e;
  RefGeom gt = g;
  gt.tiles_x = (g.W + TW - 1) / TW;
  const int total = gt.tiles_x * gt.tiles_y * nimg;
  const int band = (total + 7) / 8;
  int cap = num_cu / 8;                      // one workgroup per CU (two on 8 x 32 tiles measured -6 %: the kernel is
                                             // memory bound and spills at 256 registers, profiles/r06_x3_wpc_ab.txt)
  if (cap < 1) cap = 1;
  const int rounds = (band + cap - 1) / cap;
  const int nlb = (band + rounds - 1) / rounds;
  hipLaunchKernelGGL(kern, dim3(nlb * 8), dim3(256), lds_bytes, st, in, out, res, lo_slots, L.wfrag, L.bias, gt, nimg,
                     lrelu ? 1 : 0);
  return hipGetLastError();
}

hipError_t ref_conv_f16x3(hipStream_t st, const RefLayerF16& L, const RefGeom& g, int num_cu, int dil, const uint4* in,
                          uint4* out, const uint4* res, size_t lo_slots, int nimg, bool lrelu) {
  switch (dil) {
    case 1: return launch_ref_f16x3<1, 64, 3>(st, L, g, num_cu, in, out, res, lo_slots, nimg, lrelu);
    case 2: return launch_ref_f16x3<2, 64, 3>(st, L, g, num_cu, in, out, res, lo_slots, nimg, lrelu);
    case 4: return launch_ref_f16x3<4, 32, 3>(st, L, g, num_cu, in, out, res, lo_slots, nimg, lrelu);
    case 8: return launch_ref_f16x3<8, 32, 2>(st, L, g, num_cu, in, out, res, lo_slots, nimg, lrelu);
    default: return hipErrorInvalidValue;
  }
}

template <int DIL, int TW, int NB = 3>
hipError_t launch_ref_f16_v2(hipStream_t st, const RefLayerF16& L, const RefGeom& g, int num_cu, const uint4* in,
                             uint4* out, const uint4* res, int nimg, bool lrelu, unsigned* tile_ctr) {
  using T = RefTile2<DIL, TW, 8, NB>;
  // tile_ctr: the launch's tile queue (8 zeroed counters, 64 B apart)
  auto kern = res ? k_ref_conv_f16_v2<DIL, TW, true, 8, 2, NB> : k_ref_conv_f16_v2<DIL, TW, false, 8, 2, NB>;
  if (tile_ctr == nullptr) return hipErrorInvalidValue;
  if (T::LDS_BYTES > 64 * 1024) {
    hipError_t e = ensure_lds_attr(kern, T::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  static_assert(2 * T::LDS_BYTES <= 160 * 1024, "two tower workgroups per CU");
  RefGeom gt = g;                      // tile grid of this variant (the buffer geometry is for 8x64 tiles)
  gt.tiles_x = (g.W + TW - 1) / TW;
  gt.tiles_y = (g.H + 7) / 8;
  const int total = gt.tiles_x * gt.tiles_y * nimg;
  // persistent grid: 8 XCD bands, two workgroups per CU, every slot filled (the queue balances the bands)
  const int band = (total + 7) / 8;
  int cap = num_cu * 2 / 8;
  if (cap < 1) cap = 1;
  const int nlb = cap < band ? cap : band;
  hipLaunchKernelGGL(kern, dim3(nlb * 8), dim3(256), T::LDS_BYTES, st, in, out, res, L.wfrag, L.bias, gt, nimg,
                     lrelu ? 1 : 0, tile_ctr);
  return hipGetLastError();
}

// Fused residual block, row-streaming form (sn_stream_block.hpp): one 512-thread workgroup per CU walks its share of
// the flattened (image, row phase, strip, sub-row) sequence.  x and y must be different tensors.  dump: >= 1 KB scratch.
template <class T>
hipError_t launch_ref_block_stream(hipStream_t st, const RefLayerF16& L1, const RefLayerF16& L2, const RefGeom& g, int num_cu,
                                   const uint4* x, uint4* y, int nimg, unsigned* dump, const StreamHeadArgs& ha = StreamHeadArgs{}) {
  constexpr int DIL = T::DIL;
  auto kern = k_ref_block_stream_f16<T::DIL, T::TW, T::R, T::NXS, T::NWR, T::HEAD>;
  if (dump == nullptr) return hipErrorInvalidValue;
  hipError_t e = ensure_lds_attr(kern, T::LDS_BYTES);
  if (e != hipSuccess) return e;
  StreamSched sc;
  // the tail form walks the OUTPUT maps (H x W of the head, <= the tensor's valid area), the others the whole tensor
  const int Wn = T::HEAD ? ha.W : g.W, Hn = T::HEAD ? ha.H : g.H;
  if (T::HEAD && (!ha.w || !ha.disp_low || (!ha.out_disp && !ha.out_raw) || ha.W > g.W || ha.H > g.H || ha.ups.rs > 0.5f)) return hipErrorInvalidValue;
  sc.nstrips = (Wn + T::OW - 1) / T::OW;
  sc.hsub = (Hn + DIL - 1) / DIL;
  sc.total_rows = nimg * DIL * sc.nstrips * sc.hsub;
  static const int wg_env = getenv("SN_STREAM_WGS") ? atoi(getenv("SN_STREAM_WGS")) : 0;     // experiment switch
  int nwg = wg_env > 0 ? wg_env : num_cu;
  if (nwg > sc.total_rows) nwg = sc.total_rows;
  if (nwg < 1) nwg = 1;
  sc.rows_per_wg = (sc.total_rows + nwg - 1) / nwg;
  const int grid = (sc.total_rows + sc.rows_per_wg - 1) / sc.rows_per_wg;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), T::LDS_BYTES, st, x, y, L1.wfrag, L1.bias, L2.wfrag, L2.bias, g, sc,
                     reinterpret_cast<uint4*>(dump), ha);
  return hipGetLastError();
}

// Last block of the tower + the refinement head in one launch (tail form): y never leaves the CU, the head's maps are the
// only thing written.  Same arithmetic as the streamed block followed by k_head_final_f16 (bit-identical maps).
hipError_t ref_block_stream_tail(hipStream_t st, const RefLayerF16& L1, const RefLayerF16& L2, const RefGeom& g, int num_cu,
                                 const uint4* x, int nimg, unsigned* dump, const StreamHeadArgs& ha) {
  return launch_ref_block_stream<StreamTileTail>(st, L1, L2, g, num_cu, x, nullptr, nimg, dump, ha);
}

// Strip shapes: 64 columns x 4 rows per step for dilation 1 / 2 (62 / 60 of 64 columns are outputs); 128 columns x 2 rows
// for dilation 4 / 8, where a 64-wide strip would keep only 56 / 48 of its columns (120 / 112 of 128 here).
hipError_t ref_block_stream(hipStream_t st, const RefLayerF16& L1, const RefLayerF16& L2, const RefGeom& g, int num_cu, int dil,
                            const uint4* x, uint4* y, int nimg, unsigned* dump) {
  switch (dil) {
    case 1: return launch_ref_block_stream<StreamTile1>(st, L1, L2, g, num_cu, x, y, nimg, dump);
    case 2: return launch_ref_block_stream<StreamTile2>(st, L1, L2, g, num_cu, x, y, nimg, dump);
    case 4: return launch_ref_block_stream<StreamTile4>(st, L1, L2, g, num_cu, x, y, nimg, dump);
    case 8: return launch_ref_block_stream<StreamTile8>(st, L1, L2, g, num_cu, x, y, nimg, dump);
    default: return hipErrorInvalidValue;
  }
}
// SN_STREAM_DIL: largest dilation that runs through the streaming kernel (default 8 = every block; 2 = round-3a behaviour)
inline bool stream_block_supports(int dil) {
  static const int max_dil = getenv("SN_STREAM_DIL") ? atoi(getenv("SN_STREAM_DIL")) : 8;
  return (dil == 1 || dil == 2 || dil == 4 || dil == 8) && dil <= max_dil;
}

hipError_t launch_head_final_f16(hipStream_t st, bool split, const uint4* x, size_t lo_slots, const RefGeom& g,
                                 const float* w, float bias, const float* disp_low, int hl, int wl, int H, int W, float dmax,
                                 float inv_q, UpScale ups, float* out_disp, int32_t* out_raw, int nimg,
                                 unsigned long long* stat = nullptr) {
  constexpr int TH = 16;
  using T = HeadTile<TH>;
  const int tiles_x = (W + T::TWO - 1) / T::TWO, tiles_y = (H + TH - 1) / TH;
  const dim3 grid((unsigned)(tiles_x * tiles_y * nimg));
  if (split)
    hipLaunchKernelGGL((k_head_final_f16<true, TH>), grid, dim3(256), T::LDS_BYTES, st, x, lo_slots, g, w, bias, disp_low, hl,
                       wl, H, W, dmax, inv_q, out_disp, out_raw, tiles_x, tiles_y, ups, stat);
  else
    hipLaunchKernelGGL((k_head_final_f16<false, TH>), grid, dim3(256), T::LDS_BYTES, st, x, (size_t)0, g, w, bias, disp_low,
                       hl, wl, H, W, dmax, inv_q, out_disp, out_raw, tiles_x, tiles_y, ups, stat);
  return hipGetLastError();
}

// SN_FUSE: how the residual blocks of the fp16 tower run.  4 (default) = the row-streaming fused kernel for the
// dilations it supports, 0 = two launches per block.
int fuse_env() {
  static const int mode = getenv("SN_FUSE") != nullptr ? atoi(getenv("SN_FUSE")) : 4;
  return mode;
}

// Tile width of a dilation-1 / -2 launch.  The persistent grid (two workgroups per CU) works through the tiles in
// rounds and the launch lasts ceil(tiles / workgroups) rounds: 1280x720, two pairs = 3600 8x64 tiles on 512
// workgroups = 7.03 -> 8 rounds, 12 % of the launch spent on 16 leftover tiles.  8x32 tiles cost ~3 % more per pixel
// (per-tile waits and barriers, 34/32 instead of 66/64 halo columns) but quantise twice as finely (14.06 -> 15
// half-rounds = 7.5): measured +1.7 % end to end at 1280x720, +0.7 % at 1248x384.  Chosen per launch from the
// tile count; force_tw (parity hooks): 64 or 32.
inline int tower_tile_width(const RefGeom& g, int nimg, int num_cu, int force_tw) {
  if (force_tw == 32 || force_tw == 64) return force_tw;
  const long wgs = 2L * num_cu;
  const long rows = (g.H + 7) / 8;
  const long r64 = ((long)((g.W + 63) / 64) * rows * nimg + wgs - 1) / wgs;
  const long r32 = ((long)((g.W + 31) / 32) * rows * nimg + wgs - 1) / wgs;
  return (double)r32 * 0.5 * 1.03 < (double)r64 ? 32 : 64;
}

hipError_t ref_conv_f16(hipStream_t st, const RefLayerF16& L, const RefGeom& g, int num_cu, int dil, const uint4* in,
                        uint4* out, const uint4* res, int nimg, bool lrelu, unsigned* tile_ctr, int force_tw = 0) {
  if (dil <= 2 && tower_tile_width(g, nimg, num_cu, force_tw) == 32) {
    if (dil == 1) return launch_ref_f16_v2<1, 32>(st, L, g, num_cu, in, out, res, nimg, lrelu, tile_ctr);
    if (dil == 2) return launch_ref_f16_v2<2, 32>(st, L, g, num_cu, in, out, res, nimg, lrelu, tile_ctr);
  }
  switch (dil) {
    case 1: return launch_ref_f16_v2<1, 64>(st, L, g, num_cu, in, out, res, nimg, lrelu, tile_ctr);
    case 2: return launch_ref_f16_v2<2, 64>(st, L, g, num_cu, in, out, res, nimg, lrelu, tile_ctr);
    case 4: return launch_ref_f16_v2<4, 32>(st, L, g, num_cu, in, out, res, nimg, lrelu, tile_ctr);      // 61 KB ring
    case 8: return launch_ref_f16_v2<8, 32, 2>(st, L, g, num_cu, in, out, res, nimg, lrelu, tile_ctr);   // 74 KB, two buffers
    default: return hipErrorInvalidValue;
  }
}

// One residual block of the fp16 tower on `*cur` (input and, on return, output); `*oth` is scratch.  tile_ctr: the
// block's two tile queues (kTileCtrStride apart).
hipError_t ref_block_f16(hipStream_t st, const RefLayerF16& L1, const RefLayerF16& L2, const RefGeom& g, int num_cu,
                         int dil, uint4** cur, uint4** oth, int nimg, unsigned* tile_ctr, int fuse_mode, unsigned* dump,
                         bool alt = false) {
  hipError_t e = hipErrorInvalidValue;
  bool fused = false;
  if (fuse_mode == 4 && stream_block_supports(dil)) {
    e = ref_block_stream(st, L1, L2, g, num_cu, dil, *cur, *oth, nimg, dump);
    fused = true;
  }
  if (fused) {
    uint4* t = *cur;
    *cur = *oth;
    *oth = t;
    return e;
  }
  e = ref_conv_f16(st, L1, g, num_cu, dil, *cur, *oth, nullptr, nimg, true, tile_ctr);
  if (e != hipSuccess) return e;
  RefGeom g2 = g;
  if (alt) g2.rev ^= 1;                  // the second conv walks the tiles the other way round (refine_level)
  return ref_conv_f16(st, L2, g2, num_cu, dil, *oth, *cur, *cur, nimg, true, tile_ctr + kTileCtrStride);   // in-place residual
}

// Fused residual block on split operands, row-streaming form (sn_stream_block_x3.hpp): every dilation.  x and y are
// different hi tensors, the lo tensors sit lo_slots behind them.  SN_X3_STREAM=0 keeps two k_ref_conv_f16x3 launches (A/B).
inline bool stream_x3_supports(int dil) {
  static const bool on = !(getenv("SN_X3_STREAM") != nullptr && atoi(getenv("SN_X3_STREAM")) == 0);
  return on && (dil == 1 || dil == 2 || dil == 4 || dil == 8);
}
template <int DIL, int NWR = 2>
hipError_t launch_ref_block_stream_x3(hipStream_t st, const RefLayerF16& L1, const RefLayerF16& L2, const RefGeom& g, int num_cu,
                                      const uint4* x, uint4* y, size_t lo_slots, int nimg) {
  using T = StreamTileX3<DIL, 64, 2, 5, NWR>;
  static_assert(T::ROWS_ABOVE <= kStreamRowsAbove && T::ROWS_BELOW <= kStreamRowsBelow, "inside the zero rows the tensors are allocated with");
  auto kern = k_ref_block_stream_x3<T::DIL, T::TW, T::R, T::NXS, T::NWR>;
  hipError_t e = ensure_lds_attr(kern, T::LDS_BYTES);
  if (e != hipSuccess) return e;
  StreamSched sc;
  sc.nstrips = (g.W + T::OW - 1) / T::OW;
  sc.hsub = (g.H + DIL - 1) / DIL;
  sc.total_rows = nimg * DIL * sc.nstrips * sc.hsub;
  int nwg = num_cu;
  if (nwg > sc.total_rows) nwg = sc.total_rows;
  if (nwg < 1) nwg = 1;
  sc.rows_per_wg = (sc.total_rows + nwg - 1) / nwg;
  const int grid = (sc.total_rows + sc.rows_per_wg - 1) / sc.rows_per_wg;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * T::NWR), T::LDS_BYTES, st, x, y, lo_slots * 16, L1.wfrag, L1.bias, L2.wfrag,
                     L2.bias, g, sc);
  return hipGetLastError();
}
// One residual block of the split tower on `*cur` (input and, on return, output); `*oth` is scratch.
hipError_t ref_block_f16x3(hipStream_t st, const RefLayerF16& L1, const RefLayerF16& L2, const RefGeom& g, int num_cu, int dil,
                           uint4** cur, uint4** oth, size_t lo_slots, int nimg, bool stream = true) {
  if (stream && stream_x3_supports(dil)) {
    hipError_t e = hipErrorInvalidValue;
    // waves per role: 4 (default) = two waves per SIMD, one of each role, so that one role's epilogue / DMA issue sits beside the
    // other's MFMAs (256 registers per wave: 20 bytes of scratch at dilation 1 / 2); SN_X3_NWR=2 = one wave per SIMD (A/B)
    static const int nwr = getenv("SN_X3_NWR") ? atoi(getenv("SN_X3_NWR")) : 4;
    if (nwr == 4) {
      if (dil == 1) e = launch_ref_block_stream_x3<1, 4>(st, L1, L2, g, num_cu, *cur, *oth, lo_slots, nimg);
      else if (dil == 2) e = launch_ref_block_stream_x3<2, 4>(st, L1, L2, g, num_cu, *cur, *oth, lo_slots, nimg);
      else if (dil == 4) e = launch_ref_block_stream_x3<4, 4>(st, L1, L2, g, num_cu, *cur, *oth, lo_slots, nimg);
      else e = launch_ref_block_stream_x3<8, 4>(st, L1, L2, g, num_cu, *cur, *oth, lo_slots, nimg);
    } else if (dil == 1) e = launch_ref_block_stream_x3<1>(st, L1, L2, g, num_cu, *cur, *oth, lo_slots, nimg);
    else if (dil == 2) e = launch_ref_block_stream_x3<2>(st, L1, L2, g, num_cu, *cur, *oth, lo_slots, nimg);
    else if (dil == 4) e = launch_ref_block_stream_x3<4>(st, L1, L2, g, num_cu, *cur, *oth, lo_slots, nimg);
    else e = launch_ref_block_stream_x3<8>(st, L1, L2, g, num_cu, *cur, *oth, lo_slots, nimg);
    uint4* t = *cur;
    *cur = *oth;
    *oth = t;
    return e;
  }
  hipError_t e = ref_conv_f16x3(st, L1, g, num_cu, dil, *cur, *oth, nullptr, lo_slots, nimg, true);
  if (e != hipSuccess) return e;
  return ref_conv_f16x3(st, L2, g, num_cu, dil, *oth, *cur, *cur, lo_slots, nimg, true);      // in-place residual
}

// ---- workspace -----------------------------------------------------------------------------------
// Pairs per low-resolution piece and per tower launch of level k for a workspace of nb pairs with rb pairs per
// full-resolution launch: ONE definition shared by alloc_ws (buffer sizes) and sn_create's 32-bit offset guard.
inline int piece_pairs(const sn_handle* h, int nb, int rb) {
  int pb = h->piece > 0 ? h->piece : 16;
  if (pb > nb) pb = nb;
  if (pb < rb) pb = rb;
  return pb;
}
inline int level_chunk_pairs(int rb, int pb, int lv) {     // coarse level lv runs rb * 4^lv pairs per launch, at most a piece
  const long r = (long)rb << (2 * lv);
  return lv == 0 ? rb : (r < pb ? (int)r : pb);
}

// rb_x3: pairs per tower launch while an SN_PREC_AUTO handle runs in SN_PREC_F16X3 (0 = rb: every other precision)
int alloc_ws(sn_handle* h, Workspace* ws, int nb, int rb, int ns, int rb_x3 = 0) {
  const bool is_auto = h->precision == SN_PREC_AUTO;
  if (rb_x3 <= 0 || rb_x3 > rb) rb_x3 = rb;
  ws->nb = nb;
  ws->rb = rb;
  ws->rb_x3 = rb_x3;
  ws->ns = (ns > 1 && nb > rb) ? (ns < kMaxTowerStreams ? ns : kMaxTowerStreams) : 1;
  if (h->levels > 1) ws->ns = 1;      // the level maps of a piece live in one buffer set: one tower stream
  ws->pb = piece_pairs(h, nb, rb);
  const int pb = ws->pb;
  const size_t HW = (size_t)h->H * h->W, HWp = (size_t)h->Hp * h->Wp, hw = (size_t)h->hl * h->wl;
  HIP_TRY(h, dalloc(&ws->in6, (size_t)nb * 6 * HW));
  // fp16 modes keep the tensors between the down-convs in the zero-bordered layout (downp[], below) unless SN_DOWN_DMA=0
  // or a tensor would not fit 32-bit byte offsets; the plain ones are then not allocated at all (0.9 GB per 16-pair piece)
  bool padded_down = h->precision != SN_PREC_FP32 && down_dma_enabled();
  size_t downp_bytes[3] = {0, 0, 0};
  // (folded down-convs 0 + 1: the half-resolution tensor never exists, so its size cannot veto the zero-bordered layout)
  for (int k = h->fold_down01 ? 1 : 0; k < 3 && padded_down; ++k) {      // input of down-conv k + 1: output grid (Hp, Wp) >> (k + 2)
    const SlotGeom g = down_in_geom(h->Hp >> (k + 2), h->Wp >> (k + 2));
    downp_bytes[k] = (size_t)2 * pb * 8 * g.PH * g.PW * sizeof(uint4);
    padded_down = downp_bytes[k] < ((size_t)1 << 32);
  }
  const int k_first = h->fold_down01 ? 1 : 0;       // folded down-convs 0 + 1: the half-resolution tensor never exists
  for (int k = k_first; k < 3 && !padded_down; ++k)
    HIP_TRY(h, dalloc(&ws->down[k], (size_t)2 * pb * kC * (HWp >> (2 * (k + 1)))));
  for (int k = 0; k < 3; ++k) HIP_TRY(h, dalloc(&ws->low[k], (size_t)2 * pb * kC * hw));
  HIP_TRY(h, dalloc(&ws->feat, (size_t)2 * pb * kC * hw));
  for (int k = 0; k < 2; ++k) HIP_TRY(h, dalloc(&ws->vol[k], (size_t)pb * h->Dl * kC * hw));
  for (int k = k_first; k < 3 && padded_down; ++k) {
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&ws->downp[k]), downp_bytes[k]));
    HIP_TRY(h, memset_now(ws->downp[k], 0, downp_bytes[k]));   // the borders stay zero: kernels write image pixels only
  }
  if (padded_down && feat_dma_enabled()) {       // (the last down-conv writes straight into the bordered layout)
    const FeatPad g = feat_pad(h->hl, h->wl);
    const size_t bytes = (size_t)2 * pb * g.img_slots() * sizeof(uint4);
    for (int k = 0; k < 2 && bytes < ((size_t)1 << 32); ++k) {
      HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&ws->lowp[k]), bytes));
      HIP_TRY(h, memset_now(ws->lowp[k], 0, bytes));       // the borders stay zero: kernels write image pixels only
    }
  }
  if (h->precision != SN_PREC_FP32 && agg_dma_enabled()) {
    const VolPad g = vol_pad(h->Dl, h->hl, h->wl);
    const size_t bytes = g.planes(pb) * g.plane_slots() * sizeof(uint4);
    // the kernel addresses the volume with 32-bit byte offsets; a piece that large keeps the plain volumes
    for (int k = 0; k < 2 && bytes < ((size_t)1 << 32); ++k) {
      HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&ws->volp[k]), bytes));
      HIP_TRY(h, memset_now(ws->volp[k], 0, bytes));       // the borders stay zero: kernels write image pixels only
    }
  }
  HIP_TRY(h, dalloc(&ws->cost, (size_t)nb * h->Dl * hw));
  HIP_TRY(h, dalloc(&ws->disp_low, (size_t)nb * hw));
  // hi tensor (+ lo tensor behind it in SN_PREC_F16X3).  An AUTO handle keeps the fp16 layout for rb pairs and puts the lo
  // tensor of its (smaller) split chunks BEHIND that region: the split mode's hi tensor then sits where the fp16 tensors of
  // the first pairs do — same image pixels, same zero borders — and the lo tensor never touches a border of the fp16
  // layout (with the lo tensor directly behind rb_x3 pairs its pixels landed on the zero borders of pair rb_x3's fp16 plane:
  // the first fp16 call after a split call then read non-zero padding — caught by tests/test_gpu_auto.py)
  auto tensor_slots = [&](const RefGeom& rg, int pairs, int pairs_x3) {
    const size_t one = ref16_slots(rg, pairs) + ref_slack(rg), lo = ref16_slots(rg, pairs_x3) + ref_slack(rg);
    return h->precision == SN_PREC_F16X3 ? 2 * one : (is_auto ? one + lo : one);
  };
  if (h->precision == SN_PREC_FP32) {
    for (int k = 0; k < 2 * ws->ns; ++k) HIP_TRY(h, dalloc(&ws->ref[k], (size_t)rb * kC * HWp));
  } else {
    for (int k = 0; k < 2 * ws->ns; ++k)
      HIP_TRY(h, alloc_ref16(h->tw[0].rg, tensor_slots(h->tw[0].rg, rb, rb_x3), &ws->ref16_raw[k], &ws->ref16[k]));
    // fine-grained: the queue words must be coherent across the 8 XCD L2s at device scope and with the memset
    // one counter block per tower chunk of a forward(): chunks never straddle a low-resolution piece, so every
    // piece may end with one short chunk (forward() numbers the chunks with a running ordinal)
    // a hierarchical model adds the coarse-level launches of every piece: one block per (piece, level, coarse chunk)
    ws->n_chunks = (nb + rb_x3 - 1) / rb_x3 + (nb + pb - 1) / pb + 2;
    for (int lv = 1; lv < h->levels; ++lv) {
      const int rbk = level_chunk_pairs(rb_x3, pb, lv);
      ws->n_chunks += ((nb + pb - 1) / pb + 2) * ((pb + rbk - 1) / rbk + 1);
    }
    HIP_TRY(h, hipExtMallocWithFlags(reinterpret_cast<void**>(&ws->tile_ctr), kTileCtrBytes * ws->n_chunks, hipDeviceMallocFinegrained));
  }
  ws->rbk[0] = rb;
  ws->rbk_x3[0] = rb_x3;
  for (int lv = 1; lv < h->levels; ++lv) {
    const Tower& T = h->tw[lv];
    const size_t HWk = (size_t)T.Hk * T.Wk;
    ws->rbk[lv] = level_chunk_pairs(rb, pb, lv);
    ws->rbk_x3[lv] = level_chunk_pairs(rb_x3, pb, lv);
    for (int k = 0; k < 2; ++k) {
      if (h->precision == SN_PREC_FP32) {
        HIP_TRY(h, dalloc(&ws->ref_lv[lv][k], (size_t)ws->rbk[lv] * kC * HWk));
      } else {
        HIP_TRY(h, alloc_ref16(T.rg, tensor_slots(T.rg, ws->rbk[lv], ws->rbk_x3[lv]), &ws->ref16_lv_raw[lv][k], &ws->ref16_lv[lv][k]));
      }
    }
    HIP_TRY(h, dalloc(&ws->pyr[lv], (size_t)pb * 3 * HWk));
    HIP_TRY(h, dalloc(&ws->lvl_disp[lv], (size_t)pb * HWk));
  }
  HIP_TRY(h, dalloc(&ws->out_disp, (size_t)nb * HW));
  HIP_TRY(h, dalloc(&ws->out_raw, (size_t)nb * HW));
  HIP_TRY(h, dalloc(&ws->nv12, (size_t)HW * 3));
  HIP_TRY(h, dalloc(&ws->stats, kStatU64));
  HIP_TRY(h, memset_now(ws->stats, 0, kStatU64 * sizeof(unsigned long long)));
  HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&ws->stats_host), kStatU64 * sizeof(unsigned long long), hipHostMallocDefault));
  memset(ws->stats_host, 0, kStatU64 * sizeof(unsigned long long));
  return SN_OK;
}

void free_ws(Workspace* ws) {
  hipFree(ws->in6);
  hipFree(ws->tile_ctr);
  for (auto p : ws->down) hipFree(p);
  for (auto p : ws->low) hipFree(p);
  hipFree(ws->feat);
  for (auto p : ws->vol) hipFree(p);
  for (auto p : ws->volp) hipFree(p);
  for (auto p : ws->downp) hipFree(p);
  for (auto p : ws->lowp) hipFree(p);
  hipFree(ws->cost);
  hipFree(ws->disp_low);
  for (auto p : ws->ref) hipFree(p);
  for (auto p : ws->ref16_raw) hipFree(p);
  for (auto& lv : ws->ref_lv)
    for (auto p : lv) hipFree(p);
  for (auto& lv : ws->ref16_lv_raw)
    for (auto p : lv) hipFree(p);
  for (auto p : ws->pyr) hipFree(p);
  for (auto p : ws->lvl_disp) hipFree(p);
  hipFree(ws->out_disp);
  hipFree(ws->out_raw);
  hipFree(ws->nv12);
  hipFree(ws->stats);
  if (ws->stats_host) hipHostFree(ws->stats_host);
  *ws = Workspace();
}

// ---- the forward pass on device buffers ------------------------------------------------------------
// Low-resolution branch for pairs [p0, p0+m): Siamese features -> cost volume -> 3-D aggregation ->
// soft-argmin.  Intermediate buffers are piece-local; disp_low (and cost) are indexed by p0.
// Low-resolution branch of the fp16 modes on split-slot activations (SlotIn): every layer's epilogue writes the
// hi/lo fp16 pair its consumer's split-operand MFMAs read, the weights-stationary kernel stages them as plain
// 16-byte copies.  Only the tensors other kernels read stay fp32 NCHW: the feature map (cost-volume loader, parity
// hook) and the last aggregation volume (soft-argmin head).
int lowres_slots(sn_handle* h, Workspace& ws, hipStream_t st, int p0, int m, const int8_t* in6, bool want_cost,
                 bool prof) {
  const int Hp = h->Hp, Wp = h->Wp, hl = h->hl, wl = h->wl, Dl = h->Dl;
  const size_t HW = (size_t)h->H * h->W;
  const int8_t* in = in6 + (size_t)p0 * 6 * HW;
  const int ncu = h->num_cu, ni = 2 * m;
  auto U4 = [](float* p) { return reinterpret_cast<const uint4*>(p); };
  if (ws.downp[1] != nullptr) {       // zero-bordered tensors between the down-convs, LDS-DMA kernel
    SlotGeom gin[3];
    for (int i = 0; i < 3; ++i) gin[i] = down_in_geom(Hp >> (i + 2), Wp >> (i + 2));
    if (h->fold_down01)      // down-convs 0 and 1 as one 13x13 stride-4 conv straight from the int8 input (sn_down01.hpp)
      HIP_TRY(h, launch_down01(st, h->down01, in, h->H, h->W, ni, Hp / 4, Wp / 4, ws.downp[1], gin[1], ncu));
    else
      HIP_TRY(h, launch_down0_f16(st, h->down0, h->down[0].bias, in, h->H, h->W, ni, Hp / 2, Wp / 2,
                                  reinterpret_cast<float*>(ws.downp[0]), ncu, &gin[0]));
    for (int i = h->fold_down01 ? 1 : 0; i < 3; ++i) {
      const int Ho = Hp >> (i + 2), Wo = Wp >> (i + 2);
      const SlotGeom plain{Ho, Wo, 0, 0};
      const FeatPad fp = feat_pad(hl, wl);
      const SlotGeom bordered{fp.PH, fp.PW, 1, 1};           // the feature layers' zero-bordered layout (sn_feat_dma.hpp)
      void* const last = ws.lowp[0] ? (void*)ws.lowp[0] : (void*)ws.low[0];
      HIP_TRY(h, launch_down_dma(st, h->down[i + 1], ws.downp[i], ni, Ho, Wo, i < 2 ? (void*)ws.downp[i + 1] : last,
                                 i < 2 ? gin[i + 1] : (ws.lowp[0] ? bordered : plain), false, ncu));
    }
  } else {
  if (h->fold_down01)
    HIP_TRY(h, launch_down01(st, h->down01, in, h->H, h->W, ni, Hp / 4, Wp / 4, reinterpret_cast<uint4*>(ws.down[1]),
                             SlotGeom{Hp / 4, Wp / 4, 0, 0}, ncu));
  else
    HIP_TRY(h, launch_down0_f16(st, h->down0, h->down[0].bias, in, h->H, h->W, ni, Hp / 2, Wp / 2, ws.down[0], ncu));
  {
    float* src[3] = {ws.down[0], ws.down[1], ws.down[2]};
    float* dst[3] = {ws.down[1], ws.down[2], ws.low[0]};
    for (int i = h->fold_down01 ? 1 : 0; i < 3; ++i) {
      const int Hi = Hp >> (i + 1), Wi = Wp >> (i + 1);
      if ((h->ablate_x >> (kAblDown + i)) & 1u) HIP_TRY(h, zero_lo_slots(st, src[i], ni, (size_t)Hi * Wi));
      SlotIn ld{U4(src[i]), 0, Hi, Wi};
      HIP_TRY(h, (launch_conv_x3s<5, 2, 32, 4, 32, 32, 1, true, SlotIn>(st, h->down[i + 1], ld, ni, Hi / 2, Wi / 2, dst[i],
                                                                       nullptr, false, ncu)));
    }
  }
  }
  if (ws.lowp[0] != nullptr && ws.downp[1] != nullptr) {     // zero-bordered (x, t), LDS-DMA kernel
    const FeatPad fp = feat_pad(hl, wl);
    uint4 *x = ws.lowp[0], *t = ws.lowp[1];
    // (one launch per layer: a single launch for all twelve with per-image group barriers in device memory was built and
    // measured in round 5 — bit-identical, 609 us instead of 167 us per 16-pair piece: an agent-scope hand-off costs several
    // kernel boundaries, DESIGN.md §5d, profiles/r05_feat_chain_ab.txt)
    for (int i = 0; i < kNFeatRes; ++i) {
      HIP_TRY(h, (launch_feat_dma<true, false>(st, h->fres[i][0], x, fp, ni, t, nullptr, true, ncu)));
      HIP_TRY(h, (launch_feat_dma<true, true>(st, h->fres[i][1], t, fp, ni, x, x, true, ncu)));     // in-place residual
    }
    HIP_TRY(h, (launch_feat_dma<false, false>(st, h->fout, x, fp, ni, ws.feat, nullptr, false, ncu)));
  } else {
  float* x = ws.low[0];
  float* t = ws.low[1];
  for (int i = 0; i < kNFeatRes; ++i) {
    SlotIn lx{U4(x), 0, hl, wl}, lt{U4(t), 0, hl, wl};
    if ((h->ablate_x >> (kAblFeat + 2 * i)) & 1u) HIP_TRY(h, zero_lo_slots(st, x, ni, (size_t)hl * wl));
    HIP_TRY(h, (launch_conv_x3s<3, 1, 32, 8, 16, 16, 2, true, SlotIn>(st, h->fres[i][0], lx, ni, hl, wl, t, nullptr, true, ncu)));
    if ((h->ablate_x >> (kAblFeat + 2 * i + 1)) & 1u) HIP_TRY(h, zero_lo_slots(st, t, ni, (size_t)hl * wl));
    HIP_TRY(h, (launch_conv_x3s<3, 1, 32, 8, 16, 16, 2, true, SlotIn>(st, h->fres[i][1], lt, ni, hl, wl, x, x, true, ncu)));
  }
  {
    if ((h->ablate_x >> (kAblFeat + 12)) & 1u) HIP_TRY(h, zero_lo_slots(st, x, ni, (size_t)hl * wl));
    SlotIn lx{U4(x), 0, hl, wl};
    HIP_TRY(h, (launch_conv_x3s<3, 1, 32, 8, 16, 16, 1, false, SlotIn>(st, h->fout, lx, ni, hl, wl, ws.feat, nullptr, false, ncu)));
  }
  }
  if (prof) HIP_TRY(h, hipEventRecord(h->ev[1], st));
  // cost volume -> slots (vol[1]), then every aggregation layer reads slots: agg0 vol[1] -> vol[0], agg1 -> vol[1], ...
  // (zero-bordered volumes volp[] and the LDS-DMA kernel by default; the last layer writes fp32 into vol[] either way)
  if (ws.volp[0] != nullptr) {
    const VolPad g = vol_pad(Dl, hl, wl);
    const long total = (long)m * Dl * 4 * hl * wl;
    hipLaunchKernelGGL(k_cost_slots_pad, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws.feat, ws.volp[1], g, m);
    for (int i = 0; i < kNAgg; ++i) {
      const uint4* src = ws.volp[(i + 1) & 1];
      if (i + 1 < kNAgg)
        HIP_TRY(h, launch_agg_dma<true>(st, h->agg[i], src, g, m, ws.volp[i & 1], true, ncu));
      else if (head_fold_enabled() && h->aout.pfrag)     // the output conv's contraction rides on this layer's epilogue
        HIP_TRY(h, (launch_agg_dma<false, true>(st, h->agg[i], src, g, m, ws.vol[i & 1], true, ncu, h->aout.pfrag)));
      else
        HIP_TRY(h, launch_agg_dma<false>(st, h->agg[i], src, g, m, ws.vol[i & 1], true, ncu));
    }
    if (head_fold_enabled() && h->aout.pfrag) {       // soft-argmin on the partial sums P [m Dl][27][hl][wl]
      const int npix = m * hl * wl;
      hipLaunchKernelGGL(k_softargmin_p<16>, dim3((npix + 63) / 64), dim3(64 * Dl), 0, st, ws.vol[(kNAgg - 1) & 1], h->aout.bias,
                         Dl, hl, wl, npix, ws.disp_low + (size_t)p0 * hl * wl,
                         want_cost ? ws.cost + (size_t)p0 * Dl * hl * wl : nullptr);
      HIP_TRY(h, hipGetLastError());
      return SN_OK;
    }
  } else {
  {
    const long total = (long)m * Dl * 4 * hl * wl;
    hipLaunchKernelGGL(k_cost_slots, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws.feat,
                       reinterpret_cast<uint4*>(ws.vol[1]), Dl, hl, wl, m);
    if ((h->ablate_x >> kAblAgg) & 1u) HIP_TRY(h, zero_lo_slots(st, ws.vol[1], m * Dl, (size_t)hl * wl));
    SlotIn lc{U4(ws.vol[1]), Dl, hl, wl};
    HIP_TRY(h, (launch_conv_x3s<3, 1, 96, 8, 16, 16, 1, true, SlotIn>(st, h->agg[0], lc, m * Dl, hl, wl, ws.vol[0], nullptr, true, ncu)));
  }
  for (int i = 1; i < kNAgg; ++i) {
    if ((h->ablate_x >> (kAblAgg + i)) & 1u) HIP_TRY(h, zero_lo_slots(st, ws.vol[(i - 1) & 1], m * Dl, (size_t)hl * wl));
    SlotIn lv{U4(ws.vol[(i - 1) & 1]), Dl, hl, wl};
    if (i + 1 < kNAgg)
      HIP_TRY(h, (launch_conv_x3s<3, 1, 96, 8, 16, 16, 1, true, SlotIn>(st, h->agg[i], lv, m * Dl, hl, wl, ws.vol[i & 1], nullptr, true, ncu)));
    else
      HIP_TRY(h, (launch_conv_x3s<3, 1, 96, 8, 16, 16, 1, false, SlotIn>(st, h->agg[i], lv, m * Dl, hl, wl, ws.vol[i & 1], nullptr, true, ncu)));
  }
  }
  const float* v = ws.vol[(kNAgg - 1) & 1];
  const int npix = m * hl * wl;
  hipLaunchKernelGGL(k_head_softargmin<16>, dim3((npix + 63) / 64), dim3(64 * kSamWaves), 0, st, v, h->aout.w, h->aout.bias, Dl,
                     hl, wl, npix, ws.disp_low + (size_t)p0 * hl * wl,
                     want_cost ? ws.cost + (size_t)p0 * Dl * hl * wl : nullptr);
  HIP_TRY(h, hipGetLastError());
  return SN_OK;
}

int lowres(sn_handle* h, Workspace& ws, hipStream_t st, int p0, int m, const int8_t* in6, bool want_cost, bool prof) {
  const int Hp = h->Hp, Wp = h->Wp, hl = h->hl, wl = h->wl, Dl = h->Dl;
  const size_t HW = (size_t)h->H * h->W;
  const int8_t* in = in6 + (size_t)p0 * 6 * HW;
  if (h->precision != SN_PREC_FP32) return lowres_slots(h, ws, st, p0, m, in6, want_cost, prof);
  // SN_PREC_FP32: every layer on the exact-fp32 MFMA, fp32 NCHW activations
  // --- Siamese feature tower: images = 2m (left, right interleaved), shared weights ---
  {
    LoadI8Eye ld{in, h->H, h->W};
    const int Ho = Hp / 2, Wo = Wp / 2;
    if (Ho * Wo <= 64 * 128)
      HIP_TRY(h, (launch_conv<5, 2, 1, 4, 4, 32>(st, h->down[0], ld, 2 * m, Ho, Wo, ws.down[0], nullptr, false)));
    else
      HIP_TRY(h, (launch_conv<5, 2, 1, 4, 8, 64>(st, h->down[0], ld, 2 * m, Ho, Wo, ws.down[0], nullptr, false)));
  }
  HIP_TRY(h, conv5x5s2(st, h->down[1], ws.down[0], 2 * m, Hp / 2, Wp / 2, ws.down[1]));
  HIP_TRY(h, conv5x5s2(st, h->down[2], ws.down[1], 2 * m, Hp / 4, Wp / 4, ws.down[2]));
  HIP_TRY(h, conv5x5s2(st, h->down[3], ws.down[2], 2 * m, Hp / 8, Wp / 8, ws.low[0]));
  float* x = ws.low[0];
  float* t = ws.low[1];
  for (int i = 0; i < kNFeatRes; ++i) {
    HIP_TRY(h, conv3x3(st, h->fres[i][0], x, 2 * m, hl, wl, 1, t, nullptr, true));
    HIP_TRY(h, conv3x3(st, h->fres[i][1], t, 2 * m, hl, wl, 1, x, x, true));   // in-place residual
  }
  HIP_TRY(h, conv3x3(st, h->fout, x, 2 * m, hl, wl, 1, ws.feat, nullptr, false));
  if (prof) HIP_TRY(h, hipEventRecord(h->ev[1], st));

  // --- cost volume (fused into the first 3-D conv's loader) + 3-D aggregation + soft-argmin ---
  LoadCostVol ld{ws.feat, Dl, hl, wl};
  HIP_TRY(h, (launch_conv<3, 1, 1, 8, 4, 32>(st, h->agg[0], ld, m * Dl, hl, wl, ws.vol[0], nullptr, true)));
  for (int i = 1; i < kNAgg; ++i) {
    LoadVol3D lv{ws.vol[(i - 1) & 1], Dl, hl, wl};
    HIP_TRY(h, (launch_conv<3, 1, 1, 8, 4, 32>(st, h->agg[i], lv, m * Dl, hl, wl, ws.vol[i & 1], nullptr, true)));
  }
  const float* v = ws.vol[(kNAgg - 1) & 1];
  const int npix = m * hl * wl;
  hipLaunchKernelGGL(k_head_softargmin<16>, dim3((npix + 63) / 64), dim3(64 * kSamWaves), 0, st, v, h->aout.w, h->aout.bias, Dl,
                     hl, wl, npix, ws.disp_low + (size_t)p0 * hl * wl,
                     want_cost ? ws.cost + (size_t)p0 * Dl * hl * wl : nullptr);
  HIP_TRY(h, hipGetLastError());
  return SN_OK;
}

// Pieces of one forward(): [p0, p0 + m), ws.pb pairs each.  Rounds 1-4 started with a short piece (2-4 pairs: the towers
// can only start when the first piece's low-resolution branch is done).  With the round-5 low-resolution branch a whole
// first piece measures faster in the fp16 modes (fewer, fuller launches of kernels that are mostly fixed cost: 3041-3049 ->
// 3064-3068 pairs/s at 1280x720, 4070 -> 4121 at 1242x375, profiles/r05_schedule_sweep.txt) — the device is never idle
// either way, so what counts is the sum of the kernel times.  SN_PREC_FP32 keeps the short first piece: its low-resolution
// branch (generic fp32 kernel) is five times longer.  SN_FIRST_PIECE=n forces n pairs.
inline int first_piece(const sn_handle* h, const Workspace& ws, int n) {
  static const int forced = getenv("SN_FIRST_PIECE") ? atoi(getenv("SN_FIRST_PIECE")) : 0;     // experiment switch
  int m = ws.pb;
  if (h->precision == SN_PREC_FP32) m = ws.rb * ws.ns > 2 ? ws.rb * ws.ns : 2;
  if (forced > 0) m = forced;
  if (m > ws.pb) m = ws.pb;
  return m < n ? m : n;
}

// One refinement level of one tower chunk (c pairs) on stream `st`.
//   T          the level's tower (weights + geometry); rx / rt (fp32) or rx16 / rt16 (fp16 modes) its activation pair
//   src        [c][sh][sw] map the level starts from, upsampled by `ups` (x16: soft-argmin map, single-scale; x2: the
//              level below, hierarchical)
//   img_src    int8 model input of the chunk (pyr = false) or the level's float image pyramid [c][3][Hk][Wk]
//   H, W       size of the level's output map (the image for level 0, the whole padded level otherwise)
//   dnorm      D / 2^level: disparity normalisation at the tower input and residual scale at its output
//   od / orw   float map and (level 0 only) wire map, both nullable
//   cap        pairs the hi region of the activation buffers holds: the lo tensor of SN_PREC_F16X3 starts behind it (c <= cap)
//   mode       SN_PREC_F16 / SN_PREC_F16X3 / SN_PREC_FP32: the arithmetic of this call (an SN_PREC_AUTO handle holds two)
//   stat       the level's refinement statistic (sum of |D r|, refine_stat_commit)
int refine_level(sn_handle* h, Workspace& ws, hipStream_t st, const Tower& T, float* rx, float* rt, uint4* rx16,
                 uint4* rt16, const float* src, int sh, int sw, UpScale ups, const void* img_src, bool pyr, int H, int W,
                 float dnorm, float* od, int32_t* orw, unsigned* chunk_ctr, int c, int cap, bool pe, int mode,
                 unsigned long long* stat) {
  const int ncu = h->num_cu;
  const int tcu = ws.tower_cu > 0 ? ws.tower_cu : ncu;      // workgroups of the streamed tower launches
  const int Hk = T.Hk, Wk = T.Wk;
  // The wire factor is the reference's literal 16 * 12 for EVERY dmax (parser.cpp:86, stereonet_node.cpp:288,
  // publisher_member_function.py:75): the unmodified consumers recover pixels whatever D the model was built for.
  const float inv_q = (float)(1.0 / (kWireFactor * (double)kOutScale));
  if (mode == SN_PREC_FP32) {
    LoadRefineIn ld{src, reinterpret_cast<const int8_t*>(img_src), sh, sw, H, W, Hk, Wk, 1.0f / dnorm, ups,
                    pyr ? reinterpret_cast<const float*>(img_src) : nullptr};
    if (Hk * Wk <= 64 * 128)
      HIP_TRY(h, (launch_conv<3, 1, 1, 4, 4, 32>(st, T.rin, ld, c, Hk, Wk, rx, nullptr, true)));
    else
      HIP_TRY(h, (launch_conv<3, 1, 1, 4, 8, 64>(st, T.rin, ld, c, Hk, Wk, rx, nullptr, true)));
    if (pe) HIP_TRY(h, hipEventRecord(h->ev[4], st));
    for (int i = 0; i < kNRefRes; ++i) {
      HIP_TRY(h, conv3x3(st, T.rres[i][0], rx, c, Hk, Wk, kRefDil[i], rt, nullptr, true, ncu));
      HIP_TRY(h, conv3x3(st, T.rres[i][1], rt, c, Hk, Wk, kRefDil[i], rx, rx, true, ncu));
    }
    if (pe) HIP_TRY(h, hipEventRecord(h->ev[5], st));
    // head on the fp32 MFMA with the nine taps as M (k_head_final_mfma32); SN_HEAD_MFMA32=0 keeps the per-pixel kernel (A/B)
    static const bool head_mfma = !(getenv("SN_HEAD_MFMA32") != nullptr && atoi(getenv("SN_HEAD_MFMA32")) == 0);
    if (head_mfma) {
      constexpr int TH = 14;               // 16-row P window: 32 segments, 8 per wave (TH = 6 measured the same 41 us without the statistic)
      using HT = HeadTile<TH>;
      const int tiles_x = (W + HT::TWO - 1) / HT::TWO, tiles_y = (H + TH - 1) / TH;
      hipLaunchKernelGGL(k_head_final_mfma32<TH>, dim3((unsigned)(tiles_x * tiles_y * c)), dim3(256), HT::LDS_BYTES, st, rx,
                         T.rout.w, T.rout.bias, src, sh, sw, Hk, Wk, H, W, dnorm, inv_q, od, orw, tiles_x, tiles_y, ups, stat);
    } else {
      dim3 grid((W + 63) / 64, (H + 3) / 4, c);
      hipLaunchKernelGGL(k_head_final, grid, dim3(256), 0, st, rx, T.rout.w, T.rout.bias, src, sh, sw, Hk, Wk, H, W, dnorm,
                         inv_q, od, orw, ups, stat);
    }
  } else {
    // fp16 tower: ref.in writes the NCHW8c fp16 tensor, the 12 C->C convs run on v_mfma_f32_32x32x16_f16, the head
    // reads fp16 and finishes in fp32
    uint4* x16 = rx16;
    uint4* t16 = rt16;
    if (pe) h->dom_pairs = 0;
    // Consecutive launches of a tower walk their tiles in OPPOSITE directions (g.rev): a launch then starts on the part
    // of the tensor its predecessor wrote LAST — what a cache that is slightly too small for the chunk still holds —
    // instead of on the lines an LRU policy has just evicted.
    // Neutral while the chunk fits the Infinity Cache (1280x720, two pairs: 2304 vs 2290 pairs/s), +11 % when it does
    // not (three pairs: 82 instead of 95 us per launch; any geometry whose single pair exceeds the cache).  SN_REV=0
    // disables it (diagnostic).
    static const int rev_env = getenv("SN_REV") ? atoi(getenv("SN_REV")) : 1;
    RefGeom g = T.rg;
    int launch_no = 0;
    auto flip = [&]() { g.rev = rev_env ? (launch_no++ & 1) : 0; };
    flip();
    const bool x3 = mode == SN_PREC_F16X3;
    const size_t lo_slots = ref16_slots(g, cap) + ref_slack(g);         // hi tensor -> lo tensor (F16X3); cap = pairs the buffers hold
    HIP_TRY(h, launch_refin_f16(st, T.refin, T.rin.bias, src, img_src, pyr, sh, sw, H, W, 1.0f / dnorm, ups, g, c, x16, x3,
                                lo_slots * 16, ncu));
    if (pe) HIP_TRY(h, hipEventRecord(h->ev[4], st));
    flip();
    // tail form: the streamed last block computes the head too (its output tensor is never written, no head launch)
    const bool last_streamed = h->fuse_mode == 4 && stream_block_supports(kRefDil[kNRefRes - 1]);
    const bool tail = !x3 && last_streamed && h->tail_fuse && kRefDil[kNRefRes - 1] == 1 && ups.rs <= 0.5f;
    for (int i = 0; i < kNRefRes; ++i) {
      if (x3) {
        const bool dom = pe && stream_x3_supports(kRefDil[i]) && h->dom_pairs < 6;
        if (dom) HIP_TRY(h, hipEventRecord(h->ev_dom[2 * h->dom_pairs], st));
        HIP_TRY(h, ref_block_f16x3(st, T.rres16x3[i][0], T.rres16x3[i][1], g, tcu, kRefDil[i], &x16, &t16, lo_slots, c));
        if (dom) HIP_TRY(h, hipEventRecord(h->ev_dom[2 * h->dom_pairs++ + 1], st));
      } else if (tail && i == kNRefRes - 1) {
        if (pe) HIP_TRY(h, hipEventRecord(h->ev[5], st));          // the plain tower launches end here
        StreamHeadArgs ha{T.rout.w, src, od, orw, T.rout.bias, dnorm, inv_q, sh, sw, H, W, ups, stat};
        HIP_TRY(h, ref_block_stream_tail(st, T.rres16[i][0], T.rres16[i][1], g, tcu, x16, c, h->dump, ha));
      } else {
        const bool dom = pe && h->fuse_mode == 4 && stream_block_supports(kRefDil[i]) && h->dom_pairs < 6;
        if (dom) HIP_TRY(h, hipEventRecord(h->ev_dom[2 * h->dom_pairs], st));
        HIP_TRY(h, ref_block_f16(st, T.rres16[i][0], T.rres16[i][1], g, tcu, kRefDil[i], &x16, &t16, c,
                                 chunk_ctr + 2 * i * kTileCtrStride, h->fuse_mode, h->dump, rev_env != 0));
        if (dom) HIP_TRY(h, hipEventRecord(h->ev_dom[2 * h->dom_pairs++ + 1], st));
      }
    }
    if (!tail) {
      if (pe) HIP_TRY(h, hipEventRecord(h->ev[5], st));
      HIP_TRY(h, launch_head_final_f16(st, x3, x16, lo_slots, g, T.rout.w, T.rout.bias, src, sh, sw, H, W, dnorm, inv_q, ups,
                                       od, orw, c, stat));
    }
  }
  HIP_TRY(h, hipGetLastError());
  return SN_OK;
}

// pairs per tower launch of level lv for a call in `mode` (an SN_PREC_AUTO handle in SN_PREC_F16X3 packs fewer pairs into the
// same buffers)
inline int chunk_pairs(const sn_handle* h, const Workspace& ws, int mode, int lv = 0) {
  return (h->precision == SN_PREC_AUTO && mode == SN_PREC_F16X3) ? ws.rbk_x3[lv] : ws.rbk[lv];
}

// Next tile-queue block of this forward() (nullptr for the fp32 path, which has no queues); the pool is sized by
// alloc_ws for the worst case, running past it would alias another launch's counters -> refuse loudly.
inline int take_ctr_block(sn_handle* h, Workspace& ws, int* ctr_block, unsigned** out) {
  *out = nullptr;
  if (!ws.tile_ctr) return SN_OK;
  if (*ctr_block >= ws.n_chunks) {
    set_err(h, "internal: tile-queue pool exhausted");
    return SN_ERR_DEVICE;
  }
  *out = ws.tile_ctr + (size_t)(*ctr_block)++ * (kTileCtrBytes / sizeof(unsigned));
  return SN_OK;
}

// Hierarchical model (SURVEY.md appendix A `multi`), coarse part, once per low-resolution piece [p0, p0+m): the image
// pyramid of the left eye, then the towers of levels levels-1 .. 1, each starting from the x2 upsample of the map below
// it (the soft-argmin map for the coarsest), values x2, normalised by D / 2^level.  Level k runs in chunks of
// ws.rbk[k] pairs.  Leaves the level-1 maps of the piece in ws.lvl_disp[1].  *ctr_block: next free tile-queue block.
int refine_coarse(sn_handle* h, Workspace& ws, hipStream_t st, int p0, int m, const int8_t* in6, int* ctr_block, int mode) {
  const size_t HW = (size_t)h->H * h->W;
  const int8_t* in_piece = in6 + (size_t)p0 * 6 * HW;
  for (int lv = 1; lv < h->levels; ++lv) {       // level 1 from the int8 input, the others from the level above
    const Tower& T = h->tw[lv];
    const long total = (long)m * 3 * T.Hk * T.Wk;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (lv == 1)
      hipLaunchKernelGGL(k_img_pool2<true>, grid, dim3(256), 0, st, (const void*)in_piece, h->H, h->W, T.Hk, T.Wk,
                         ws.pyr[lv], total);
    else
      hipLaunchKernelGGL(k_img_pool2<false>, grid, dim3(256), 0, st, (const void*)ws.pyr[lv - 1], 0, 0, T.Hk, T.Wk,
                         ws.pyr[lv], total);
  }
  HIP_TRY(h, hipGetLastError());
  const float* src = ws.disp_low + (size_t)p0 * h->hl * h->wl;
  int sh = h->hl, sw = h->wl;
  for (int lv = h->levels - 1; lv >= 1; --lv) {
    const Tower& T = h->tw[lv];
    const size_t HWk = (size_t)T.Hk * T.Wk;
    const float dnorm = (float)h->D / (float)(1 << lv);
    const int rbk = chunk_pairs(h, ws, mode, lv);
    for (int q = 0; q < m; q += rbk) {
      const int c = (m - q) < rbk ? (m - q) : rbk;
      unsigned* ctr = nullptr;
      int rc = take_ctr_block(h, ws, ctr_block, &ctr);
      if (rc) return rc;
      rc = refine_level(h, ws, st, T, ws.ref_lv[lv][0], ws.ref_lv[lv][1], ws.ref16_lv[lv][0], ws.ref16_lv[lv][1],
                                  src + (size_t)q * sh * sw, sh, sw, UpScale{0.5f, 2.0f}, ws.pyr[lv] + (size_t)q * 3 * HWk, true,
                                  T.Hk, T.Wk, dnorm, ws.lvl_disp[lv] + (size_t)q * HWk, nullptr, ctr, c, ws.rbk[lv], false, mode,
                                  ws.stats + (size_t)lv * kStatWordStride);
      if (rc) return rc;
    }
    src = ws.lvl_disp[lv];
    sh = T.Hk;
    sw = T.Wk;
  }
  return SN_OK;
}

// Full-resolution refinement of ONE tower chunk: pairs [q0, q0+c), c <= ws.rb, of the piece that starts at p0, on
// stream `st` with the activation pair of tower stream `sidx`; *ctr_block: next free tile-queue block.
//   single-scale model: x16 upsample of the soft-argmin map;
//   hierarchical model: x2 upsample of the piece's level-1 maps (refine_coarse ran before on the same stream).
int refine_chunk(sn_handle* h, Workspace& ws, hipStream_t st, int sidx, int* ctr_block, int p0, int q0, int c,
                 const int8_t* in6, float* out_disp, int32_t* out_raw, bool pe, int mode) {
  const int hl = h->hl, wl = h->wl;
  const size_t HW = (size_t)h->H * h->W;
  float* od = out_disp ? out_disp + (size_t)q0 * HW : nullptr;
  int32_t* orw = out_raw ? out_raw + (size_t)q0 * HW : nullptr;
  const int8_t* in_chunk = in6 + (size_t)q0 * 6 * HW;
  unsigned* ctr = nullptr;
  const int rc0 = take_ctr_block(h, ws, ctr_block, &ctr);
  if (rc0) return rc0;
  const float* src = ws.disp_low + (size_t)q0 * hl * wl;
  int sh = hl, sw = wl;
  UpScale ups{1.0f / 16.0f, 16.0f};
  if (h->levels > 1) {
    sh = h->tw[1].Hk;
    sw = h->tw[1].Wk;
    src = ws.lvl_disp[1] + (size_t)(q0 - p0) * sh * sw;
    ups = UpScale{0.5f, 2.0f};
  }
  return refine_level(h, ws, st, h->tw[0], ws.ref[2 * sidx], ws.ref[2 * sidx + 1], ws.ref16[2 * sidx], ws.ref16[2 * sidx + 1],
                      src, sh, sw, ups, in_chunk, false, h->H, h->W, (float)h->D, od, orw, ctr, c, ws.rb, pe, mode, ws.stats);
}

// in6: device int8 [n][6][H][W]; out_disp / out_raw: device, nullable.
// The batch is cut into pieces of ws.pb pairs and every piece into tower chunks of ws.rb pairs.  More than one chunk:
// three streams forked from / joined back into the caller's stream with events (plain stream semantics for the caller):
//   s_low      the low-resolution branch of piece k+1 (matrix-pipe bound, little HBM traffic) runs under
//   s_tow[0/1] the refinement towers of piece k (HBM bound); consecutive chunks ALTERNATE between the two tower
//              streams.  A tower launch costs bytes / 7.5 TB/s plus ~14 us that do not depend on its size (kernel
//              boundary, weight / first-tile prologue, and a tail in which the last tiles of the persistent grid
//              finish one by one); with two independent chunks in flight the workgroups of chunk B's launch take over
//              the CUs that chunk A's launch drains, and A's next launch (which depends only on A) is ready by the
//              time B drains.  Two one-pair chunks in flight = four 61 MB tensors = the footprint of one two-pair
//              chunk, still inside the 256 MB Infinity Cache.
// mode: the arithmetic of this call (SN_PREC_F16 / F16X3 / FP32; 0 = the handle's current one).  Ends with the copy of the
// refinement statistic to the workspace's pinned twin, in stream order.
int forward(sn_handle* h, Workspace& ws, hipStream_t st, int n, const int8_t* in6, float* out_disp,
            int32_t* out_raw, bool want_cost, int mode = 0) {
  if (mode == 0) mode = h->precision == SN_PREC_AUTO ? h->actl.st.mode : h->precision;
  const int rb = chunk_pairs(h, ws, mode);
  const bool prof = h->profiling && (&ws == &h->ws);
  const bool piped = !prof && (&ws == &h->ws) && h->overlap && n > rb;
  int rc;
  int ctr_block = 0;          // tile-queue blocks are handed out in launch order (alloc_ws sized the pool)
  // the tile queues belong to the per-layer fp16 kernel (k_ref_conv_f16_v2): with every block of this call streamed (the
  // default) or on split operands nobody reads them, and the fill is a 4 us launch of its own in front of a single pair
  bool need_queues = ws.tile_ctr != nullptr && mode == SN_PREC_F16;
  if (need_queues && h->fuse_mode == 4) {
    need_queues = false;
    for (int i = 0; i < kNRefRes; ++i) need_queues = need_queues || !stream_block_supports(kRefDil[i]);
  }
  if (need_queues) HIP_TRY(h, hipMemsetAsync(ws.tile_ctr, 0, kTileCtrBytes * ws.n_chunks, st));
  HIP_TRY(h, hipMemsetAsync(ws.stats, 0, kMaxLevels * kStatWordStride * sizeof(unsigned long long), st));
  auto finish = [&]() -> int {
    HIP_TRY(h, hipMemcpyAsync(ws.stats_host, ws.stats, (size_t)h->levels * kStatWordStride * sizeof(unsigned long long),
                              hipMemcpyDeviceToHost, st));        // the levels this model has (2 KB each)
    return SN_OK;
  };
  if (!piped) {
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[0], st));
    for (int p0 = 0, m = 0; p0 < n; p0 += m) {
      m = (n - p0) < ws.pb ? (n - p0) : ws.pb;
      if ((rc = lowres(h, ws, st, p0, m, in6, want_cost, prof && p0 == 0))) return rc;
      if (prof && p0 == 0) HIP_TRY(h, hipEventRecord(h->ev[2], st));
      if (h->levels > 1 && (rc = refine_coarse(h, ws, st, p0, m, in6, &ctr_block, mode))) return rc;
      for (int q0 = p0; q0 < p0 + m; q0 += rb) {
        const int c = (p0 + m - q0) < rb ? (p0 + m - q0) : rb;
        if ((rc = refine_chunk(h, ws, st, 0, &ctr_block, p0, q0, c, in6, out_disp, out_raw, prof && q0 == 0, mode))) return rc;
      }
    }
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[3], st));
    return finish();
  }
  const int ns = ws.ns;
  HIP_TRY(h, hipEventRecord(h->ev_fork, st));
  HIP_TRY(h, hipStreamWaitEvent(h->s_low, h->ev_fork, 0));
  for (int s = 0; s < ns; ++s) HIP_TRY(h, hipStreamWaitEvent(h->s_tow[s], h->ev_fork, 0));
  int k = 0, chunk = 0;
  for (int p0 = 0, m = 0; p0 < n; p0 += m, ++k) {
    m = p0 == 0 ? first_piece(h, ws, n) : ((n - p0) < ws.pb ? (n - p0) : ws.pb);
    // the piece-local low-res buffers are reused by the next piece: only disp_low crosses streams
    if ((rc = lowres(h, ws, h->s_low, p0, m, in6, want_cost, false))) return rc;
    hipEvent_t e = h->ev_piece[k % kMaxPieceEvents];
    HIP_TRY(h, hipEventRecord(e, h->s_low));
    bool waited[kMaxTowerStreams] = {};
    if (h->levels > 1) {                   // coarse levels of the whole piece first (one tower stream: alloc_ws)
      HIP_TRY(h, hipStreamWaitEvent(h->s_tow[0], e, 0));
      waited[0] = true;
      if ((rc = refine_coarse(h, ws, h->s_tow[0], p0, m, in6, &ctr_block, mode))) return rc;
    }
    for (int q0 = p0; q0 < p0 + m; q0 += rb, ++chunk) {
      const int c = (p0 + m - q0) < rb ? (p0 + m - q0) : rb;
      const int s = chunk % ns;
      if (!waited[s]) {
        HIP_TRY(h, hipStreamWaitEvent(h->s_tow[s], e, 0));
        waited[s] = true;
      }
      if ((rc = refine_chunk(h, ws, h->s_tow[s], s, &ctr_block, p0, q0, c, in6, out_disp, out_raw, false, mode))) return rc;
    }
  }
  HIP_TRY(h, hipEventRecord(h->ev_join, h->s_low));
  HIP_TRY(h, hipStreamWaitEvent(st, h->ev_join, 0));
  for (int s = 0; s < ns; ++s) {
    HIP_TRY(h, hipEventRecord(h->ev_tow_join[s], h->s_tow[s]));
    HIP_TRY(h, hipStreamWaitEvent(st, h->ev_tow_join[s], 0));
  }
  return finish();
}

int collect_profile(sn_handle* h) {
  if (!h->profiling) return SN_OK;
  HIP_TRY(h, hipEventSynchronize(h->ev[3]));
  float ms = 0.f;
  HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  h->stage_ms[SN_STAGE_FEATURES] = ms;
  HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[1], h->ev[2]));
  h->stage_ms[SN_STAGE_AGGREGATE] = ms;
  HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[2], h->ev[3]));
  h->stage_ms[SN_STAGE_REFINE] = ms;
  HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[4], h->ev[5]));
  h->stage_ms[SN_STAGE_REFINE_CONV] = ms;    // first refinement chunk only
  HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[0], h->ev[3]));
  h->stage_ms[SN_STAGE_TOTAL] = ms;
  // the dominant kernel's launches of the first chunk: the streamed blocks one by one, else the tower span
  if (h->dom_pairs > 0) {
    float sum = 0.f;
    for (int i = 0; i < h->dom_pairs; ++i) {
      HIP_TRY(h, hipEventElapsedTime(&ms, h->ev_dom[2 * i], h->ev_dom[2 * i + 1]));
      sum += ms;
    }
    h->stage_ms[SN_STAGE_DOMINANT] = sum;
  } else {
    h->stage_ms[SN_STAGE_DOMINANT] = h->stage_ms[SN_STAGE_REFINE_CONV];
  }
  return SN_OK;
}

int check_device(sn_handle* h) {
  HIP_TRY(h, hipSetDevice(h->device));
  return SN_OK;
}

// ---- refinement statistic and SN_PREC_AUTO (include/stereonet_hip.h) ---------------------------------------------------
__global__ __launch_bounds__(256) void k_abs_diff_sum(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                      unsigned long long* __restrict__ out) {
  float sum = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) sum += fabsf(a[i] - b[i]);
  refine_stat_commit_block(out, sum);
}

// mean |D_k r_k| per level from a workspace's pinned statistic of an n-pair call (valid once the stream that ran forward()
// has been synchronised): level 0 writes the H x W output maps, a coarse level its whole padded map
inline unsigned long long stat_word(const Workspace& ws, int word) {      // the word's partial sums (refine_stat_commit)
  unsigned long long sum = 0;
  for (int s = 0; s < kStatSlots; ++s) sum += ws.stats_host[(size_t)word * kStatWordStride + (size_t)s * kStatLine];
  return sum;
}
void read_stats(const sn_handle* h, const Workspace& ws, int n, double* level_px, double* residual_px) {
  double res = 0.0;
  for (int lv = 0; lv < kMaxLevels; ++lv) {
    level_px[lv] = 0.0;
    if (lv >= h->levels || n <= 0) continue;
    const double px = lv == 0 ? (double)h->H * h->W : (double)h->tw[lv].Hk * h->tw[lv].Wk;
    level_px[lv] = (double)stat_word(ws, lv) / (double)kStatScale / (px * n);
    res += level_px[lv] * (double)(1 << lv);
  }
  *residual_px = res;
}

// every call is counted when it is issued (the statistic of an enqueue-only call may be superseded by the next call's before
// anybody looks at it; the count may not)
void count_call(sn_handle* h, int n) {
  std::lock_guard<std::mutex> lk(h->mu);
  ++h->actl.calls;
  h->actl.pairs += (uint64_t)n;
}

// Folds the statistic of one finished call (run in `mode`) into the handle; returns the arithmetic the handle is in
// afterwards.  observe = false: a repeated call (its first run has been observed already).
int fold_stats(sn_handle* h, const double* level_px, double residual_px, int n, int mode, bool observe = true) {
  std::lock_guard<std::mutex> lk(h->mu);
  AutoCtl& a = h->actl;
  for (int lv = 0; lv < kMaxLevels; ++lv) a.last_level[lv] = level_px[lv];
  a.last_res = residual_px;
  a.last_mode = mode;
  if (!observe) {
    ++a.reruns;
    return a.st.mode;
  }
  if (h->precision != SN_PREC_AUTO) {
    a.st.running_px = a.st.running_px < 0.0 ? residual_px : 0.75 * a.st.running_px + 0.25 * residual_px;
    return h->precision;
  }
  const int before = a.st.mode;
  const int after = sn_auto_observe(&a.st, residual_px);
  if (before == SN_PREC_F16X3 && after == SN_PREC_F16) a.calibrated = false;      // re-entering F16: check it again
  return after;
}

// SN_PREC_AUTO's self-check: ONE pair in both arithmetics (the low-resolution branch is the same code, so the maps differ by
// the towers' arithmetic alone), mean |F16 - F16X3| against the pair's residual -> the handle's measured EPE per pixel of
// residual.  Runs on `st` with the workspace of the call that triggers it and returns after synchronising.
int auto_selfcheck(sn_handle* h, Workspace& ws, hipStream_t st, const int8_t* in6_pair) {
  std::lock_guard<std::mutex> cal(h->mu_cal);
  {
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->actl.calibrated) return SN_OK;
  }
  const bool prof = h->profiling;
  h->profiling = false;                  // the stage events belong to the caller's own forward()
  int rc = forward(h, ws, st, 1, in6_pair, h->chk[1], nullptr, false, SN_PREC_F16X3);
  if (!rc) rc = forward(h, ws, st, 1, in6_pair, h->chk[0], nullptr, false, SN_PREC_F16);    // last: the intermediates sn_dbg_read sees
  h->profiling = prof;
  if (rc) return rc;
  const size_t HW = (size_t)h->H * h->W;
  HIP_TRY(h, hipMemsetAsync(ws.stats + 4 * kStatWordStride, 0, kStatWordStride * sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_abs_diff_sum, dim3(512), dim3(256), 0, st, h->chk[0], h->chk[1], HW, ws.stats + 4 * kStatWordStride);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(ws.stats_host, ws.stats, kStatU64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipStreamSynchronize(st));
  double lvl[kMaxLevels], res = 0.0;
  read_stats(h, ws, 1, lvl, &res);
  const double epe = (double)stat_word(ws, 4) / (double)kStatScale / (double)HW;
  std::lock_guard<std::mutex> lk(h->mu);
  AutoCtl& a = h->actl;
  a.selfcheck_epe = epe;
  a.selfcheck_res = res;
  // (a model whose refinement adds nothing has nothing to lose in fp16: keep the envelope alone)
  a.st.epe_per_px = res > 1e-6 ? epe / res : 0.0;
  a.calibrated = true;
  return SN_OK;
}

// A statistic of an earlier call that only enqueued its work (device buffers + caller stream): folded in once its copy has
// landed (wait = false: only if it already has).
int fold_pending(sn_handle* h, bool wait) {
  AutoCtl& a = h->actl;
  if (!a.pending) return SN_OK;
  if (wait) {
    HIP_TRY(h, hipEventSynchronize(h->ev_stats));
  } else if (hipEventQuery(h->ev_stats) != hipSuccess) {
    (void)hipGetLastError();               // hipErrorNotReady: try again at the next call
    return SN_OK;
  }
  a.pending = false;
  double lvl[kMaxLevels], res = 0.0;
  read_stats(h, h->ws, a.pending_n, lvl, &res);
  fold_stats(h, lvl, res, a.pending_n, a.pending_mode);
  return SN_OK;
}

// forward() on the handle's own workspace for the synchronous entry points.  post() enqueues what follows the network
// (device-to-host copies).  blocking: the entry point returns after completion — the statistic is folded in before it
// does and, under SN_PREC_AUTO, a call that left the fp16 tower's envelope is REPEATED in SN_PREC_F16X3.  Not blocking
// (work only enqueued on the caller's stream): the statistic is folded in by a later call; an AUTO handle that has not had
// its self-check yet blocks once.
template <class Post>
int run_forward(sn_handle* h, hipStream_t st, int n, const int8_t* din, float* ddisp, int32_t* draw, bool want_cost,
                bool blocking, Post post) {
  const bool is_auto = h->precision == SN_PREC_AUTO;
  AutoCtl& a = h->actl;
  int rc = fold_pending(h, false);
  if (rc) return rc;
  int mode, calibrated;
  {
    std::lock_guard<std::mutex> lk(h->mu);
    mode = is_auto ? a.st.mode : h->precision;
    calibrated = a.calibrated;
  }
  const bool check = is_auto && mode == SN_PREC_F16 && !calibrated;
  if ((rc = forward(h, h->ws, st, n, din, ddisp, draw, want_cost, mode))) return rc;
  if ((rc = post())) return rc;
  count_call(h, n);
  if (!blocking && !check) {
    HIP_TRY(h, hipEventRecord(h->ev_stats, st));
    a.pending = true;
    a.pending_n = n;
    a.pending_mode = mode;
    return SN_OK;
  }
  HIP_TRY(h, hipStreamSynchronize(st));
  double lvl[kMaxLevels], res = 0.0;
  read_stats(h, h->ws, n, lvl, &res);
  if (check && (rc = auto_selfcheck(h, h->ws, st, din))) return rc;
  const int next = fold_stats(h, lvl, res, n, mode);
  if (is_auto && mode == SN_PREC_F16 && next == SN_PREC_F16X3) {
    if ((rc = forward(h, h->ws, st, n, din, ddisp, draw, want_cost, SN_PREC_F16X3))) return rc;
    if ((rc = post())) return rc;
    HIP_TRY(h, hipStreamSynchronize(st));
    read_stats(h, h->ws, n, lvl, &res);
    fold_stats(h, lvl, res, n, SN_PREC_F16X3, false);
  }
  return collect_profile(h);
}

// host <-> split-slot layout (SlotIn): src/dst fp32 [nimg][32][H][W]
void host_to_slots(const float* src, int nimg, int H, int W, std::vector<_Float16>& dst) {
  const size_t plane = (size_t)H * W;
  dst.assign((size_t)nimg * 8 * plane * 8, (_Float16)0.f);
  for (int img = 0; img < nimg; ++img)
    for (int c = 0; c < kC; ++c)
      for (size_t i = 0; i < plane; ++i) {
        const float v = src[((size_t)img * kC + c) * plane + i];
        const _Float16 hi = (_Float16)v;
        const size_t base = (((size_t)img * 4 + (c >> 3)) * 2) * plane;
        dst[(base + i) * 8 + (c & 7)] = hi;
        dst[(base + plane + i) * 8 + (c & 7)] = (_Float16)((v - (float)hi) * kSplitScale);
      }
}
void host_from_slots(const std::vector<_Float16>& src, int nimg, int H, int W, float* dst) {
  const size_t plane = (size_t)H * W;
  for (int img = 0; img < nimg; ++img)
    for (int c = 0; c < kC; ++c)
      for (size_t i = 0; i < plane; ++i) {
        const size_t base = (((size_t)img * 4 + (c >> 3)) * 2) * plane;
        dst[((size_t)img * kC + c) * plane + i] =
            (float)src[(base + i) * 8 + (c & 7)] + (float)src[(base + plane + i) * 8 + (c & 7)] * kSplitInv;
      }
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

const char* sn_strerror(int code) {
  switch (code) {
    case SN_OK: return "ok";
    case SN_ERR_ARG: return "invalid argument";
    case SN_ERR_FILE: return "model file missing or unreadable";
    case SN_ERR_FORMAT: return "model file is not an SN-K4 SNW1 weight file";
    case SN_ERR_DEVICE: return "HIP device error (a gfx950 GPU is required; there is no CPU path)";
    case SN_ERR_NOMEM: return "out of memory";
    case SN_ERR_BUSY: return "no free task slot";
    case SN_ERR_TICKET: return "unknown ticket";
    default: return "unknown error";
  }
}

// detail of the last failed sn_create on this thread (there is no handle to carry it)
static thread_local std::string g_create_err;
static int create_fail(int code, const char* what) {
  const hipError_t e = hipGetLastError();
  g_create_err = std::string(what) + (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string());
  return code;
}

const char* sn_last_error(const sn_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int sn_create(const char* model_file, const sn_config* cfg, sn_handle** out) {
  return sn_create_prio(model_file, cfg, -1, out);
}

// stream_prio: 1 = pipeline streams at the device's highest priority, 0 = default priority, -1 = SN_STREAM_PRIORITY decides
// (unset: default).  An explicit SN_STREAM_PRIORITY always wins, so the A/B switch stays usable for every caller.
int sn_create_prio(const char* model_file, const sn_config* cfg, int stream_prio, sn_handle** out) {
  if (!model_file || !out) return SN_ERR_ARG;
  *out = nullptr;
  FILE* f = fopen(model_file, "rb");
  if (!f) return SN_ERR_FILE;
  SnwHeader hd;
  if (fread(&hd, 1, sizeof hd, f) != sizeof hd || memcmp(hd.magic, "SNW1", 4) != 0) {
    fclose(f);
    return SN_ERR_FORMAT;
  }
  const uint32_t dil_ok[6] = {1, 2, 4, 8, 1, 1};
  if (hd.version != 1 || hd.channels != kC || hd.n_down != kNDown || hd.n_fres != kNFeatRes ||
      hd.n_agg != kNAgg || hd.n_rres != kNRefRes || memcmp(hd.dil, dil_ok, sizeof dil_ok) != 0 ||
      (hd.reserved != 0 && hd.reserved != 1 && hd.reserved != (uint64_t)kMultiLevels)) {
    fclose(f);
    return SN_ERR_FORMAT;
  }
  // header word 72: refinement levels (0 / 1 = single-scale tower, 4 = hierarchical; weights.py documents the layout)
  const int levels = hd.reserved > 1 ? (int)hd.reserved : 1;
  if (hd.n_params != param_count(levels)) {
    fclose(f);
    return SN_ERR_FORMAT;
  }
  std::vector<float> blob(hd.n_params);
  const size_t got = fread(blob.data(), sizeof(float), blob.size(), f);
  fclose(f);
  if (got != blob.size()) return SN_ERR_FORMAT;

  sn_config c{};
  if (cfg) c = *cfg; else c.device = -1;
  const int W = c.width > 0 ? c.width : (int)hd.width;
  const int H = c.height > 0 ? c.height : (int)hd.height;
  const int D = c.dmax > 0 ? c.dmax : (int)hd.dmax;
  if (W <= 0 || H <= 0 || D < 16 || D % 16 || D > 256) return SN_ERR_ARG;   // NV12 entry points add w%4, h%2
  if (c.precision == SN_PREC_DEFAULT) {
    // SN_PRECISION=f16|f16x3|fp32|auto: what "default" means for this process (A/B runs of unmodified callers)
    const char* e = getenv("SN_PRECISION");
    c.precision = SN_PREC_AUTO;
    if (e && !strcmp(e, "f16")) c.precision = SN_PREC_F16;
    else if (e && !strcmp(e, "f16x3")) c.precision = SN_PREC_F16X3;
    else if (e && !strcmp(e, "fp32")) c.precision = SN_PREC_FP32;
  }
  if (c.precision != SN_PREC_FP32 && c.precision != SN_PREC_F16 && c.precision != SN_PREC_F16X3 && c.precision != SN_PREC_AUTO)
    return SN_ERR_ARG;

  int ndev = 0;
  g_create_err.clear();
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return create_fail(SN_ERR_DEVICE, "hipGetDeviceCount");
  int dev = c.device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return create_fail(SN_ERR_DEVICE, "hipGetDevice");
  if (dev >= ndev) return SN_ERR_ARG;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return create_fail(SN_ERR_DEVICE, "hipGetDeviceProperties");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fprintf(stderr, "stereonet_hip: device %d is %s, this library is built for gfx950 only\n", dev,
            prop.gcnArchName);
    return SN_ERR_DEVICE;
  }

  sn_handle* h = new sn_handle();
  h->device = dev;
  h->W = W;
  h->H = H;
  h->D = D;
  h->Wp = (W + 15) / 16 * 16;
  h->Hp = (H + 15) / 16 * 16;
  h->wl = h->Wp / 16;
  h->hl = h->Hp / 16;
  h->Dl = D / 16;
  h->max_batch = c.max_batch > 0 ? c.max_batch : 1;
  h->precision = c.precision;
  h->task_num = c.task_num > 0 ? c.task_num : 4;
  h->refine_chunk = c.refine_chunk;                            // <= 0: chosen below from the tensor size
  h->piece = c.piece > 0 ? c.piece : 16;
  h->levels = levels;
  sn_auto_init(&h->actl.st, levels);
  if (c.precision != SN_PREC_AUTO) h->actl.st.mode = c.precision;
  h->actl.last_mode = h->actl.st.mode;
  for (int k = 0; k < levels; ++k) {
    h->tw[k].Hk = h->Hp >> k;
    h->tw[k].Wk = h->Wp >> k;
    h->tw[k].rg = make_ref_geom(h->tw[k].Hk, h->tw[k].Wk);
  }
  {
    const char* e = getenv("SN_TOWER_STREAMS");        // 2 = consecutive tower chunks alternate between two streams
    // default 1; 2 in SN_PREC_FP32: a one-pair launch of the fp32 tower kernel is 3.5 rounds of tiles on the persistent
    // grid, and the next chunk's launch on the other stream takes the CUs the last half round leaves idle (+6 %)
    h->tower_streams = e ? atoi(e) : (h->precision == SN_PREC_FP32 ? 2 : 1);
    if (h->tower_streams < 1) h->tower_streams = 1;
    if (h->tower_streams > kMaxTowerStreams) h->tower_streams = kMaxTowerStreams;
  }
  const bool want_f16 = c.precision == SN_PREC_F16 || c.precision == SN_PREC_AUTO;
  const bool want_x3 = c.precision == SN_PREC_F16X3 || c.precision == SN_PREC_AUTO;
  {
    // the Infinity-Cache sizing of the per-layer forms, for the split tensors of SN_PREC_F16X3 (what an AUTO handle falls back to)
    const double tensor_mb = 4.0 * h->tw[0].rg.Hs * h->tw[0].rg.Ws * 16.0 / 1048576.0 * 2.0;
    int r = (int)(256.0 / (2.0 * tensor_mb * h->tower_streams) + 0.5);
    // with the split blocks streamed too (sn_stream_block_x3.hpp) the chunk no longer has to live in the Infinity Cache:
    // the fp16 rule below (1280x720: one pair 1067, four 1099-1134, six 1137-1139, eight 1142 pairs/s, profiles/r06_x3_stream_ab.txt)
    if (stream_x3_supports(8)) {
      const int by_px = (int)(5.5e6 / ((double)h->Hp * h->Wp) + 0.5);
      if (by_px > r) r = by_px;
    }
    h->refine_chunk_x3 = r < 1 ? 1 : (r > 8 ? 8 : r);
  }
  if (h->refine_chunk <= 0) {
    // Pairs per tower launch: as many as keep the activations in flight — (x, t) per tower stream — inside the
    // 256 MB Infinity Cache.  A launch costs bytes / ~7 TB/s while its tensors stay cache resident plus ~8 us that do
    // not depend on its size, and ~5 TB/s per byte once they spill (scripts/mall_probe.hip): 1280x720
    // -> 2 pairs (4 x 61 MB), 1248x384 -> 4 pairs (8 x 32 MB); measured equal to one-pair chunks alternating on two
    // streams (SN_TOWER_STREAMS=2), with fewer and fuller launches.
    const double tensor_mb = 4.0 * h->tw[0].rg.Hs * h->tw[0].rg.Ws * 16.0 / 1048576.0 * (c.precision == SN_PREC_F16X3 ? 2.0 : c.precision == SN_PREC_FP32 ? 2.0 : 1.0);
    int rc_auto = (int)(256.0 / (2.0 * tensor_mb * h->tower_streams) + 0.5);
    // With every residual block streamed (fp16 mode, SN_FUSE=4) a launch reads x and writes y ONCE while it does two
    // convolutions: ~2.8 TB/s at the rate the matrix pipes allow, which HBM sustains — the chunk no longer has to live in
    // the Infinity Cache, and fuller launches amortise the restart rows and the launch itself: ~5.5 Mpx per launch
    // (1280x720: 6 pairs; round 3: 4 pairs 2583 -> 2653 pairs/s; round 5, three interleaved runs: 4 pairs 3064-3068,
    // 6 pairs 3074-3081, 8 pairs 3071-3078, profiles/r05_schedule_sweep.txt).
    if ((want_f16 && fuse_env() == 4 && stream_block_supports(8)) || (c.precision == SN_PREC_F16X3 && stream_x3_supports(8))) {
      const int by_px = (int)(5.5e6 / ((double)h->Hp * h->Wp) + 0.5);
      if (by_px > rc_auto) rc_auto = by_px;
    }
    h->refine_chunk = rc_auto < 1 ? 1 : (rc_auto > 8 ? 8 : rc_auto);
  }
  h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (h->refine_chunk > h->max_batch) h->refine_chunk = h->max_batch;
  if (h->refine_chunk_x3 > h->refine_chunk) h->refine_chunk_x3 = h->refine_chunk;

  // the kernels use 32-bit element / byte offsets inside one tensor: keep every tensor below 2^32
  {
    // with the piece / chunk sizes alloc_ws will really use (a piece is never smaller than a chunk), and for the
    // activation tensor of EVERY refinement level (a coarse level holds up to rb * 4^k pairs of a relatively more
    // padded plane)
    const int pb = piece_pairs(h, h->max_batch, h->refine_chunk);
    const double low_elems = 2.0 * pb * kC * (h->Hp / 2.0) * (h->Wp / 2.0);
    const double vol_elems = (double)pb * h->Dl * kC * h->hl * h->wl;
    double ref_bytes = 0;
    for (int lv = 0; lv < h->levels; ++lv) {
      const RefGeom& rg = h->tw[lv].rg;
      // (SN_PREC_F16X3's lo tensor sits behind the hi tensor; its offset is folded into 64-bit base pointers)
      const double b = ((double)level_chunk_pairs(h->refine_chunk, pb, lv) * 4.0 * rg.Hs * rg.Ws + (double)ref_slack(rg) + (double)ref_front(rg)) * 16.0;
      if (b > ref_bytes) ref_bytes = b;
    }
    if (low_elems >= 4.0e9 || ref_bytes >= 4.0e9 || vol_elems >= 4.0e9) {
      delete h;
      return SN_ERR_ARG;
    }
  }
  int rc = check_device(h);
  auto fail = [&](int code) {
    create_fail(code, h->err.empty() ? "engine set-up" : h->err.c_str());
    sn_destroy(h);
    return code;
  };
  if (rc) return fail(rc);
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(SN_ERR_DEVICE);
  for (auto& e : h->ev)
    if (hipEventCreate(&e) != hipSuccess) return fail(SN_ERR_DEVICE);
  for (auto& e : h->ev_dom)
    if (hipEventCreate(&e) != hipSuccess) return fail(SN_ERR_DEVICE);
  if (hipEventCreateWithFlags(&h->ev_stats, hipEventDisableTiming) != hipSuccess) return fail(SN_ERR_DEVICE);
  if (c.precision == SN_PREC_AUTO)
    for (auto& p : h->chk)
      if (dalloc(&p, (size_t)H * W) != hipSuccess) return fail(SN_ERR_NOMEM);
  // (Disjoint CU sets for the pipeline streams through hipExtStreamCreateWithCUMask were measured and dropped:
  // 1940 pairs/s shared vs 1700 / 1680 / 1510 with 64 / 96 / 128 CUs split off for the low-resolution branch.)
  // High-priority pipeline streams (stream_prio = 1: sn_mgpu_create for its own exchange streams when more than one
  // device takes part; SN_STREAM_PRIORITY=1 / 0 forces it on / off for any caller).  HIP multiplexes streams onto
  // GPU_MAX_HW_QUEUES (4) hardware queues PER PRIORITY LEVEL, and two streams that share a hardware queue run in order: a
  // caller's other streams (a communication library's receive kernels on the gather root, copy streams) can land on the
  // tower's queue and serialise with it.  High-priority streams draw from their own queues.  Not the default for a
  // single engine: the host-to-host paths measured 30 % slower with it.
  int prio = 0;
  {
    int least = 0, greatest = 0;
    const char* e = getenv("SN_STREAM_PRIORITY");
    const bool want = (e && *e) ? atoi(e) == 1 : stream_prio == 1;
    if (want && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) prio = greatest;
  }
  h->stream_prio = prio != 0;
  auto mk_stream = [&](hipStream_t* st) {
    return prio != 0 ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
  };
  if (mk_stream(&h->s_low) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess)
    return fail(SN_ERR_DEVICE);
  for (int i = 0; i < kMaxTowerStreams; ++i)
    if (mk_stream(&h->s_tow[i]) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_tow_join[i], hipEventDisableTiming) != hipSuccess)
      return fail(SN_ERR_DEVICE);
  for (auto& e : h->ev_piece)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(SN_ERR_DEVICE);
  h->overlap = getenv("SN_NO_OVERLAP") == nullptr;
  h->use_graphs = getenv("SN_NO_GRAPH") == nullptr;
  h->fuse_mode = fuse_env();
  h->tail_fuse = !(getenv("SN_TAIL_FUSE") != nullptr && atoi(getenv("SN_TAIL_FUSE")) == 0);
  if (hipMalloc(reinterpret_cast<void**>(&h->dump), 4096) != hipSuccess) return fail(SN_ERR_NOMEM);

  BlobWalker bw{blob.data()};
  const bool low_x3 = h->precision != SN_PREC_FP32;     // fp16 modes: low-resolution layers on split fp16 operands
  const unsigned abl_w = ablate_mask("SN_ABLATE_W");
#if SN_DIAGNOSTICS
  h->ablate_x = ablate_mask("SN_ABLATE_X");
#endif
  int feat_idx = 0;
  h->fold_down01 = low_x3 && down01_enabled();
  HostLayer hl_down0{};
  for (int i = 0; i < kNDown; ++i) {
    const HostLayer hl_ = bw.next(kC, i == 0 ? 3 : kC, 25);
    if ((rc = upload_conv2d(h, hl_, 4, &h->down[i]))) return fail(rc);
    if (low_x3 && i == 0 && (rc = upload_down0_f16(h, hl_, &h->down0))) return fail(rc);
    if (i == 0) hl_down0 = hl_;
    if (h->fold_down01 && i == 1 && (rc = upload_down01(h, hl_down0, hl_, &h->down01))) return fail(rc);
    if (low_x3 && i > 0 &&
        (rc = upload_x3(h, kC, [&](int co, int c, int tap) { return hl_.w[((size_t)co * kC + c) * 25 + tap]; },
                        &h->down[i], 25, (abl_w >> (kAblDown + i - 1)) & 1u)))
      return fail(rc);
  }
  // fp16 modes: the low-resolution 3x3 / 3x3x3 layers also get split fp16 A-fragments (22-bit operands on the
  // fp16 MFMA, k_conv3x3_c32_x3); SN_PREC_FP32 keeps every contraction on the exact-fp32 MFMA
  auto up2d = [&](ConvLayer* L) -> int {
    const HostLayer hl_ = bw.next(kC, kC, 9);
    int r = upload_conv2d(h, hl_, 8, L);
    if (r || !low_x3) return r;
    const bool z = (abl_w >> (kAblFeat + feat_idx)) & 1u;
    ++feat_idx;
    return upload_x3(h, kC, [&](int co, int c, int tap) { return hl_.w[((size_t)co * kC + c) * 9 + tap]; }, L, 9, z);
  };
  for (int i = 0; i < kNFeatRes; ++i)
    for (int j = 0; j < 2; ++j)
      if ((rc = up2d(&h->fres[i][j]))) return fail(rc);
  if ((rc = up2d(&h->fout))) return fail(rc);
  for (int i = 0; i < kNAgg; ++i) {
    const HostLayer hl_ = bw.next(kC, kC, 27);
    if ((rc = upload_conv3d(h, hl_, &h->agg[i]))) return fail(rc);
    if (low_x3 && (rc = upload_x3(h, 96, [&](int co, int c, int tap) {      // c = kz*32 + ci
          return hl_.w[(((size_t)co * kC + (c & 31)) * 3 + (c >> 5)) * 9 + tap];
        }, &h->agg[i], 9, (abl_w >> (kAblAgg + i)) & 1u)))
      return fail(rc);
  }
  {
    const HostLayer hl_ = bw.next(1, kC, 27);
    if ((rc = upload_head(h, hl_, &h->aout))) return fail(rc);
    if (low_x3 && (rc = upload_agg_head_frag(h, hl_, &h->aout))) return fail(rc);
  }
  for (int lv = 0; lv < h->levels; ++lv) {          // blob order: tower of level 0, then (multi) levels 1, 2, 3
    Tower& T = h->tw[lv];
    {
      const HostLayer hl_ = bw.next(kC, 4, 9);
      if ((rc = upload_conv2d(h, hl_, 4, &T.rin))) return fail(rc);
      if (h->precision != SN_PREC_FP32 && (rc = upload_refin_f16(h, hl_, &T.refin))) return fail(rc);
    }
    for (int i = 0; i < kNRefRes; ++i)
      for (int j = 0; j < 2; ++j) {
        const HostLayer hl_ = bw.next(kC, kC, 9);
        if (h->precision == SN_PREC_FP32 && (rc = upload_conv2d(h, hl_, 8, &T.rres[i][j]))) return fail(rc);
        if (want_x3 && (rc = upload_ref_f16x3(h, hl_, &T.rres16x3[i][j]))) return fail(rc);
        if (want_f16 && (rc = upload_ref_f16(h, hl_, &T.rres16[i][j]))) return fail(rc);
      }
    if ((rc = upload_head(h, bw.next(1, kC, 9), &T.rout))) return fail(rc);
  }
  if (bw.off != blob.size()) return fail(SN_ERR_FORMAT);

  if ((rc = alloc_ws(h, &h->ws, h->max_batch, h->refine_chunk, h->tower_streams, h->refine_chunk_x3)))
    return fail(rc == SN_ERR_DEVICE ? SN_ERR_NOMEM : rc);
  *out = h;
  return SN_OK;
}

int sn_destroy(sn_handle* h) {
  if (!h) return SN_ERR_ARG;
  hipSetDevice(h->device);
  hipDeviceSynchronize();
  auto free_conv = [](ConvLayer& l) {
    hipFree(l.wx3);
    hipFree(l.wpk);
    hipFree(l.bias);
  };
  for (auto& l : h->down) free_conv(l);
  hipFree(h->down0.wfrag);
  hipFree(h->down01.wfrag);
  hipFree(h->down01.bias);
  for (auto& b : h->fres)
    for (auto& l : b) free_conv(l);
  free_conv(h->fout);
  for (auto& l : h->agg) free_conv(l);
  for (auto& T : h->tw) {
    hipFree(T.refin.wfrag);
    free_conv(T.rin);
    for (auto& b : T.rres)
      for (auto& l : b) free_conv(l);
    for (auto& b : T.rres16)
      for (auto& l : b) {
        hipFree(l.wfrag);
        hipFree(l.bias);
      }
    for (auto& b : T.rres16x3)
      for (auto& l : b) {
        hipFree(l.wfrag);
        hipFree(l.bias);
      }
    hipFree(T.rout.w);
  }
  hipFree(h->dump);
  hipFree(h->aout.w);
  hipFree(h->aout.pfrag);
  for (auto p : h->chk) hipFree(p);
  if (h->ev_stats) hipEventDestroy(h->ev_stats);
  free_ws(&h->ws);
  for (auto& s : h->slots) {
    free_ws(&s.ws);
    if (s.pin_in) hipHostFree(s.pin_in);
    if (s.pin_raw) hipHostFree(s.pin_raw);
    if (s.pin_disp) hipHostFree(s.pin_disp);
    for (auto& gk : s.gexec)
      for (auto& gm : gk)
        for (auto& g : gm)
          if (g) hipGraphExecDestroy(g);
    if (s.ev0) hipEventDestroy(s.ev0);
    if (s.ev1) hipEventDestroy(s.ev1);
    if (s.stream) hipStreamDestroy(s.stream);
  }
  for (auto& e : h->ev)
    if (e) hipEventDestroy(e);
  for (auto& e : h->ev_dom)
    if (e) hipEventDestroy(e);
  for (auto& e : h->ev_piece)
    if (e) hipEventDestroy(e);
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->ev_join) hipEventDestroy(h->ev_join);
  if (h->s_low) hipStreamDestroy(h->s_low);
  for (auto& st : h->s_tow)
    if (st) hipStreamDestroy(st);
  for (auto& e : h->ev_tow_join)
    if (e) hipEventDestroy(e);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return SN_OK;
}

int sn_get_io_info(const sn_handle* h, sn_io_info* info) {
  if (!h || !info) return SN_ERR_ARG;
  memset(info, 0, sizeof *info);
  info->width = h->W;
  info->height = h->H;
  info->dmax = h->D;
  info->in_channels = 6;
  info->max_batch = h->max_batch;
  info->precision = h->precision;
  info->task_num = h->task_num;
  info->device = h->device;
  info->out_scale = kOutScale;
  info->in_bytes = (size_t)6 * h->H * h->W;
  info->out_bytes = (size_t)4 * h->H * h->W;
  double mac = 0;
  const double wp = h->Wp, hp = h->Hp, wl = h->wl, hl = h->hl, dl = h->Dl;
  for (int k = 1; k <= kNDown; ++k) mac += 2.0 * (wp * hp / (double)(1 << (2 * k))) * kC * (k == 1 ? 3 : kC) * 25;
  mac += 2.0 * (2 * kNFeatRes + 1) * wl * hl * kC * kC * 9;
  mac += kNAgg * dl * hl * wl * kC * kC * 27 + dl * hl * wl * kC * 27;
  for (int k = 0; k < h->levels; ++k)
    mac += (wp * hp / (double)(1 << (2 * k))) * (4.0 * kC * 9 + 2.0 * kNRefRes * kC * kC * 9 + kC * 9);
  info->flops_per_pair = 2.0 * mac;
  info->refine_levels = h->levels;
  info->precision_selected = h->precision == SN_PREC_AUTO ? h->actl.st.mode : h->precision;
  info->refine_chunk = chunk_pairs(h, h->ws, info->precision_selected);
  info->piece = h->ws.pb;
  info->tower_streams = h->ws.ns;
  return SN_OK;
}

int sn_abi_version(void) { return SN_ABI_VERSION; }

// ---- SN_PREC_AUTO's state machine: pure functions, no device (tests/test_auto_precision.py) -------------------------------
// Envelope of the fp16 tower per shape class, in full-resolution pixels of residual_px = sum_k 2^k mean |D_k r_k|: the
// largest value below which EVERY weight draw of the sensitivity tables kept EPE < 1e-3 px against the oracle
// = SN_AUTO_BUDGET_PX over the worst error per pixel of residual of the table (profiles/r06_auto_envelope_*.txt: 8 seeds x
// head gain {1, 2, 4, 8}; single-scale 1280x720: 1.8e-4 .. 8.3e-4 px per px; hierarchical 1242x375: 1.1e-4 .. 2.9e-4 px per
// px of the 2^k-weighted sum — a coarse level's error is upsampled with its map).  The self-check replaces this prior by the
// model's own slope (sn_auto_limit_px).
double sn_auto_envelope_px(int refine_levels) { return refine_levels > 1 ? kAutoEnvelopeMulti : kAutoEnvelopeSingle; }

int sn_auto_init(sn_auto_state* s, int refine_levels) {
  if (!s) return SN_ERR_ARG;
  s->mode = SN_PREC_F16;
  s->calm = 0;
  s->envelope_px = sn_auto_envelope_px(refine_levels);
  s->epe_per_px = 0.0;
  s->running_px = -1.0;
  s->switches = 0;
  return SN_OK;
}

double sn_auto_limit_px(const sn_auto_state* s) {
  if (!s) return 0.0;
  if (!(s->epe_per_px > 0.0)) return s->envelope_px;        // nothing measured on this model yet: the class envelope
  const double cap = SN_AUTO_ENVELOPE_CAP * s->envelope_px, own = SN_AUTO_BUDGET_PX / s->epe_per_px;
  return own < cap ? own : cap;
}

int sn_auto_observe(sn_auto_state* s, double residual_px) {
  if (!s) return SN_ERR_ARG;
  if (!(residual_px >= 0.0)) residual_px = 1e30;       // NaN / negative: nothing the fp16 tower should be trusted with
  s->running_px = s->running_px < 0.0 ? residual_px : 0.75 * s->running_px + 0.25 * residual_px;
  const double lim = sn_auto_limit_px(s);
  if (s->mode == SN_PREC_F16) {
    if (residual_px > lim) {
      s->mode = SN_PREC_F16X3;
      s->calm = 0;
      ++s->switches;
    }
  } else {
    if (residual_px < SN_AUTO_REENTRY * lim) {
      if (++s->calm >= SN_AUTO_CALM_CALLS) {
        s->mode = SN_PREC_F16;
        s->calm = 0;
        ++s->switches;
      }
    } else {
      s->calm = 0;
    }
  }
  return s->mode;
}

int sn_get_refine_stats(sn_handle* h, sn_refine_stats* out) {
  if (!h || !out) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  if ((rc = fold_pending(h, true))) return rc;
  std::lock_guard<std::mutex> lk(h->mu);
  const AutoCtl& a = h->actl;
  memset(out, 0, sizeof *out);
  out->levels = h->levels;
  out->precision = h->precision;
  out->precision_selected = h->precision == SN_PREC_AUTO ? a.st.mode : h->precision;
  out->precision_last = a.last_mode;
  out->calls = a.calls;
  out->pairs = a.pairs;
  out->switches = a.st.switches;
  out->reruns = a.reruns;
  for (int lv = 0; lv < kMaxLevels; ++lv) out->level_px[lv] = a.last_level[lv];
  out->residual_px = a.last_res;
  out->running_px = a.st.running_px < 0.0 ? 0.0 : a.st.running_px;
  out->envelope_px = a.st.envelope_px;
  out->limit_px = sn_auto_limit_px(&a.st);
  out->selfcheck_epe_px = a.selfcheck_epe;
  out->selfcheck_residual_px = a.selfcheck_res;
  return SN_OK;
}

int sn_infer_batch(sn_handle* h, int n, const int8_t* in, int32_t* out_i32, float* out_disp, int mem,
                   void* stream) {
  if (!h) return SN_ERR_ARG;
  if (!in || (!out_i32 && !out_disp) || n <= 0 || n > h->max_batch || (mem != SN_MEM_HOST && mem != SN_MEM_DEVICE)) {
    set_err(h, "sn_infer_batch: bad arguments");
    return SN_ERR_ARG;
  }
  int rc = check_device(h);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  const size_t HW = (size_t)h->H * h->W;
  const int8_t* din = in;
  int32_t* draw = out_i32;
  float* ddisp = out_disp;
  if (mem == SN_MEM_HOST) {
    HIP_TRY(h, hipMemcpyAsync(h->ws.in6, in, (size_t)n * 6 * HW, hipMemcpyHostToDevice, st));
    din = h->ws.in6;
    draw = out_i32 ? h->ws.out_raw : nullptr;
    ddisp = out_disp ? h->ws.out_disp : nullptr;
  }
  auto post = [&]() -> int {
    if (mem == SN_MEM_HOST) {
      if (out_i32) HIP_TRY(h, hipMemcpyAsync(out_i32, draw, (size_t)n * HW * 4, hipMemcpyDeviceToHost, st));
      if (out_disp) HIP_TRY(h, hipMemcpyAsync(out_disp, ddisp, (size_t)n * HW * 4, hipMemcpyDeviceToHost, st));
    }
    return SN_OK;
  };
  return run_forward(h, st, n, din, ddisp, draw, n == 1, mem == SN_MEM_HOST || !stream, post);
}

int sn_infer_i8(sn_handle* h, const int8_t* in, int32_t* out_i32, float* out_disp, int mem, void* stream) {
  return sn_infer_batch(h, 1, in, out_i32, out_disp, mem, stream);
}

static int pre_args_ok(sn_handle* h, int w, int hp) { return w == h->W && hp == h->H; }

int sn_preprocess_nv12(sn_handle* h, const uint8_t* left, const uint8_t* right, int w, int h_px,
                       int8_t* out6, int mem, void* stream) {
  if (!h) return SN_ERR_ARG;
  if (!left || !right || !out6 || w <= 0 || h_px <= 0 || (w & 3) || (h_px & 1) ||
      (size_t)w * h_px > (size_t)h->W * h->H || (mem != SN_MEM_HOST && mem != SN_MEM_DEVICE)) {
    set_err(h, "sn_preprocess_nv12: bad arguments");
    return SN_ERR_ARG;
  }
  int rc = check_device(h);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  const size_t eye = (size_t)w * h_px * 3 / 2;
  const uint8_t *dl = left, *dr = right;
  int8_t* dout = out6;
  if (mem == SN_MEM_HOST) {
    HIP_TRY(h, hipMemcpyAsync(h->ws.nv12, left, eye, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->ws.nv12 + eye, right, eye, hipMemcpyHostToDevice, st));
    dl = h->ws.nv12;
    dr = h->ws.nv12 + eye;
    dout = h->ws.in6;
  } else if (((uintptr_t)left | (uintptr_t)right | (uintptr_t)out6) & 3) {
    return SN_ERR_ARG;
  }
  const long total = 6L * h_px * (w >> 2);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_pre_nv12, dim3(blocks), dim3(256), 0, st, dl, dr, w, w, h_px, dout);
  HIP_TRY(h, hipGetLastError());
  if (mem == SN_MEM_HOST)
    HIP_TRY(h, hipMemcpyAsync(out6, dout, (size_t)6 * w * h_px, hipMemcpyDeviceToHost, st));
  if (mem == SN_MEM_HOST || !stream) HIP_TRY(h, hipStreamSynchronize(st));
  return SN_OK;
}

int sn_infer_sbs_nv12(sn_handle* h, const uint8_t* sbs, int w2, int h_px, int32_t* out_i32, float* out_disp,
                      int8_t* out_tensor, int mem, void* stream) {
  if (!h) return SN_ERR_ARG;
  // geometry check of FeedImg (stereonet_node.cpp:682-690): height == model h, width == 2 * model w
  if (!sbs || (!out_i32 && !out_disp) || !pre_args_ok(h, w2 / 2, h_px) || (w2 & 7) || (h_px & 1) ||
      (mem != SN_MEM_HOST && mem != SN_MEM_DEVICE)) {
    set_err(h, "sn_infer_sbs_nv12: image size does not match the model input");
    return SN_ERR_ARG;
  }
  int rc = check_device(h);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  const int w = w2 / 2;
  const size_t HW = (size_t)h->H * h->W;
  const uint8_t* dsrc = sbs;
  if (mem == SN_MEM_HOST) {
    HIP_TRY(h, hipMemcpyAsync(h->ws.nv12, sbs, HW * 3, hipMemcpyHostToDevice, st));
    dsrc = h->ws.nv12;
  } else if ((uintptr_t)sbs & 3) {
    return SN_ERR_ARG;
  }
  int8_t* din = (mem == SN_MEM_DEVICE && out_tensor) ? out_tensor : h->ws.in6;
  const long total = 6L * h_px * (w >> 2);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_pre_nv12, dim3(blocks), dim3(256), 0, st, dsrc, dsrc + w, w2, w, h_px, din);
  HIP_TRY(h, hipGetLastError());
  int32_t* draw = out_i32;
  float* ddisp = out_disp;
  if (mem == SN_MEM_HOST) {
    draw = out_i32 ? h->ws.out_raw : nullptr;
    ddisp = out_disp ? h->ws.out_disp : nullptr;
  }
  auto post = [&]() -> int {
    if (mem == SN_MEM_HOST) {
      if (out_i32) HIP_TRY(h, hipMemcpyAsync(out_i32, draw, HW * 4, hipMemcpyDeviceToHost, st));
      if (out_disp) HIP_TRY(h, hipMemcpyAsync(out_disp, ddisp, HW * 4, hipMemcpyDeviceToHost, st));
      if (out_tensor) HIP_TRY(h, hipMemcpyAsync(out_tensor, din, HW * 6, hipMemcpyDeviceToHost, st));
    }
    return SN_OK;
  };
  return run_forward(h, st, 1, din, ddisp, draw, true, mem == SN_MEM_HOST || !stream, post);
}

// FeedImg's split + CvtNV12Data2Tensors for a batch of side-by-side frames (device or host buffers): n frames of
// 3*H*W bytes -> n int8 model tensors of 6*H*W bytes.  The streaming ingest of bench.py --stream: the host ships the
// 2.76 MB camera frame instead of the 5.53 MB tensor.
int sn_preprocess_sbs_nv12_batch(sn_handle* h, int n, const uint8_t* sbs, int w2, int h_px, int8_t* out6, int mem,
                                 void* stream) {
  if (!h) return SN_ERR_ARG;
  if (!sbs || !out6 || n <= 0 || !pre_args_ok(h, w2 / 2, h_px) || (w2 & 7) || (h_px & 1) ||
      (mem != SN_MEM_HOST && mem != SN_MEM_DEVICE) || (mem == SN_MEM_HOST && n > h->max_batch)) {
    set_err(h, "sn_preprocess_sbs_nv12_batch: bad arguments");
    return SN_ERR_ARG;
  }
  if (mem == SN_MEM_DEVICE && (((uintptr_t)sbs | (uintptr_t)out6) & 3)) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : h->stream;
  const int w = w2 / 2;
  const size_t HW = (size_t)h->H * h->W;
  const long total = 6L * h_px * (w >> 2);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  for (int i = 0; i < n; ++i) {
    const uint8_t* src = sbs + (size_t)i * 3 * HW;
    int8_t* dst = out6 + (size_t)i * 6 * HW;
    if (mem == SN_MEM_HOST) {           // one frame at a time through the NV12 staging buffer
      HIP_TRY(h, hipMemcpyAsync(h->ws.nv12, src, 3 * HW, hipMemcpyHostToDevice, st));
      src = h->ws.nv12;
      dst = h->ws.in6 + (size_t)i * 6 * HW;
    }
    hipLaunchKernelGGL(k_pre_nv12, dim3(blocks), dim3(256), 0, st, src, src + w, w2, w, h_px, dst);
  }
  HIP_TRY(h, hipGetLastError());
  if (mem == SN_MEM_HOST) HIP_TRY(h, hipMemcpyAsync(out6, h->ws.in6, (size_t)n * 6 * HW, hipMemcpyDeviceToHost, st));
  if (mem == SN_MEM_HOST || !stream) HIP_TRY(h, hipStreamSynchronize(st));
  return SN_OK;
}

// ---- async task slots (DnnNode::Run with is_sync_mode = false) ---------------------------------------
static int ensure_slots(sn_handle* h) {
  if (!h->slots.empty()) return SN_OK;
  const size_t HW = (size_t)h->H * h->W;
  h->slots.resize(h->task_num);
  for (auto& s : h->slots) {
    HIP_TRY(h, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    HIP_TRY(h, hipEventCreate(&s.ev0));
    HIP_TRY(h, hipEventCreate(&s.ev1));
    int rc = alloc_ws(h, &s.ws, 1, 1, 1, 1);
    if (rc) return rc;
    HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&s.pin_in), 6 * HW, hipHostMallocDefault));
    HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&s.pin_raw), 4 * HW, hipHostMallocDefault));
    HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&s.pin_disp), 4 * HW, hipHostMallocDefault));
  }
  return SN_OK;
}

// kind 0: `in` is the int8 model tensor (6*H*W bytes); kind 1: the raw 2W x H side-by-side NV12 frame of FeedImg
// (3*H*W bytes: half the H2D traffic; split + chroma replication + ^0x80 run on the GPU in k_pre_nv12)
static int submit_common(sn_handle* h, const void* in, int kind, int32_t* out_i32, float* out_disp, int timeout_ms,
                         uint64_t* ticket) {
  if (!h || !in || (!out_i32 && !out_disp) || !ticket) return SN_ERR_ARG;
  std::unique_lock<std::mutex> lk(h->mu);
  int rc = check_device(h);
  if (rc) return rc;
  if ((rc = ensure_slots(h))) return rc;
  Slot* s = nullptr;
  auto find_free = [&]() {
    for (auto& c : h->slots)
      if (c.ticket == 0) {
        s = &c;
        return true;
      }
    return false;
  };
  if (timeout_ms < 0) {
    h->cv.wait(lk, find_free);
  } else if (!h->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), find_free)) {
    return SN_ERR_BUSY;
  }
  const size_t HW = (size_t)h->H * h->W;
  s->ticket = h->next_ticket++;
  s->user_raw = out_i32;
  s->user_disp = out_disp;
  // Every error exit below must hand the slot back (a slot left busy would make a later submit with
  // timeout -1 — what the node passes — block forever); *ticket is written on success only.
  struct SlotGuard {
    sn_handle* h;
    Slot* s;
    bool armed = true;
    ~SlotGuard() {
      if (!armed) return;
      s->ticket = 0;            // h->mu is still held by the caller's unique_lock
      h->cv.notify_one();
    }
  } guard{h, s};
  memcpy(s->pin_in, in, (kind == 1 ? 3 : 6) * HW);   // the caller may release its buffer as soon as we return
  const int mask = (out_i32 ? 1 : 0) | (out_disp ? 2 : 0);
  // arithmetic of this request: an SN_PREC_AUTO handle's current one (sn_wait folds the request's statistic in and repeats
  // a request that left the fp16 tower's envelope)
  const int mode = h->precision == SN_PREC_AUTO ? h->actl.st.mode : h->precision;
  // A streamed tower launch is one 160 KB-LDS workgroup per CU: while it runs nothing of another request fits the chip, so
  // requests in flight together simply queue (four in flight: the sum of their kernel times).  A request submitted while
  // others are in flight therefore leaves an eighth of the CUs to them: their low-resolution launches (tens of workgroups,
  // latency bound) run beside its tower — 2015-2030 -> 2135 pairs/s with four in flight (profiles/r06_async_wgs.txt), at
  // +14 % tower time for that request; a request that finds the GPU idle keeps every CU (latency of a lone frame unchanged).
  int others = 0;
  for (auto& c : h->slots)
    if (&c != s && c.ticket != 0) ++others;
  static const bool share_env = !(getenv("SN_ASYNC_SHARE") != nullptr && atoi(getenv("SN_ASYNC_SHARE")) == 0);
  const int shared = (others > 0 && share_env) ? 1 : 0;
  s->ws.tower_cu = shared ? h->num_cu * 7 / 8 : 0;
  const int mi = (mode == SN_PREC_F16X3 ? 0 : 2) + shared;
  s->mode_run = mode;
  auto enqueue = [&]() -> int {
    if (kind == 1) {
      HIP_TRY(h, hipMemcpyAsync(s->ws.nv12, s->pin_in, 3 * HW, hipMemcpyHostToDevice, s->stream));
      const long total = 6L * h->H * (h->W >> 2);
      const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
      hipLaunchKernelGGL(k_pre_nv12, dim3(blocks), dim3(256), 0, s->stream, s->ws.nv12, s->ws.nv12 + h->W, 2 * h->W, h->W,
                         h->H, s->ws.in6);
      HIP_TRY(h, hipGetLastError());
    } else {
      HIP_TRY(h, hipMemcpyAsync(s->ws.in6, s->pin_in, 6 * HW, hipMemcpyHostToDevice, s->stream));
    }
    const bool prof = h->profiling;
    h->profiling = false;   // stage events belong to the synchronous path
    const int r = forward(h, s->ws, s->stream, 1, s->ws.in6, out_disp ? s->ws.out_disp : nullptr,
                          out_i32 ? s->ws.out_raw : nullptr, false, mode);
    h->profiling = prof;
    if (r) return r;
    if (out_i32) HIP_TRY(h, hipMemcpyAsync(s->pin_raw, s->ws.out_raw, 4 * HW, hipMemcpyDeviceToHost, s->stream));
    if (out_disp) HIP_TRY(h, hipMemcpyAsync(s->pin_disp, s->ws.out_disp, 4 * HW, hipMemcpyDeviceToHost, s->stream));
    return SN_OK;
  };
  if (h->use_graphs && !s->gexec[kind][mask][mi] && s->uses[kind][mask][mi] >= 1) {
    // capture on the second use (the first, un-captured run has done every one-time initialisation)
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      const int r = enqueue();
      const hipError_t e = hipStreamEndCapture(s->stream, &graph);
      if (r == SN_OK && e == hipSuccess && graph &&
          hipGraphInstantiate(&s->gexec[kind][mask][mi], graph, nullptr, nullptr, 0) != hipSuccess)
        s->gexec[kind][mask][mi] = nullptr;
      if (graph) hipGraphDestroy(graph);
    }
    if (!s->gexec[kind][mask][mi]) {
      (void)hipGetLastError();
      h->use_graphs = false;          // capture unsupported here: keep issuing plain launches (same kernels)
    }
  }
  HIP_TRY(h, hipEventRecord(s->ev0, s->stream));
  if (s->gexec[kind][mask][mi]) {
    HIP_TRY(h, hipGraphLaunch(s->gexec[kind][mask][mi], s->stream));
  } else {
    rc = enqueue();
    if (rc) return rc;
    ++s->uses[kind][mask][mi];
  }
  HIP_TRY(h, hipEventRecord(s->ev1, s->stream));
  guard.armed = false;
  *ticket = s->ticket;
  return SN_OK;
}

int sn_submit(sn_handle* h, const int8_t* in, int32_t* out_i32, float* out_disp, int timeout_ms,
              uint64_t* ticket) {
  return submit_common(h, in, 0, out_i32, out_disp, timeout_ms, ticket);
}

int sn_submit_nv12(sn_handle* h, const uint8_t* sbs, int w2, int h_px, int32_t* out_i32, float* out_disp,
                   int timeout_ms, uint64_t* ticket) {
  if (!h) return SN_ERR_ARG;
  // geometry check of FeedImg (stereonet_node.cpp:682-690): height == model h, width == 2 * model w
  if (!pre_args_ok(h, w2 / 2, h_px) || (w2 & 7) || (h_px & 1)) {
    set_err(h, "sn_submit_nv12: image size does not match the model input");
    return SN_ERR_ARG;
  }
  return submit_common(h, sbs, 1, out_i32, out_disp, timeout_ms, ticket);
}

int sn_wait(sn_handle* h, uint64_t ticket, float* infer_ms) {
  if (!h || ticket == 0) return SN_ERR_ARG;
  Slot* s = nullptr;
  {
    std::lock_guard<std::mutex> lk(h->mu);
    for (auto& c : h->slots)
      if (c.ticket == ticket) s = &c;
  }
  if (!s) return SN_ERR_TICKET;
  hipSetDevice(h->device);
  HIP_TRY(h, hipEventSynchronize(s->ev1));
  const size_t HW = (size_t)h->H * h->W;
  {
    // the request's refinement statistic; SN_PREC_AUTO: self-check on the first request, and a request that left the fp16
    // tower's envelope is repeated in SN_PREC_F16X3 on its own stream before its maps are handed over
    double lvl[kMaxLevels], res = 0.0;
    read_stats(h, s->ws, 1, lvl, &res);
    const bool is_auto = h->precision == SN_PREC_AUTO;
    int rc = SN_OK;
    if (is_auto && s->mode_run == SN_PREC_F16 && (rc = auto_selfcheck(h, s->ws, s->stream, s->ws.in6))) return rc;
    count_call(h, 1);
    const int next = fold_stats(h, lvl, res, 1, s->mode_run);
    if (is_auto && s->mode_run == SN_PREC_F16 && next == SN_PREC_F16X3) {
      if ((rc = forward(h, s->ws, s->stream, 1, s->ws.in6, s->user_disp ? s->ws.out_disp : nullptr,
                        s->user_raw ? s->ws.out_raw : nullptr, false, SN_PREC_F16X3)))
        return rc;
      if (s->user_raw) HIP_TRY(h, hipMemcpyAsync(s->pin_raw, s->ws.out_raw, 4 * HW, hipMemcpyDeviceToHost, s->stream));
      if (s->user_disp) HIP_TRY(h, hipMemcpyAsync(s->pin_disp, s->ws.out_disp, 4 * HW, hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(h, hipEventRecord(s->ev1, s->stream));
      HIP_TRY(h, hipEventSynchronize(s->ev1));
      read_stats(h, s->ws, 1, lvl, &res);
      fold_stats(h, lvl, res, 1, SN_PREC_F16X3, false);
    }
  }
  if (s->user_raw) memcpy(s->user_raw, s->pin_raw, 4 * HW);
  if (s->user_disp) memcpy(s->user_disp, s->pin_disp, 4 * HW);
  if (infer_ms) {
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, s->ev0, s->ev1));
    *infer_ms = ms;
  }
  {
    std::lock_guard<std::mutex> lk(h->mu);
    s->ticket = 0;
  }
  h->cv.notify_one();
  return SN_OK;
}

int sn_synchronize(sn_handle* h) {
  if (!h) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (auto& s : h->slots) HIP_TRY(h, hipStreamSynchronize(s.stream));
  return SN_OK;
}

// ---- measurement hooks ---------------------------------------------------------------------------------
int sn_set_profiling(sn_handle* h, int enable) {
  if (!h) return SN_ERR_ARG;
  h->profiling = enable != 0;
  return SN_OK;
}

int sn_get_stage_ms(sn_handle* h, float* ms, int count) {
  if (!h || !ms || count <= 0) return SN_ERR_ARG;
  for (int i = 0; i < count && i < SN_STAGE_COUNT; ++i) ms[i] = h->stage_ms[i];
  return SN_OK;
}

int sn_get_dominant_kernel(sn_handle* h, char* name, size_t cap, int* launches, double* flops, double* bytes) {
  if (!h) return SN_ERR_ARG;
  const int cur = h->precision == SN_PREC_AUTO ? h->actl.st.mode : h->precision;
  const bool f16 = cur == SN_PREC_F16;
  const double px = (double)h->Hp * h->Wp * chunk_pairs(h, h->ws, cur);
  if (f16 && h->fuse_mode == 4) {
    // the row-streaming fused residual block: SN_STAGE_DOMINANT times its launches of the first chunk one by one
    int n = 0;
    for (int i = 0; i < kNRefRes; ++i) {
      const bool last = i == kNRefRes - 1;
      if (stream_block_supports(kRefDil[i]) && !(last && h->tail_fuse)) ++n;
    }
    if (n > 0) {     // (n == 0, e.g. SN_STREAM_DIL=0: nothing is streamed — the per-layer description below applies)
      if (name && cap)
        snprintf(name, cap, "%s", "k_ref_block_stream_f16<DIL> (fused residual block: two 3x3 C->C convs + residual, fp16 MFMA 32x32x16)");
      if (launches) *launches = n;
      if (flops) *flops = 2.0 * (2.0 * px * kC * kC * 9);           // two convolutions per launch
      if (bytes) *bytes = px * kC * 2.0 * 2.0;                       // x read once + y written once; t never leaves LDS
      return SN_OK;
    }
  }
  if (cur == SN_PREC_F16X3) {
    int n = 0;
    for (int i = 0; i < kNRefRes; ++i)
      if (stream_x3_supports(kRefDil[i])) ++n;
    if (n > 0) {     // (SN_X3_STREAM=0: the per-layer description below applies)
      if (name && cap)
        snprintf(name, cap, "%s", "k_ref_block_stream_x3<DIL> (fused residual block: two 3x3 C->C convs + residual, 3 fp16 MFMAs 32x32x16 "
                                  "per product on hi/lo split operands; the FLOPs counted are the model's, not the three products')");
      if (launches) *launches = n;
      if (flops) *flops = 2.0 * (2.0 * px * kC * kC * 9);           // two convolutions per launch (algorithmic)
      if (bytes) *bytes = px * kC * 4.0 * 2.0;                       // x (hi + lo) read once + y written once; t never leaves LDS
      return SN_OK;
    }
  }
  if (name && cap)
    snprintf(name, cap, "%s",
             cur == SN_PREC_F16     ? "k_ref_conv_f16<DIL> (refinement 3x3 C->C, fp16 MFMA 32x32x16)"
             : cur == SN_PREC_F16X3 ? "k_ref_conv_f16x3<DIL> (refinement 3x3 C->C, 3x fp16 MFMA on hi/lo split operands)"
                                             : "k_ref_conv_f32<DIL> (refinement 3x3 C->C, weights-stationary, fp32 MFMA 32x32x2)");
  const int n_plain = kNRefRes, n_res = kNRefRes;      // per-layer forms: six launches without, six with a residual
  if (launches) *launches = n_plain + n_res;   // per refinement chunk
  if (flops) *flops = 2.0 * px * kC * kC * 9;
  // algorithmic HBM bytes per launch: read the 32-channel input once + write the output once, plus the
  // residual read on the launches that have one, averaged; element = 2 B (fp16) or 4 B (fp32)
  if (bytes) *bytes = px * kC * (f16 ? 2.0 : 4.0) * (2.0 * n_plain + 3.0 * n_res) / (double)(n_plain + n_res);
  return SN_OK;
}

// ---- parity hooks ----------------------------------------------------------------------------------------
int sn_dbg_conv2d(sn_handle* h, const float* in, int cin, int h_px, int w, const float* wt, const float* bias,
                  int k, int stride, int dil, int lrelu, const float* residual, float* out) {
  DevScope ds;      // frees every tracked device buffer on every return path
  if (!h || !in || !wt || !bias || !out || cin <= 0 || cin > kC) return SN_ERR_ARG;
  if (!((k == 3 && stride == 1) || (k == 5 && stride == 2 && dil == 1))) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  const int taps = k * k;
  const int Ho = stride == 1 ? h_px : h_px / 2, Wo = stride == 1 ? w : w / 2;
  if (stride == 2 && ((h_px & 1) || (w & 1))) return SN_ERR_ARG;
  const bool x3 = (lrelu & 2) != 0, slots = (lrelu & 4) != 0;
  const bool tower32 = (lrelu & 8) != 0;     // bit 3: the fp32 tower kernel (k_ref_conv_f32) instead of the generic one
  const bool dma = (lrelu & 16) != 0;        // bit 4 (5x5 stride 2 on slots): k_down_x3s_dma on zero-bordered tensors
  lrelu &= 1;
  if (x3 != slots || (x3 && !(cin == kC && dil == 1))) return SN_ERR_ARG;     // the split-operand kernel reads slots
  if (dma && !(slots && (k == 3 || !residual))) return SN_ERR_ARG;
  if (slots) {          // split-slot tensors in and out through the weights-stationary kernel (fp16 modes' low-res path)
    ConvLayer Ls;
    HostLayer hls{wt, bias, kC, cin, taps};
    if ((rc = upload_conv2d(h, hls, 8, &Ls))) return rc;
    ds.track(Ls.bias); ds.track(Ls.wpk); ds.track(Ls.wx3);
    if ((rc = upload_x3(h, kC, [&](int co, int c, int tap) { return wt[((size_t)co * kC + c) * taps + tap]; }, &Ls, taps)))
      return rc;
    ds.track(Ls.bias); ds.track(Ls.wpk); ds.track(Ls.wx3);
    std::vector<_Float16> hin, hres, hout((size_t)8 * Ho * Wo * 8);
    host_to_slots(in, 1, h_px, w, hin);
    uint4 *din = nullptr, *dout = nullptr;
    HIP_TRY(h, dalloc(&din, hin.size() / 8));
    ds.track(din);
    HIP_TRY(h, dalloc(&dout, hout.size() / 8));
    ds.track(dout);
    HIP_TRY(h, hipMemcpy(din, hin.data(), hin.size() * 2, hipMemcpyHostToDevice));
    const float* dres = nullptr;
    if (residual) {
      host_to_slots(residual, 1, Ho, Wo, hres);
      HIP_TRY(h, hipMemcpy(dout, hres.data(), hres.size() * 2, hipMemcpyHostToDevice));
      dres = reinterpret_cast<const float*>(dout);
    }
    if (dma && k == 3) {      // k_feat_x3s_dma: zero-bordered input, output and residual (FeatPad)
      const FeatPad g = feat_pad(h_px, w);
      const size_t phw = (size_t)g.PH * g.PW;
      std::vector<_Float16> pin(8 * phw * 8, (_Float16)0.f), pout(8 * phw * 8, (_Float16)0.f);
      for (int img = 0; img < 8; ++img)      // (block, part) images
        for (int y = 0; y < h_px; ++y) {
          memcpy(&pin[((size_t)img * phw + (size_t)(y + 1) * g.PW + 1) * 8], &hin[((size_t)img * h_px + y) * w * 8], (size_t)w * 16);
          if (residual)
            memcpy(&pout[((size_t)img * phw + (size_t)(y + 1) * g.PW + 1) * 8], &hres[((size_t)img * h_px + y) * w * 8], (size_t)w * 16);
        }
      uint4 *pdin = nullptr, *pdout = nullptr;
      HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&pdin), pin.size() * 2));
      ds.track(pdin);
      HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&pdout), pout.size() * 2));
      ds.track(pdout);
      HIP_TRY(h, hipMemcpy(pdin, pin.data(), pin.size() * 2, hipMemcpyHostToDevice));
      HIP_TRY(h, hipMemcpy(pdout, pout.data(), pout.size() * 2, hipMemcpyHostToDevice));
      HIP_TRY(h, hipDeviceSynchronize());
      if (residual)
        HIP_TRY(h, (launch_feat_dma<true, true>(h->stream, Ls, pdin, g, 1, pdout, pdout, lrelu != 0, h->num_cu)));     // in place, as the pipeline
      else
        HIP_TRY(h, (launch_feat_dma<true, false>(h->stream, Ls, pdin, g, 1, pdout, nullptr, lrelu != 0, h->num_cu)));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      HIP_TRY(h, hipMemcpy(pout.data(), pdout, pout.size() * 2, hipMemcpyDeviceToHost));
      for (int img = 0; img < 8; ++img)
        for (int y = 0; y < Ho; ++y)
          memcpy(&hout[((size_t)img * Ho + y) * Wo * 8], &pout[((size_t)img * phw + (size_t)(y + 1) * g.PW + 1) * 8], (size_t)Wo * 16);
      for (size_t i = 0; i < pout.size(); ++i) {
        const size_t sl = i / 8, y = (sl % phw) / g.PW, x = sl % g.PW;
        const bool inside = y >= 1 && y < (size_t)Ho + 1 && x >= 1 && x < (size_t)Wo + 1;
        if (!inside && (float)pout[i] != 0.f) {
          set_err(h, "k_feat_x3s_dma wrote outside the image");
          return SN_ERR_DEVICE;
        }
      }
      host_from_slots(hout, 1, Ho, Wo, out);
      return SN_OK;
    }
    if (dma) {      // zero-bordered input (two pixels), an output grid with a border of its own (3 pixels of slack)
      const SlotGeom gi = down_in_geom(Ho, Wo), go{Ho + 5, Wo + 7, 2, 3};
      const size_t iphw = (size_t)gi.PH * gi.PW, ophw = (size_t)go.PH * go.PW;
      std::vector<_Float16> pin(8 * iphw * 8, (_Float16)0.f), pout(8 * ophw * 8);
      for (int img = 0; img < 8; ++img)      // (block, part) images
        for (int y = 0; y < h_px; ++y)
          memcpy(&pin[((size_t)img * iphw + (size_t)(y + gi.py) * gi.PW + gi.px) * 8], &hin[((size_t)img * h_px + y) * w * 8], (size_t)w * 16);
      uint4 *pdin = nullptr, *pdout = nullptr;
      HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&pdin), pin.size() * 2));
      ds.track(pdin);
      HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&pdout), pout.size() * 2));
      ds.track(pdout);
      HIP_TRY(h, hipMemcpy(pdin, pin.data(), pin.size() * 2, hipMemcpyHostToDevice));
      HIP_TRY(h, memset_now(pdout, 0, pout.size() * 2));
      HIP_TRY(h, launch_down_dma(h->stream, Ls, pdin, 1, Ho, Wo, pdout, go, lrelu != 0, h->num_cu));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      HIP_TRY(h, hipMemcpy(pout.data(), pdout, pout.size() * 2, hipMemcpyDeviceToHost));
      for (int img = 0; img < 8; ++img)
        for (int y = 0; y < Ho; ++y)
          memcpy(&hout[((size_t)img * Ho + y) * Wo * 8], &pout[((size_t)img * ophw + (size_t)(y + go.py) * go.PW + go.px) * 8], (size_t)Wo * 16);
      for (size_t i = 0; i < pout.size(); ++i) {
        const size_t sl = i / 8, y = (sl % ophw) / go.PW, x = sl % go.PW;
        const bool inside = y >= (size_t)go.py && y < (size_t)Ho + go.py && x >= (size_t)go.px && x < (size_t)Wo + go.px;
        if (!inside && (float)pout[i] != 0.f) {
          set_err(h, "k_down_x3s_dma wrote outside the image");
          return SN_ERR_DEVICE;
        }
      }
      host_from_slots(hout, 1, Ho, Wo, out);
      return SN_OK;
    }
    SlotIn ls{din, 0, h_px, w};
    hipError_t e2 = k == 5 ? launch_conv_x3s<5, 2, 32, 4, 32, 32, 1, true, SlotIn>(h->stream, Ls, ls, 1, Ho, Wo, reinterpret_cast<float*>(dout), dres, lrelu != 0, h->num_cu)
                           : launch_conv_x3s<3, 1, 32, 8, 16, 16, 2, true, SlotIn>(h->stream, Ls, ls, 1, Ho, Wo, reinterpret_cast<float*>(dout), dres, lrelu != 0, h->num_cu);
    HIP_TRY(h, e2);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(hout.data(), dout, hout.size() * 2, hipMemcpyDeviceToHost));
    host_from_slots(hout, 1, Ho, Wo, out);
    return SN_OK;
  }
  ConvLayer L;
  HostLayer hl{wt, bias, kC, cin, taps};
  if ((rc = upload_conv2d(h, hl, (k == 5 || cin <= 4) ? 4 : 8, &L))) return rc;
  ds.track(L.bias); ds.track(L.wpk); ds.track(L.wx3);
  float *din = nullptr, *dout = nullptr;
  const size_t nin = (size_t)cin * h_px * w, nout = (size_t)kC * Ho * Wo;
  HIP_TRY(h, dalloc(&din, nin));
  ds.track(din);
  HIP_TRY(h, dalloc(&dout, nout));
  ds.track(dout);
  HIP_TRY(h, hipMemcpy(din, in, nin * 4, hipMemcpyHostToDevice));
  const float* dres = nullptr;
  if (residual) {   // in-place form, as the pipeline uses it
    HIP_TRY(h, hipMemcpy(dout, residual, nout * 4, hipMemcpyHostToDevice));
    dres = dout;
  }
  hipStream_t st = h->stream;
  LoadF32 ld{din, cin, h_px, w};
  hipError_t e;
  if (k == 5) {
    e = (Ho * Wo <= 64 * 128) ? launch_conv<5, 2, 1, 4, 4, 32>(st, L, ld, 1, Ho, Wo, dout, dres, lrelu != 0)
                              : launch_conv<5, 2, 1, 4, 8, 64>(st, L, ld, 1, Ho, Wo, dout, dres, lrelu != 0);
  } else if (cin <= 4) {
    if (dil != 1) return SN_ERR_ARG;
    e = (Ho * Wo <= 64 * 128) ? launch_conv<3, 1, 1, 4, 4, 32>(st, L, ld, 1, Ho, Wo, dout, dres, lrelu != 0)
                              : launch_conv<3, 1, 1, 4, 8, 64>(st, L, ld, 1, Ho, Wo, dout, dres, lrelu != 0);
  } else {
    if (tower32 && ((w & 3) != 0 || cin != kC)) return SN_ERR_ARG;
    e = conv3x3(st, L, din, 1, h_px, w, dil, dout, dres, lrelu != 0, tower32 ? h->num_cu : 0);
  }
  HIP_TRY(h, e);
  HIP_TRY(h, hipStreamSynchronize(st));
  HIP_TRY(h, hipMemcpy(out, dout, nout * 4, hipMemcpyDeviceToHost));
  return SN_OK;
}

int sn_dbg_down0(sn_handle* h, const int8_t* in6, int h_px, int w, const float* wt, const float* bias, int tc,
                 float* out) {
  DevScope ds;      // frees every tracked device buffer on every return path
  if (!h || !in6 || !wt || !bias || !out || h_px <= 0 || w <= 0 || tc != 32) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  const int Hp = (h_px + 15) / 16 * 16, Wp = (w + 15) / 16 * 16, Ho = Hp / 2, Wo = Wp / 2;
  Down0F16 L;
  HostLayer hl{wt, bias, kC, 3, 25};
  if ((rc = upload_down0_f16(h, hl, &L))) return rc;
  ds.track(L.wfrag);
  int8_t* din = nullptr;
  float *dout = nullptr, *dbias = nullptr;
  const size_t nin = (size_t)6 * h_px * w, nout = (size_t)2 * kC * Ho * Wo;     // split slots: same bytes as fp32
  HIP_TRY(h, dalloc(&din, nin));
  ds.track(din);
  HIP_TRY(h, dalloc(&dout, nout));
  ds.track(dout);
  HIP_TRY(h, dalloc(&dbias, kC));
  ds.track(dbias);
  HIP_TRY(h, hipMemcpy(din, in6, nin, hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(dbias, bias, kC * 4, hipMemcpyHostToDevice));
  HIP_TRY(h, launch_down0_f16(h->stream, L, dbias, din, h_px, w, 2, Ho, Wo, dout, h->num_cu));   // the pipeline's kernel
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<_Float16> hs(nout * 2);
  HIP_TRY(h, hipMemcpy(hs.data(), dout, nout * 4, hipMemcpyDeviceToHost));
  host_from_slots(hs, 2, Ho, Wo, out);
  return SN_OK;
}

int sn_dbg_round_kernels_f16(const float* w, int nkernels, float* out) {
  if (!w || !out || nkernels < 0) return SN_ERR_ARG;
  for (int k = 0; k < nkernels; ++k) {          // host only: no device needed
    _Float16 q[9];
    round_kernel_sum_preserving(w + (size_t)k * 9, q);
    for (int t = 0; t < 9; ++t) out[(size_t)k * 9 + t] = (float)q[t];
  }
  return SN_OK;
}

int sn_dbg_compose_down01(const float* w0, const float* b0, const float* w1, const float* b1, float* weff, float* beff) {
  if (!w0 || !b0 || !w1 || !b1 || !weff || !beff) return SN_ERR_ARG;
  std::vector<double> we, be;
  compose_down01(w0, b0, w1, b1, we, be);         // host only: no device needed
  for (size_t i = 0; i < we.size(); ++i) weff[i] = (float)we[i];
  for (size_t i = 0; i < be.size(); ++i) beff[i] = (float)be[i];
  return SN_OK;
}

int sn_dbg_down01(sn_handle* h, const int8_t* in6, int h_px, int w, const float* w0, const float* b0, const float* w1,
                  const float* b1, float* out) {
  DevScope ds;      // frees every tracked device buffer on every return path
  if (!h || !in6 || !w0 || !b0 || !w1 || !b1 || !out || h_px <= 0 || w <= 0) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  const int Hp = (h_px + 15) / 16 * 16, Wp = (w + 15) / 16 * 16, Ho = Hp / 4, Wo = Wp / 4;
  Down01W L;
  const HostLayer l0{w0, b0, kC, 3, 25}, l1{w1, b1, kC, kC, 25};
  rc = upload_down01(h, l0, l1, &L);
  ds.track(L.wfrag);
  ds.track(L.bias);
  if (rc) return rc;
  int8_t* din = nullptr;
  uint4* dout = nullptr;
  const size_t nin = (size_t)6 * h_px * w, nout = (size_t)2 * kC * Ho * Wo;     // split slots: same bytes as fp32
  HIP_TRY(h, dalloc(&din, nin));
  ds.track(din);
  HIP_TRY(h, dalloc(&dout, nout / 4));
  ds.track(dout);
  HIP_TRY(h, hipMemcpy(din, in6, nin, hipMemcpyHostToDevice));
  HIP_TRY(h, launch_down01(h->stream, L, din, h_px, w, 2, Ho, Wo, dout, SlotGeom{Ho, Wo, 0, 0}, h->num_cu));   // the pipeline's kernels
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<_Float16> hs(nout * 2);
  HIP_TRY(h, hipMemcpy(hs.data(), dout, nout * 4, hipMemcpyDeviceToHost));
  host_from_slots(hs, 2, Ho, Wo, out);
  return SN_OK;
}

int sn_dbg_refin(sn_handle* h, const float* disp_low, const int8_t* in6, int h_px, int w, int dmax, const float* wt,
                 const float* bias, int split, float* out) {
  DevScope ds;      // frees every tracked device buffer on every return path
  if (!h || !disp_low || !in6 || !wt || !bias || !out || h_px <= 0 || w <= 0 || dmax <= 0) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  const int Hp = (h_px + 15) / 16 * 16, Wp = (w + 15) / 16 * 16, hl = Hp / 16, wl = Wp / 16;
  const RefGeom g = make_ref_geom(Hp, Wp);
  const size_t slots = ref16_slots(g, 1) + ref_slack(g);
  Down0F16 L;
  HostLayer hl_{wt, bias, kC, 4, 9};
  if ((rc = upload_refin_f16(h, hl_, &L))) return rc;
  ds.track(L.wfrag);
  float *ddl = nullptr, *dbias = nullptr;
  int8_t* din = nullptr;
  uint4* dout = nullptr;
  HIP_TRY(h, dalloc(&ddl, (size_t)hl * wl));
  ds.track(ddl);
  HIP_TRY(h, dalloc(&dbias, kC));
  ds.track(dbias);
  HIP_TRY(h, dalloc(&din, (size_t)6 * h_px * w));
  ds.track(din);
  HIP_TRY(h, dalloc(&dout, 2 * slots));
  ds.track(dout);
  HIP_TRY(h, hipMemcpy(ddl, disp_low, (size_t)hl * wl * 4, hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(dbias, bias, kC * 4, hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(din, in6, (size_t)6 * h_px * w, hipMemcpyHostToDevice));
  HIP_TRY(h, memset_now(dout, 0, 2 * slots * 16));
  HIP_TRY(h, launch_refin_f16(h->stream, L, dbias, ddl, din, false, hl, wl, h_px, w, 1.0f / (float)dmax,
                              UpScale{1.0f / 16.0f, 16.0f}, g, 1, dout, split != 0, slots * 16, h->num_cu));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<_Float16> hout(2 * slots * 8);
  HIP_TRY(h, hipMemcpy(hout.data(), dout, 2 * slots * 16, hipMemcpyDeviceToHost));
  for (int c = 0; c < kC; ++c)
    for (int y = 0; y < Hp; ++y)
      for (int x = 0; x < Wp; ++x) {
        const size_t i = ((((size_t)(c >> 3)) * g.Hs + y + kRefPad) * g.Ws + x + kRefPad) * 8 + (c & 7);
        float v = (float)hout[i];
        if (split) v += (float)hout[i + slots * 8] * kSplitInv;
        out[((size_t)c * Hp + y) * Wp + x] = v;
      }
  for (int c = 0; c < 4; ++c)       // the zero border must have survived
    for (int y = 0; y < g.Hs; ++y)
      for (int x = 0; x < g.Ws; ++x) {
        if (y >= kRefPad && y < kRefPad + Hp && x >= kRefPad && x < kRefPad + Wp) continue;
        for (int e = 0; e < 8; ++e)
          if ((float)hout[(((size_t)c * g.Hs + y) * g.Ws + x) * 8 + e] != 0.f) {
            set_err(h, "ref.in wrote into the zero border");
            return SN_ERR_DEVICE;
          }
      }
  return SN_OK;
}

int sn_dbg_conv3d(sn_handle* h, const float* in, int d, int h_px, int w, const float* wt, const float* bias,
                  int lrelu, float* out) {
  DevScope ds;      // frees every tracked device buffer on every return path
  if (!h || !in || !wt || !bias || !out || d <= 0) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  const bool x3 = (lrelu & 2) != 0, slots = (lrelu & 4) != 0, dma = (lrelu & 8) != 0;   // dma: zero-bordered volumes
  lrelu &= 1;
  if (slots != x3 || (dma && !slots)) return SN_ERR_ARG;        // the split-operand kernels read split-slot volumes
  ConvLayer L;
  HostLayer hl{wt, bias, kC, kC, 27};
  if ((rc = upload_conv3d(h, hl, &L))) return rc;
  ds.track(L.bias); ds.track(L.wpk); ds.track(L.wx3);
  if (x3 && (rc = upload_x3(h, 96, [&](int co, int c, int tap) {
        return wt[(((size_t)co * kC + (c & 31)) * 3 + (c >> 5)) * 9 + tap];
      }, &L)))
    return rc;
  ds.track(L.wx3);        // allocated by upload_x3 just now (the track above saw a null pointer)
  const size_t plane = (size_t)h_px * w, n = (size_t)kC * d * plane;
  // caller layout [ci][d][h][w] (PyTorch) <-> device layout [d][ci][h][w]
  std::vector<float> tmp(n);
  for (int ci = 0; ci < kC; ++ci)
    for (int z = 0; z < d; ++z)
      memcpy(&tmp[((size_t)z * kC + ci) * plane], &in[((size_t)ci * d + z) * plane], plane * 4);
  float *din = nullptr, *dout = nullptr;
  HIP_TRY(h, dalloc(&din, n));
  ds.track(din);
  HIP_TRY(h, dalloc(&dout, n));
  ds.track(dout);
  if (slots) {       // the volume as d split-slot images (same byte count as fp32)
    std::vector<_Float16> hs, ho(n * 2);
    host_to_slots(tmp.data(), d, h_px, w, hs);
    if (dma) {         // k_agg_x3s_dma on the padded layout: planes 1 .. d of d + 2, every (block, part) image with its border
      const VolPad g = vol_pad(d, h_px, w);
      const size_t phw = (size_t)g.PH * g.PW, nsl = g.planes(1) * g.plane_slots();
      std::vector<_Float16> pin(nsl * 8, (_Float16)0.f), pout(nsl * 8);
      for (size_t img = 0; img < (size_t)d * 8; ++img)      // (plane, block, part) images
        for (int y = 0; y < h_px; ++y)
          memcpy(&pin[(((img / 8 + 1) * 8 + img % 8) * phw + (size_t)(y + 1) * g.PW + 1) * 8], &hs[(img * plane + (size_t)y * w) * 8],
                 (size_t)w * 16);
      uint4 *pdin = nullptr, *pdout = nullptr;
      HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&pdin), nsl * 16));
      ds.track(pdin);
      HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&pdout), nsl * 16));
      ds.track(pdout);
      HIP_TRY(h, hipMemcpy(pdin, pin.data(), nsl * 16, hipMemcpyHostToDevice));
      HIP_TRY(h, memset_now(pdout, 0, nsl * 16));
      HIP_TRY(h, launch_agg_dma<true>(h->stream, L, pdin, g, 1, pdout, lrelu != 0, h->num_cu));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      HIP_TRY(h, hipMemcpy(pout.data(), pdout, nsl * 16, hipMemcpyDeviceToHost));
      for (size_t img = 0; img < (size_t)d * 8; ++img)
        for (int y = 0; y < h_px; ++y)
          memcpy(&ho[(img * plane + (size_t)y * w) * 8], &pout[(((img / 8 + 1) * 8 + img % 8) * phw + (size_t)(y + 1) * g.PW + 1) * 8],
                 (size_t)w * 16);
      // the borders must still hold the zeros of the allocation
      for (size_t i = 0; i < nsl * 8; ++i) {
        const size_t sl = i / 8, P = sl / (8 * phw), y = (sl % phw) / g.PW, x = sl % g.PW;
        const bool inside = P >= 1 && P <= (size_t)d && y >= 1 && y <= (size_t)h_px && x >= 1 && x <= (size_t)w;
        if (!inside && (float)pout[i] != 0.f) {
          set_err(h, "k_agg_x3s_dma wrote outside the image");
          return SN_ERR_DEVICE;
        }
      }
    } else {
    HIP_TRY(h, hipMemcpy(din, hs.data(), n * 4, hipMemcpyHostToDevice));
    SlotIn ls{reinterpret_cast<const uint4*>(din), d, h_px, w};
    HIP_TRY(h, (launch_conv_x3s<3, 1, 96, 8, 16, 16, 1, true, SlotIn>(h->stream, L, ls, d, h_px, w, dout, nullptr, lrelu != 0, h->num_cu)));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(ho.data(), dout, n * 4, hipMemcpyDeviceToHost));
    }
    host_from_slots(ho, d, h_px, w, tmp.data());
  } else {
  HIP_TRY(h, hipMemcpy(din, tmp.data(), n * 4, hipMemcpyHostToDevice));
  LoadVol3D lv{din, d, h_px, w};
  HIP_TRY(h, (launch_conv<3, 1, 1, 8, 4, 32>(h->stream, L, lv, d, h_px, w, dout, nullptr, lrelu != 0)));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(tmp.data(), dout, n * 4, hipMemcpyDeviceToHost));
  }
  for (int co = 0; co < kC; ++co)
    for (int z = 0; z < d; ++z)
      memcpy(&out[((size_t)co * d + z) * plane], &tmp[((size_t)z * kC + co) * plane], plane * 4);
  return SN_OK;
}

int sn_dbg_ref_conv_f16(sn_handle* h, const float* in, int h_px, int w, const float* wt, const float* bias, int dil,
                        int lrelu, const float* residual, float* out) {
  DevScope ds;      // frees every tracked device buffer on every return path
  if (!h || !in || !wt || !bias || !out || h_px <= 0 || w <= 0) return SN_ERR_ARG;
  if (dil != 1 && dil != 2 && dil != 4 && dil != 8) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  const RefGeom g = make_ref_geom(h_px, w);
  const size_t slots = ref16_slots(g, 1);
  auto to_dev_layout = [&](const float* src, std::vector<_Float16>& dst) {
    dst.assign(slots * 8, (_Float16)0.f);
    for (int c = 0; c < kC; ++c)
      for (int y = 0; y < h_px; ++y)
        for (int x = 0; x < w; ++x)
          dst[((((size_t)(c >> 3)) * g.Hs + y + kRefPad) * g.Ws + x + kRefPad) * 8 + (c & 7)] =
              (_Float16)src[((size_t)c * h_px + y) * w + x];
  };
  std::vector<_Float16> hin, hres;
  to_dev_layout(in, hin);
  RefLayerF16 L;
  HostLayer hl{wt, bias, kC, kC, 9};
  if ((rc = upload_ref_f16(h, hl, &L))) return rc;
  ds.track(L.bias); ds.track(L.wfrag);
  uint4 *din = nullptr, *dout = nullptr;
  HIP_TRY(h, dalloc(&din, slots));
  ds.track(din);
  HIP_TRY(h, dalloc(&dout, slots));
  ds.track(dout);
  HIP_TRY(h, hipMemcpy(din, hin.data(), slots * 16, hipMemcpyHostToDevice));
  const uint4* dres = nullptr;
  if (residual) {
    to_dev_layout(residual, hres);
    HIP_TRY(h, hipMemcpy(dout, hres.data(), slots * 16, hipMemcpyHostToDevice));
    dres = dout;
  } else {
    HIP_TRY(h, memset_now(dout, 0, slots * 16));
  }
  unsigned* ctr = h->ws.tile_ctr;
  if (!ctr) {
    set_err(h, "sn_dbg_ref_conv_f16 needs an engine created in an fp16 mode");
    return SN_ERR_ARG;
  }
  HIP_TRY(h, hipMemsetAsync(ctr, 0, kTileCtrBytes, h->stream));
  // lrelu bits 1 / 2: force the 8x64 / 8x32 tile variant of the dilation-1 / -2 kernels (default: chosen per launch)
  HIP_TRY(h, ref_conv_f16(h->stream, L, g, h->num_cu, dil, din, dout, dres, 1, (lrelu & 1) != 0, ctr,
                          (lrelu & 2) ? 64 : (lrelu & 4) ? 32 : 0));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<_Float16> hout(slots * 8);
  HIP_TRY(h, hipMemcpy(hout.data(), dout, slots * 16, hipMemcpyDeviceToHost));
  for (int c = 0; c < kC; ++c)
    for (int y = 0; y < h_px; ++y)
      for (int x = 0; x < w; ++x)
        out[((size_t)c * h_px + y) * w + x] =
            (float)hout[((((size_t)(c >> 3)) * g.Hs + y + kRefPad) * g.Ws + x + kRefPad) * 8 + (c & 7)];
  // the zero border must have survived (the kernel never writes outside the valid area)
  for (int c = 0; c < 4; ++c)
    for (int y = 0; y < g.Hs; ++y)
      for (int x = 0; x < g.Ws; ++x) {
        const bool inside = y >= kRefPad && y < kRefPad + h_px && x >= kRefPad && x < kRefPad + w;
        if (inside) continue;
        for (int e = 0; e < 8; ++e)
          if ((float)hout[(((size_t)c * g.Hs + y) * g.Ws + x) * 8 + e] != 0.f) {
            set_err(h, "fp16 conv wrote into the zero border");
            return SN_ERR_DEVICE;
          }
      }
  return SN_OK;
}

int sn_dbg_ref_conv_f16x3(sn_handle* h, const float* in, int h_px, int w, const float* wt, const float* bias, int dil,
                          int lrelu, const float* residual, float* out) {
  DevScope ds;      // frees every tracked device buffer on every return path
  if (!h || !in || !wt || !bias || !out || h_px <= 0 || w <= 0) return SN_ERR_ARG;
  if (dil != 1 && dil != 2 && dil != 4 && dil != 8) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  const RefGeom g = make_ref_geom(h_px, w);
  const size_t lo_slots = ref16_slots(g, 1) + ref_slack(g), slots = 2 * lo_slots;
  auto idx = [&](int c, int y, int x) { return ((((size_t)(c >> 3)) * g.Hs + y + kRefPad) * g.Ws + x + kRefPad) * 8 + (c & 7); };
  auto split_to = [&](const float* src, std::vector<_Float16>& dst) {
    dst.assign(slots * 8, (_Float16)0.f);
    for (int c = 0; c < kC; ++c)
      for (int y = 0; y < h_px; ++y)
        for (int x = 0; x < w; ++x) {
          const float v = src[((size_t)c * h_px + y) * w + x];
          const _Float16 hi = (_Float16)v;
          dst[idx(c, y, x)] = hi;
          dst[lo_slots * 8 + idx(c, y, x)] = (_Float16)((v - (float)hi) * kSplitScale);
        }
  };
  std::vector<_Float16> hin, hres;
  split_to(in, hin);
  RefLayerF16 L;
  if ((rc = upload_ref_f16x3(h, HostLayer{wt, bias, kC, kC, 9}, &L))) return rc;
  ds.track(L.bias); ds.track(L.wfrag);
  uint4 *din = nullptr, *dout = nullptr;
  HIP_TRY(h, dalloc(&din, slots));
  ds.track(din);
  HIP_TRY(h, dalloc(&dout, slots));
  ds.track(dout);
  HIP_TRY(h, hipMemcpy(din, hin.data(), slots * 16, hipMemcpyHostToDevice));
  const uint4* dres = nullptr;
  if (residual) {
    split_to(residual, hres);
    HIP_TRY(h, hipMemcpy(dout, hres.data(), slots * 16, hipMemcpyHostToDevice));
    dres = dout;
  } else {
    HIP_TRY(h, memset_now(dout, 0, slots * 16));
  }
  HIP_TRY(h, ref_conv_f16x3(h->stream, L, g, h->num_cu, dil, din, dout, dres, lo_slots, 1, lrelu != 0));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<_Float16> hout(slots * 8);
  HIP_TRY(h, hipMemcpy(hout.data(), dout, slots * 16, hipMemcpyDeviceToHost));
  for (int c = 0; c < kC; ++c)
    for (int y = 0; y < h_px; ++y)
      for (int x = 0; x < w; ++x)
        out[((size_t)c * h_px + y) * w + x] =
            (float)hout[idx(c, y, x)] + (float)hout[lo_slots * 8 + idx(c, y, x)] * kSplitInv;
  for (int part = 0; part < 2; ++part)       // both zero borders must have survived
    for (int c = 0; c < 4; ++c)
      for (int y = 0; y < g.Hs; ++y)
        for (int x = 0; x < g.Ws; ++x) {
          if (y >= kRefPad && y < kRefPad + h_px && x >= kRefPad && x < kRefPad + w) continue;
          for (int e = 0; e < 8; ++e)
            if ((float)hout[part * lo_slots * 8 + (((size_t)c * g.Hs + y) * g.Ws + x) * 8 + e] != 0.f) {
              set_err(h, "f16x3 conv wrote into the zero border");
              return SN_ERR_DEVICE;
            }
        }
  return SN_OK;
}

int sn_dbg_ref_block_f16x3(sn_handle* h, const float* in, int h_px, int w, const float* w1, const float* b1, const float* w2,
                           const float* b2, int dil, int form, float* out) {
  DevScope ds;      // frees every tracked device buffer on every return path
  if (!h || !in || !w1 || !b1 || !w2 || !b2 || !out || h_px <= 0 || w <= 0) return SN_ERR_ARG;
  if (dil != 1 && dil != 2 && dil != 4 && dil != 8) return SN_ERR_ARG;
  if (form != 0 && form != 1) return SN_ERR_ARG;            // 0 = two k_ref_conv_f16x3 launches, 1 = the streamed block
  if (form == 1 && !stream_x3_supports(dil)) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  const RefGeom g = make_ref_geom(h_px, w);
  const size_t lo_slots = ref16_slots(g, 1) + ref_slack(g), slots = 2 * lo_slots;      // [hi | slack | lo | slack]
  auto idx = [&](int c, int y, int x) { return ((((size_t)(c >> 3)) * g.Hs + y + kRefPad) * g.Ws + x + kRefPad) * 8 + (c & 7); };
  std::vector<_Float16> hin(slots * 8, (_Float16)0.f);
  for (int c = 0; c < kC; ++c)
    for (int y = 0; y < h_px; ++y)
      for (int x = 0; x < w; ++x) {
        const float v = in[((size_t)c * h_px + y) * w + x];
        const _Float16 hi = (_Float16)v;
        hin[idx(c, y, x)] = hi;
        hin[lo_slots * 8 + idx(c, y, x)] = (_Float16)((v - (float)hi) * kSplitScale);
      }
  RefLayerF16 L1, L2;
  if ((rc = upload_ref_f16x3(h, HostLayer{w1, b1, kC, kC, 9}, &L1))) return rc;
  ds.track(L1.bias); ds.track(L1.wfrag);
  if ((rc = upload_ref_f16x3(h, HostLayer{w2, b2, kC, kC, 9}, &L2))) return rc;
  ds.track(L2.bias); ds.track(L2.wfrag);
  uint4 *da = nullptr, *db = nullptr, *da_raw = nullptr, *db_raw = nullptr;
  HIP_TRY(h, alloc_ref16(g, slots, &da_raw, &da));
  ds.track(da_raw);
  HIP_TRY(h, alloc_ref16(g, slots, &db_raw, &db));
  ds.track(db_raw);
  HIP_TRY(h, hipMemcpy(da, hin.data(), slots * 16, hipMemcpyHostToDevice));
  uint4 *cur = da, *oth = db;
  HIP_TRY(h, ref_block_f16x3(h->stream, L1, L2, g, h->num_cu, dil, &cur, &oth, lo_slots, 1, form == 1));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<_Float16> hout(slots * 8);
  HIP_TRY(h, hipMemcpy(hout.data(), cur, slots * 16, hipMemcpyDeviceToHost));
  for (int c = 0; c < kC; ++c)
    for (int y = 0; y < h_px; ++y)
      for (int x = 0; x < w; ++x)
        out[((size_t)c * h_px + y) * w + x] = (float)hout[idx(c, y, x)] + (float)hout[lo_slots * 8 + idx(c, y, x)] * kSplitInv;
  for (int part = 0; part < 2; ++part)       // both zero borders must have survived
    for (int c = 0; c < 4; ++c)
      for (int y = 0; y < g.Hs; ++y)
        for (int x = 0; x < g.Ws; ++x) {
          if (y >= kRefPad && y < kRefPad + h_px && x >= kRefPad && x < kRefPad + w) continue;
          for (int e = 0; e < 8; ++e)
            if ((float)hout[part * lo_slots * 8 + (((size_t)c * g.Hs + y) * g.Ws + x) * 8 + e] != 0.f) {
              set_err(h, "f16x3 residual block wrote into the zero border");
              return SN_ERR_DEVICE;
            }
        }
  return SN_OK;
}

int sn_dbg_ref_block_f16(sn_handle* h, const float* in, int h_px, int w, const float* w1, const float* b1,
                         const float* w2, const float* b2, int dil, float* out) {
  DevScope ds;      // frees every tracked device buffer on every return path
  if (!h || !in || !w1 || !b1 || !w2 || !b2 || !out || h_px <= 0 || w <= 0) return SN_ERR_ARG;
  // tests: bits 8.. select the form: 0 = two launches, 2 = row-streaming fused kernel (1 was the tile-fused kernel of round 2)
  const int form = dil >> 8;
  if (form != 0 && form != 2) return SN_ERR_ARG;
  const int fuse_mode = form == 2 ? 4 : 0;
  dil &= 0xff;
  if (dil != 1 && dil != 2 && dil != 4 && dil != 8) return SN_ERR_ARG;
  if (fuse_mode == 4 && !stream_block_supports(dil)) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  const RefGeom g = make_ref_geom(h_px, w);
  const size_t slots = ref16_slots(g, 1);
  auto idx = [&](int c, int y, int x) { return ((((size_t)(c >> 3)) * g.Hs + y + kRefPad) * g.Ws + x + kRefPad) * 8 + (c & 7); };
  std::vector<_Float16> hin(slots * 8, (_Float16)0.f);
  for (int c = 0; c < kC; ++c)
    for (int y = 0; y < h_px; ++y)
      for (int x = 0; x < w; ++x) hin[idx(c, y, x)] = (_Float16)in[((size_t)c * h_px + y) * w + x];
  RefLayerF16 L1, L2;
  if ((rc = upload_ref_f16(h, HostLayer{w1, b1, kC, kC, 9}, &L1))) return rc;
  ds.track(L1.bias); ds.track(L1.wfrag);
  if ((rc = upload_ref_f16(h, HostLayer{w2, b2, kC, kC, 9}, &L2))) return rc;
  ds.track(L2.bias); ds.track(L2.wfrag);
  uint4 *da = nullptr, *db = nullptr, *da_raw = nullptr, *db_raw = nullptr;
  HIP_TRY(h, alloc_ref16(g, slots + ref_slack(g), &da_raw, &da));
  ds.track(da_raw);
  HIP_TRY(h, alloc_ref16(g, slots + ref_slack(g), &db_raw, &db));
  ds.track(db_raw);
  HIP_TRY(h, hipMemcpy(da, hin.data(), slots * 16, hipMemcpyHostToDevice));
  uint4 *cur = da, *oth = db;
  if (!h->ws.tile_ctr) {
    set_err(h, "sn_dbg_ref_block_f16 needs an engine created in an fp16 mode");
    return SN_ERR_ARG;
  }
  HIP_TRY(h, hipMemsetAsync(h->ws.tile_ctr, 0, kTileCtrBytes, h->stream));
  HIP_TRY(h, ref_block_f16(h->stream, L1, L2, g, h->num_cu, dil, &cur, &oth, 1, h->ws.tile_ctr, fuse_mode, h->dump));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<_Float16> hout(slots * 8);
  HIP_TRY(h, hipMemcpy(hout.data(), cur, slots * 16, hipMemcpyDeviceToHost));
  for (int c = 0; c < kC; ++c)
    for (int y = 0; y < h_px; ++y)
      for (int x = 0; x < w; ++x) out[((size_t)c * h_px + y) * w + x] = (float)hout[idx(c, y, x)];
  for (int c = 0; c < 4; ++c)        // the zero border of the result tensor must have survived
    for (int y = 0; y < g.Hs; ++y)
      for (int x = 0; x < g.Ws; ++x) {
        if (y >= kRefPad && y < kRefPad + h_px && x >= kRefPad && x < kRefPad + w) continue;
        for (int e = 0; e < 8; ++e)
          if ((float)hout[(((size_t)c * g.Hs + y) * g.Ws + x) * 8 + e] != 0.f) {
            set_err(h, "fp16 residual block wrote into the zero border");
            return SN_ERR_DEVICE;
          }
      }
  return SN_OK;
}

int sn_dbg_ref_tail_f16(sn_handle* h, int n, const float* in, int hk, int wk, const float* w1, const float* b1, const float* w2,
                        const float* b2, const float* head_w, float head_b, const float* low, int ups, float dnorm, int h_out,
                        int w_out, int form, float* out_disp, int32_t* out_raw) {
  DevScope ds;
  if (!h || !in || !w1 || !b1 || !w2 || !b2 || !head_w || !low || !out_disp || !out_raw) return SN_ERR_ARG;
  if (n <= 0 || hk <= 0 || wk <= 0 || h_out <= 0 || w_out <= 0 || h_out > hk || w_out > wk || (ups != 16 && ups != 2) ||
      hk % ups || wk % ups || (form != 0 && form != 1) || !(dnorm > 0.f))
    return SN_ERR_ARG;
  if (h->precision != SN_PREC_F16 && h->precision != SN_PREC_AUTO) {
    set_err(h, "sn_dbg_ref_tail_f16 needs an engine created with SN_PREC_F16 or SN_PREC_AUTO");
    return SN_ERR_ARG;
  }
  int rc = check_device(h);
  if (rc) return rc;
  const RefGeom g = make_ref_geom(hk, wk);
  const size_t per = ref16_slots(g, 1), slots = ref16_slots(g, n);
  if ((slots + ref_slack(g) + ref_front(g)) * 16 >= ((size_t)1 << 32)) return SN_ERR_ARG;      // 32-bit byte offsets inside a tensor
  std::vector<_Float16> hin(slots * 8, (_Float16)0.f);
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < kC; ++c)
      for (int y = 0; y < hk; ++y) {
        const float* src = in + (((size_t)i * kC + c) * hk + y) * wk;
        _Float16* dst = &hin[(i * per + (((size_t)(c >> 3)) * g.Hs + y + kRefPad) * g.Ws + kRefPad) * 8 + (c & 7)];
        for (int x = 0; x < wk; ++x) dst[(size_t)x * 8] = (_Float16)src[x];
      }
  RefLayerF16 L1, L2;
  HeadLayer hd;
  // (tracked before the status is looked at: an upload that fails half-way has allocated its first buffer)
  rc = upload_ref_f16(h, HostLayer{w1, b1, kC, kC, 9}, &L1);
  ds.track(L1.bias); ds.track(L1.wfrag);
  if (rc) return rc;
  rc = upload_ref_f16(h, HostLayer{w2, b2, kC, kC, 9}, &L2);
  ds.track(L2.bias); ds.track(L2.wfrag);
  if (rc) return rc;
  rc = upload_head(h, HostLayer{head_w, &head_b, 1, kC, 9}, &hd);
  ds.track(hd.w);
  if (rc) return rc;
  uint4 *da = nullptr, *db = nullptr, *da_raw = nullptr, *db_raw = nullptr;
  HIP_TRY(h, alloc_ref16(g, slots + ref_slack(g), &da_raw, &da));
  ds.track(da_raw);
  HIP_TRY(h, alloc_ref16(g, slots + ref_slack(g), &db_raw, &db));
  ds.track(db_raw);
  HIP_TRY(h, hipMemcpy(da, hin.data(), slots * 16, hipMemcpyHostToDevice));
  const int sh = hk / ups, sw = wk / ups;
  const size_t nlow = (size_t)n * sh * sw, nout = (size_t)n * h_out * w_out;
  float *dlow = nullptr, *dd = nullptr;
  int32_t* dr = nullptr;
  HIP_TRY(h, dalloc(&dlow, nlow));
  ds.track(dlow);
  HIP_TRY(h, dalloc(&dd, nout));
  ds.track(dd);
  HIP_TRY(h, dalloc(&dr, nout));
  ds.track(dr);
  HIP_TRY(h, hipMemcpy(dlow, low, nlow * 4, hipMemcpyHostToDevice));
  HIP_TRY(h, memset_now(dd, 0xff, nout * 4));        // NaN / -1: a pixel the kernel does not write shows up
  HIP_TRY(h, memset_now(dr, 0xff, nout * 4));
  const float inv_q = (float)(1.0 / (kWireFactor * (double)kOutScale));
  const UpScale us{1.0f / (float)ups, (float)ups};
  if (form == 1) {
    StreamHeadArgs ha{hd.w, dlow, dd, dr, hd.bias, dnorm, inv_q, sh, sw, h_out, w_out, us};
    HIP_TRY(h, ref_block_stream_tail(h->stream, L1, L2, g, h->num_cu, da, n, h->dump, ha));
  } else {
    HIP_TRY(h, ref_block_stream(h->stream, L1, L2, g, h->num_cu, 1, da, db, n, h->dump));
    HIP_TRY(h, launch_head_final_f16(h->stream, false, db, 0, g, hd.w, hd.bias, dlow, sh, sw, h_out, w_out, dnorm, inv_q, us, dd, dr, n));
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(out_disp, dd, nout * 4, hipMemcpyDeviceToHost));
  HIP_TRY(h, hipMemcpy(out_raw, dr, nout * 4, hipMemcpyDeviceToHost));
  return SN_OK;
}

// Parse's arithmetic per element (parser.cpp:84-86): the product f * B is a float, everything after it is double
__global__ __launch_bounds__(256) void k_depth_from_raw(const int32_t* __restrict__ raw, size_t n, float scale, float fB,
                                                        float* __restrict__ depth, float* __restrict__ disp) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float dis = (float)raw[i] * scale;
    depth[i] = (float)((double)fB / ((double)dis * 16.0 * 12.0) / 1000.0);
    if (disp) disp[i] = dis * 16.0f * 12.0f;
  }
}

int sn_depth_from_raw(sn_handle* h, int n, const int32_t* raw, float focal_px, float baseline_mm, float* depth_m, float* disp_px,
                      int mem, void* stream) {
  if (!h || !raw || !depth_m || n <= 0 || n > h->max_batch || (mem != SN_MEM_HOST && mem != SN_MEM_DEVICE)) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  const size_t cnt = (size_t)n * h->H * h->W;
  hipStream_t st = stream ? static_cast<hipStream_t>(stream) : h->stream;
  const float fB = focal_px * baseline_mm;      // float product, as in the reference expression
  const int32_t* draw = raw;
  float *ddepth = depth_m, *ddisp = disp_px;
  DevScope ds;
  if (mem == SN_MEM_HOST) {
    int32_t* a = nullptr;
    float *b = nullptr, *c = nullptr;
    HIP_TRY(h, dalloc(&a, cnt));
    ds.track(a);
    HIP_TRY(h, dalloc(&b, cnt));
    ds.track(b);
    if (disp_px) {
      HIP_TRY(h, dalloc(&c, cnt));
      ds.track(c);
    }
    HIP_TRY(h, hipMemcpyAsync(a, raw, cnt * 4, hipMemcpyHostToDevice, st));
    draw = a;
    ddepth = b;
    ddisp = c;
  }
  unsigned grid = (unsigned)((cnt + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_depth_from_raw, dim3(grid), dim3(256), 0, st, draw, cnt, kOutScale, fB, ddepth, ddisp);
  HIP_TRY(h, hipGetLastError());
  if (mem == SN_MEM_HOST) {
    HIP_TRY(h, hipMemcpyAsync(depth_m, ddepth, cnt * 4, hipMemcpyDeviceToHost, st));
    if (disp_px) HIP_TRY(h, hipMemcpyAsync(disp_px, ddisp, cnt * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
  }
  return SN_OK;
}

__global__ __launch_bounds__(256) void k_copy_limited(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int sn_dbg_copy_limited(void* dst, const void* src, size_t bytes, int workgroups, void* stream) {
  if (!dst || !src || (bytes & 15) || workgroups <= 0 || workgroups > 65535 || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return SN_ERR_ARG;
  hipLaunchKernelGGL(k_copy_limited, dim3((unsigned)workgroups), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<uint4*>(dst), static_cast<const uint4*>(src), bytes / 16);
  return hipGetLastError() == hipSuccess ? SN_OK : SN_ERR_DEVICE;
}

int sn_dbg_read(sn_handle* h, const char* what, float* dst, size_t cap, size_t* n) {
  if (!h || !what || !n) return SN_ERR_ARG;
  int rc = check_device(h);
  if (rc) return rc;
  const size_t hw = (size_t)h->hl * h->wl;
  const float* src = nullptr;
  size_t cnt = 0;
  if (!strcmp(what, "stream_prio")) {      // host state: 1 = the pipeline streams were created with the highest priority
    *n = 1;
    if (dst && cap >= 1) dst[0] = h->stream_prio ? 1.f : 0.f;
    return (dst && cap < 1) ? SN_ERR_ARG : SN_OK;
  }
  if (!strcmp(what, "feat_l")) { src = h->ws.feat; cnt = kC * hw; }
  else if (!strcmp(what, "feat_r")) { src = h->ws.feat + kC * hw; cnt = kC * hw; }
  else if (!strcmp(what, "cost")) { src = h->ws.cost; cnt = h->Dl * hw; }
  else if (!strcmp(what, "disp_low")) { src = h->ws.disp_low; cnt = hw; }
  else if (!strcmp(what, "tile_ctr") && h->ws.tile_ctr) { src = reinterpret_cast<const float*>(h->ws.tile_ctr); cnt = kTileCtrBytes / 4 * h->ws.n_chunks; }
  else if (!strcmp(what, "refine_x") && h->precision == SN_PREC_FP32) { src = h->ws.ref[0]; cnt = (size_t)kC * h->Hp * h->Wp; }
  else if (!strncmp(what, "level", 5) && what[5] >= '1' && what[5] < '0' + h->levels && what[6] == 0) {
    // hierarchical refinement: the map of level k (first pair of the last piece)
    const int k = what[5] - '0';
    src = h->ws.lvl_disp[k];
    cnt = (size_t)h->tw[k].Hk * h->tw[k].Wk;
  }
  else return SN_ERR_ARG;
  *n = cnt;
  if (!dst) return SN_OK;
  if (cap < cnt) return SN_ERR_ARG;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(dst, src, cnt * 4, hipMemcpyDeviceToHost));
  return SN_OK;
}

}  // extern "C"
