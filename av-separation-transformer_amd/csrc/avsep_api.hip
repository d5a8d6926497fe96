// avsep_api.hip -- context, weight packer, workspace carving and the forward schedule behind the C ABI
// of include/avsep.h.  Host-side C++ only; every device kernel lives in gemm.hip / attention.hip /
// rowops.hip.  The schedule follows AVSeparationTransformer.forward (model.py:268-276): audio encoder
// and visual encoder are independent until fusion, so they are enqueued on two HIP streams.
#include "../../include/avsep.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
int fail_hip(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return AVSEP_EHIP;
}
// Entry points that touch STL containers are function-try-blocks ending in this: no C++ exception ever crosses the
// C ABI (the caller may be ctypes / cgo / JNI, where an escaping exception is an abort).
int on_exception() noexcept {
  try {
    try { throw; }
    catch (const std::bad_alloc&) { return fail(AVSEP_ENOMEM, "host memory allocation failed"); }
    catch (const std::exception& e) { return fail(AVSEP_EINTERNAL, std::string("internal error: ") + e.what()); }
    catch (...) { return fail(AVSEP_EINTERNAL, "internal error: unknown C++ exception"); }
  } catch (...) {
    return AVSEP_EINTERNAL;   // even recording the message failed
  }
}
#define HCK(x)                                         \
  do {                                                 \
    hipError_t e_ = (x);                               \
    if (e_ != hipSuccess) return fail_hip(e_, #x);     \
  } while (0)
#define RCK(x)                  \
  do {                          \
    int r_ = (x);               \
    if (r_ != AVSEP_OK) return r_; \
  } while (0)

inline size_t align_up(size_t n, size_t a) { return (n + a - 1) / a * a; }
inline int conv_out(int x) { return (x - 1) / 2 + 1; }   // k3 s2 p1 (model.py:82)

struct RawW {
  const float* ptr = nullptr;
  std::vector<int64_t> shape;
};

struct EncLayerW {   // nn.TransformerEncoderLayer (model.py:48-52)
  float *wqkv, *bqkv, *wo, *bo, *w1, *b1, *w2, *b2, *g1, *be1, *g2, *be2;
};
struct FusLayerW {   // CrossAttentionLayer (model.py:152-164); its K/V rows live in wkv_all
  float *wq, *bq, *wo, *bo, *w1, *b1, *w2, *b2, *g1, *be1, *g2, *be2;
};

struct LnxW { float *w, *c1, *c2; };   // operands of the LayerNorm-in-the-epilogue GEMM (kernels.h GemmParams::lnx_c1)

struct Workspace {   // all float*, carved from the caller's buffer
  float *xt, *a_h0, *a_x, *ln, *qkv, *att, *ffn;
  float *act1, *act2, *act3, *pool, *v_x, *v_ln, *v_qkv, *v_att, *v_ffn, *v_up;
  float *kv_all, *f_q;
  float* taps;
  // Pre-split GEMM operands (gemm_planes.hip, round 5): the bf16 planes of the tensors that only GEMMs read -- LayerNorm outputs,
  // attention outputs, FFN hidden activations, the resized visual stream -- written by their producers INSTEAD of the fp32 tensor.
  // rows_a / rows_v: row counts of the audio-length / frame-length buffers (the planes' slab height; a half-batch view keeps them)
  unsigned short *ln_p, *att_p, *ffn_p, *v_ln_p, *v_att_p, *v_ffn_p, *v_up_p;
  float* v_up_rs;            // per-row descale of the resized visual stream's two-term planes (2^-e of each row)
  int* f_clip_exp;           // [B][Lf][2]: per clip and fusion layer, the exponents of the cross-attention's k and v (clip_exp_kernel)
  float* f_att_rs;           // per-row descale of the cross-attention output's two-term planes (2^-ev of the row's clip)
  long long rows_a, rows_v;
  size_t floats;
};

struct GraphEntry {
  const void *mixed, *lips, *masks, *sep, *ws;
  size_t ws_bytes;
  int B, T, N, H, W;
  hipGraphExec_t exec;       // ONE executable, re-launched every step: alternating two executables of the same graph was
                             // measured 25 % SLOWER with two steps in flight (profiles/r03_graph_exec_alternation.txt)
  hipStream_t last_stream;   // where the graph was launched last (drained before the exec is destroyed)
  // Every graph is captured on its OWN pair of streams.  Two graphs captured one after the other on the same pair do
  // not overlap when replayed on different caller streams (measured: 2 steps in flight 0.467 ms/step with shared
  // capture streams, 0.381 with separate contexts, i.e. separate capture streams -- the runtime appears to place the
  // branches of an executable graph by the streams they were captured on).
  hipStream_t cap, side;
};

}  // namespace

struct avsep_ctx {
  avsep_config cfg;
  int F, Fp, d, h, dh, Le, Lf, S;
  std::unordered_map<std::string, RawW> raw;
  float* arena = nullptr;
  size_t arena_floats = 0;
  bool finalized = false;
  bool ok_audio = false, ok_visual = false, ok_fusion = false, ok_decoder = false;
  int pe_len_a = 0, pe_len_v = 0;
  // packed weights
  float *a_w1, *a_b1, *a_w2, *a_b2, *a_pe;
  std::vector<EncLayerW> a_layers, v_layers;
  float *c1_w, *c1_b, *c2_w, *c2_b, *c3_w, *c3_b, *fp_w, *fp_b, *v_pe;
  float *wkv_all, *bkv_all;
  std::vector<FusLayerW> f_layers;
  float *fn_g, *fn_b, *d_w1, *d_b1, *d_w2, *d_b2;
  float* zeros = nullptr;
  std::unordered_map<const float*, LnxW> lnx;     // by the packed weight of the linear layer that follows a LayerNorm
  bool use_lnx = true, lnx_all = false;
  // Split-precision GEMM (gemm_split.hip; round 4): in models with d_model >= 512, every GEMM whose weight has N >= 512 and
  // K >= 512 -- all nn.Linear layers, the two Conv1d of the audio front-end (3-tap A operand) and the mask head -- runs its
  // products as six bf16 MFMAs per fp32 product (operands cut into three bf16 terms, fp32 accumulation): fp32-equivalent
  // results, ~1.6x the fp32 MFMA GEMM's speed.  The rule looks at the MODEL and the WEIGHT's shape only, never at the row
  // count, so every batch size of a model computes the same bits; d_model = 256 models keep the fp32 MFMA everywhere (their
  // GEMMs are latency-bound: 128 x 128 tiles would leave the chip empty).
  bool split_gemm = true;
  int split_min = 512;                             // smallest d_model (and weight N, K) of the split-precision rule
  // ... and with operands PRE-SPLIT (gemm_planes.hip; round 5): the weights' planes are cut once by avsep_finalize_weights (keyed by
  // the packed fp32 weight), the activations' by their producers, whenever a stage has enough rows for the 256 x 128 kernel
  // (planes_rows()).  Same bits as the in-flight split, so the choice may look at the row count.
  bool use_planes = true;
  std::unordered_map<const float*, unsigned short*> wplanes;
  std::vector<std::pair<const float*, std::pair<int, int>>> wplane_sites;   // (weight, (N, K)) in arena order
  // Two fp16 terms, three products (gemm_h2.hip; round 5): the GEMM sites whose A operand has a STATIC bound from the weights alone
  // (LayerNorm outputs, act(LayerNorm(x) W^T + b), self-attention outputs).  Per site: the weight's H2 planes (row n scaled by
  // 2^ew[n]), the A operand's exponent eA, cscale[n] = 2^-(eA + ew[n]).  A rule on the model and the weight, never on the batch.
  struct H2Site { unsigned short* wp; float* cscale; int* ew; float* l2; int eA, N, K; int eq, ek, ev; bool qkv; };   // eq / ek / ev: an in-projection's output bounds (attention_h2_kernel)
  std::unordered_map<const float*, H2Site> h2;
  bool use_h2 = true;
  float* fkv_const = nullptr;    // [Lf][2][2]: per fusion layer, {sqrt(d) max ||w_n||_2, max |b_n|} of its K rows and of its V rows (clip_exp_kernel)
  bool cross_h2 = false;         // the cross-attention and its out-projection on two fp16 terms with per-clip exponents
  // ... and the fused conv stack's conv2 / conv3 (conv_stack_h2_kernel), every model size: weights as H2 planes + row exponents
  ConvH2 conv_h2{};
  unsigned short *c2_h = nullptr, *c3_h = nullptr;
  float *c2_sc = nullptr, *c3_sc = nullptr, *c2_l2 = nullptr, *c3_l2 = nullptr;
  int *c2_ew = nullptr, *c3_ew = nullptr;
  // streams / events for the audio || visual fork-join and graph replay
  int device = 0;                                  // the device the context (arena, streams, events, graphs) lives on
  hipStream_t side = nullptr;                      // eager forwards: the visual branch's stream
  hipEvent_t ev_fork = nullptr, ev_vdone = nullptr, ev_adone = nullptr, ev_tdone = nullptr;
  bool no_fused_conv = false;                      // developer A/B switch (AVSEP_NO_FUSED_CONV)
  bool paired = false;                             // developer experiment: encoder layers of both branches in shared launches
  bool tail_split = true;                          // two-stream schedule: fusion+decoder, half the batch per stream after the join
  std::vector<GraphEntry> graphs;
  // schedule 1 (avsep_set_schedule): the encoder layers of a branch as ONE dependency-driven persistent launch (chain.hip)
  int schedule = 0, chain_group = 8;
  float chain_skew = 0.0f;
  struct ChainEntry { const float* x; int B, L; const void* layers; ChainPlanImpl* plan; };
  std::vector<ChainEntry> chains;
  // live per-kernel profiler (HIP events around every launch, on the launch's own stream)
  struct ProfRec { std::string name; double flops, bytes; hipEvent_t e0, e1; };
  bool prof_on = false;
  std::vector<ProfRec> prof;
  // debug timeline: device wall-clock stamps at stage boundaries (AVSEP_STAMPS=1, avsep_read_stamps)
  unsigned long long* stamps = nullptr;
  // debug taps
  bool keep_taps = false;
  struct Tap { std::string name; size_t off, n; };
  std::vector<Tap> taps;
  size_t tap_cursor = 0;
};

namespace {

constexpr int PE_MAX_LEN = 5000;   // PositionalEncoding(max_len=5000), model.py:286

// ------------------------------------------------------------------------------------------ arena
template <typename F>
void layout_arena(avsep_ctx* c, F&& take) {
  const int d = c->d, Fp = c->Fp, F_ = c->F, S = c->S;
  (void)F_;
  c->a_w1 = take((size_t)d * 3 * Fp);
  c->a_b1 = take(d);
  c->a_w2 = take((size_t)d * 3 * d);
  c->a_b2 = take(d);
  c->a_pe = take((size_t)PE_MAX_LEN * d);
  auto enc = [&](std::vector<EncLayerW>& v) {
    v.resize(c->Le);
    for (auto& L : v) {
      L.wqkv = take((size_t)3 * d * d); L.bqkv = take(3 * d);
      L.wo = take((size_t)d * d); L.bo = take(d);
      L.w1 = take((size_t)4 * d * d); L.b1 = take(4 * d);
      L.w2 = take((size_t)4 * d * d); L.b2 = take(d);
      L.g1 = take(d); L.be1 = take(d); L.g2 = take(d); L.be2 = take(d);
    }
  };
  enc(c->a_layers);
  c->c1_w = take(9 * 32); c->c1_b = take(32);
  c->c2_w = take(64 * 9 * 32); c->c2_b = take(64);
  c->c3_w = take(128 * 9 * 64); c->c3_b = take(128);
  c->c2_h = reinterpret_cast<unsigned short*>(take(64 * 9 * 32)); c->c3_h = reinterpret_cast<unsigned short*>(take(128 * 9 * 64));   // 4 B per weight
  c->c2_sc = take(64); c->c3_sc = take(128); c->c2_l2 = take(64); c->c3_l2 = take(128);
  c->c2_ew = reinterpret_cast<int*>(take(64)); c->c3_ew = reinterpret_cast<int*>(take(128));
  c->fp_w = take((size_t)d * 128); c->fp_b = take(d);
  c->v_pe = take((size_t)PE_MAX_LEN * d);
  enc(c->v_layers);
  c->wkv_all = take((size_t)c->Lf * 2 * d * d);
  c->bkv_all = take((size_t)c->Lf * 2 * d);
  c->f_layers.resize(c->Lf);
  for (auto& L : c->f_layers) {
    L.wq = take((size_t)d * d); L.bq = take(d);
    L.wo = take((size_t)d * d); L.bo = take(d);
    L.w1 = take((size_t)4 * d * d); L.b1 = take(4 * d);
    L.w2 = take((size_t)4 * d * d); L.b2 = take(d);
    L.g1 = take(d); L.be1 = take(d); L.g2 = take(d); L.be2 = take(d);
  }
  c->fn_g = take(d); c->fn_b = take(d);
  c->d_w1 = take((size_t)2 * d * d); c->d_b1 = take(2 * d);
  c->d_w2 = take((size_t)S * c->F * 2 * d); c->d_b2 = take((size_t)S * c->F);
  c->zeros = take((size_t)std::max(c->Fp, c->d) + 64);   // GemmParams::zeros (zeroed by avsep_create)
  // planes of the weights the split-precision rule sends to the bf16 pipe (d_model >= 512: N >= 512 and K >= 512), PLAIN A operand
  c->wplanes.clear();
  c->wplane_sites.clear();
  if (d >= c->split_min) {
    const int smin = c->split_min;
    auto wp = [&](const float* w, int n, int k) {
      if (n < smin || k < smin || (k & 31)) return;
      float* q = take(((size_t)n * k * 3 + 1) / 2);                  // 6 bytes per weight
      if (w) {
        c->wplanes[w] = reinterpret_cast<unsigned short*>(q);
        c->wplane_sites.push_back({w, {n, k}});
      }
    };
    for (auto* v : {&c->a_layers, &c->v_layers})
      for (auto& L : *v) { wp(L.wqkv, 3 * d, d); wp(L.wo, d, d); wp(L.w1, 4 * d, d); wp(L.w2, d, 4 * d); }
    wp(c->wkv_all, c->Lf * 2 * d, d);
    for (auto& L : c->f_layers) { wp(L.wq, d, d); wp(L.wo, d, d); wp(L.w1, 4 * d, d); wp(L.w2, d, 4 * d); }
    wp(c->d_w1, 2 * d, d);
    wp(c->d_w2, S * c->F, 2 * d);
    c->h2.clear();
    auto h2 = [&](const float* w, int n, int k) {
      if (n < smin || k < smin || (k & 31)) return;
      avsep_ctx::H2Site t{};
      t.wp = reinterpret_cast<unsigned short*>(take((size_t)n * k));           // 4 bytes per weight
      t.cscale = take(n);
      t.ew = reinterpret_cast<int*>(take(n));
      t.l2 = take(n);
      t.N = n; t.K = k;
      if (w) c->h2[w] = t;
    };
    for (auto* v : {&c->a_layers, &c->v_layers})
      for (auto& L : *v) { h2(L.wqkv, 3 * d, d); h2(L.wo, d, d); h2(L.w1, 4 * d, d); h2(L.w2, d, 4 * d); }
    for (auto& L : c->f_layers) { h2(L.wq, d, d); h2(L.wo, d, d); h2(L.w1, 4 * d, d); h2(L.w2, d, 4 * d); }   // wo: its operand carries one power of two per CLIP (eA stays 0)
    c->fkv_const = take((size_t)std::max(c->Lf, 1) * 4);
    h2(c->wkv_all, c->Lf * 2 * d, d);   // its input (the resized visual stream) carries one power of two per ROW (eA stays 0)
    h2(c->d_w1, 2 * d, d);
    h2(c->d_w2, S * c->F, 2 * d);
  }
  c->lnx.clear();
  if (c->use_lnx) {
    auto site = [&](const float* w, int n) {
      LnxW x;
      x.w = take((size_t)n * d); x.c1 = take(n); x.c2 = take(n);
      if (w) c->lnx[w] = x;
    };
    for (auto* v : {&c->a_layers, &c->v_layers})
      for (auto& L : *v) { site(L.wqkv, 3 * d); site(L.w1, 4 * d); }
    for (auto& L : c->f_layers) { site(L.wq, d); site(L.w1, 4 * d); }
    site(c->d_w1, 2 * d);
  }
}

// ------------------------------------------------------------------------------------------ raw weights
const RawW* find(const avsep_ctx* c, const std::string& key) {
  auto it = c->raw.find(key);
  return it == c->raw.end() ? nullptr : &it->second;
}
bool has_shape(const RawW* w, std::initializer_list<int64_t> shp) {
  if (!w || w->shape.size() != shp.size()) return false;
  size_t i = 0;
  for (int64_t s : shp)
    if (w->shape[i++] != s) return false;
  return true;
}

__global__ void pe_fill_kernel(float* pe, int max_len, int d) {
  // fallback when the caller supplied no `pe` buffer: same formula as model.py:289-297
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)max_len * d) return;
  const int p = (int)(idx / d), j = (int)(idx % d);
  const float div = expf((float)(j & ~1) * (-logf(10000.0f) / (float)d));
  const float a = (float)p * div;
  pe[idx] = (j & 1) ? cosf(a) : sinf(a);
}

struct Packer {
  avsep_ctx* c;
  hipStream_t s;
  hipError_t err = hipSuccess;
  bool missing = false;
  std::string first_missing;

  const float* need(const std::string& key, std::initializer_list<int64_t> shp) {
    const RawW* w = find(c, key);
    if (!has_shape(w, shp)) {
      if (!missing) first_missing = key;
      missing = true;
      return nullptr;
    }
    return w->ptr;
  }
  void keep(hipError_t e) {
    if (err == hipSuccess && e != hipSuccess) err = e;
  }
  void copy(const std::string& key, float* dst, int64_t n) {
    const float* p = need(key, {n});
    if (p) keep(launch_scale_copy(p, dst, (int)n, 1.0f, 0, s));
  }
  void rows(const std::string& key, float* dst, int64_t r, int64_t k) {
    const float* p = need(key, {r, k});
    if (p) keep(launch_pack_rows(p, dst, (int)r, (int)k, (int)k, 1.0f, 0, s));
  }
};

bool pack_encoder(Packer& P, const std::string& pre, std::vector<EncLayerW>& layers) {
  avsep_ctx* c = P.c;
  const int d = c->d;
  const float qs = 1.0f / std::sqrt((float)c->dh);
  for (size_t i = 0; i < layers.size(); ++i) {
    const std::string p = pre + "transformer.layers." + std::to_string(i) + ".";
    EncLayerW& L = layers[i];
    // packed in_proj rows = [Wq;Wk;Wv]; fold 1/sqrt(dh) into the q rows and q bias
    if (const float* w = P.need(p + "self_attn.in_proj_weight", {3 * d, d}))
      P.keep(launch_pack_rows(w, L.wqkv, 3 * d, d, d, qs, d, P.s));
    if (const float* b = P.need(p + "self_attn.in_proj_bias", {3 * d}))
      P.keep(launch_scale_copy(b, L.bqkv, 3 * d, qs, d, P.s));
    P.rows(p + "self_attn.out_proj.weight", L.wo, d, d);
    P.copy(p + "self_attn.out_proj.bias", L.bo, d);
    P.rows(p + "linear1.weight", L.w1, 4 * d, d);
    P.copy(p + "linear1.bias", L.b1, 4 * d);
    P.rows(p + "linear2.weight", L.w2, d, 4 * d);
    P.copy(p + "linear2.bias", L.b2, d);
    P.copy(p + "norm1.weight", L.g1, d);
    P.copy(p + "norm1.bias", L.be1, d);
    P.copy(p + "norm2.weight", L.g2, d);
    P.copy(p + "norm2.bias", L.be2, d);
  }
  return !P.missing;
}

int pack_pe(Packer& P, const std::string& key, float* dst, int* len_out) {
  avsep_ctx* c = P.c;
  const RawW* w = find(c, key);
  if (w) {
    if (w->shape.size() != 3 || w->shape[0] != 1 || w->shape[2] != c->d || w->shape[1] > PE_MAX_LEN)
      return fail(AVSEP_EINVAL, key + ": expected shape (1, <=5000, d_model)");
    *len_out = (int)w->shape[1];
    P.keep(launch_scale_copy(w->ptr, dst, (int)(w->shape[1] * c->d), 1.0f, 0, P.s));
  } else {
    *len_out = PE_MAX_LEN;
    const size_t n = (size_t)PE_MAX_LEN * c->d;
    hipLaunchKernelGGL(pe_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, P.s, dst, PE_MAX_LEN, c->d);
    P.keep(hipGetLastError());
  }
  return AVSEP_OK;
}

// ------------------------------------------------------------------------------------------ workspace
size_t carve(const avsep_ctx* c, Workspace* w, float* base, int B, int T, int N, int H, int W) {
  size_t off = 0;
  auto take = [&](size_t n) {
    float* p = base ? base + off : nullptr;
    off += align_up(n ? n : 1, 64);
    return p;
  };
  const size_t Ma = (size_t)B * T, Mv = (size_t)B * N;
  const int d = c->d;
  const int H1 = conv_out(H), W1 = conv_out(W), H2 = conv_out(H1), W2 = conv_out(W1), H3 = conv_out(H2),
            W3 = conv_out(W2);
  Workspace t{};
  t.xt = take(Ma * c->Fp);
  t.a_h0 = take(Ma * d);
  t.a_x = take(Ma * d);
  t.ln = take(Ma * d);
  t.qkv = take(Ma * 3 * d);
  t.att = take(Ma * d);
  t.ffn = take(Ma * 4 * d);
  t.act1 = take(Mv * H1 * W1 * 32);
  t.act2 = take(Mv * H2 * W2 * 64);
  t.act3 = take(Mv * H3 * W3 * 128);
  t.pool = take(Mv * 128);
  t.v_x = take(Mv * d);
  t.v_ln = take(Mv * d);
  t.v_qkv = take(Mv * 3 * d);
  t.v_att = take(Mv * d);
  t.v_ffn = take(Mv * 4 * d);
  t.v_up = take(Ma * d);
  t.kv_all = take(Ma * (size_t)c->Lf * 2 * d);
  t.f_q = take(Ma * d);
  if (d >= c->split_min && !(d & 31)) {                                           // 6 bytes per element = 1.5 floats
    auto planes = [&](size_t n) { return reinterpret_cast<unsigned short*>(take((n * 3 + 1) / 2)); };
    t.ln_p = planes(Ma * d); t.att_p = planes(Ma * d); t.ffn_p = planes(Ma * 4 * d);
    t.v_ln_p = planes(Mv * d); t.v_att_p = planes(Mv * d); t.v_ffn_p = planes(Mv * 4 * d);
    t.v_up_p = planes(Ma * d);
    t.v_up_rs = take(Ma);
    t.f_clip_exp = reinterpret_cast<int*>(take((size_t)std::max(B, 1) * std::max(c->Lf, 1) * 2));
    t.f_att_rs = take(Ma);
    if (!base) t.ln_p = t.att_p = t.ffn_p = t.v_ln_p = t.v_att_p = t.v_ffn_p = t.v_up_p = nullptr;
  }
  t.rows_a = (long long)Ma;
  t.rows_v = (long long)Mv;
  if (c->keep_taps) {
    // generous bound: every tap is at most one (M, d)-sized activation or one conv activation
    size_t n = (size_t)(2 + 2 * c->Le + c->Lf + 3) * align_up(Ma * d > Mv * d ? Ma * d : Mv * d, 64);
    n += align_up(Mv * H1 * W1 * 32, 64) + align_up(Mv * H2 * W2 * 64, 64) + align_up(Mv * H3 * W3 * 128, 64) +
         align_up(Mv * 128, 64);
    t.taps = take(n);
  }
  t.floats = off;
  if (w) *w = t;
  return off;
}

int record_tap(avsep_ctx* c, const Workspace& w, const char* name, const float* src, size_t n, hipStream_t s) {
  if (!c->keep_taps) return AVSEP_OK;
  HCK(hipMemcpyAsync(w.taps + c->tap_cursor, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  c->taps.push_back({name, c->tap_cursor, n});
  c->tap_cursor += align_up(n, 64);
  return AVSEP_OK;
}

// ------------------------------------------------------------------------------------------ profiled launches
// Every kernel of the forward goes through one of these wrappers.  With the profiler off they are plain
// launches; with it on, each launch is bracketed by two HIP events recorded on the stream it runs on, and
// carries its ALGORITHMIC flops/bytes (DESIGN.md "roofline accounting"), so bench.py can price every kernel
// live, on hardware, without an external profiler.
// HIP event records cost several microseconds each on this stack, so a launch is not timed alone: with the
// profiler on it is issued PROF_REPS times back to back between ONE pair of events and the record keeps the
// mean.  (In-place residual updates are applied PROF_REPS times, so a profiled forward's OUTPUT is
// meaningless -- it is a timing run only.)
constexpr int PROF_REPS = 20;

template <typename L>
int profiled(avsep_ctx* c, const char* name, double flops, double bytes, hipStream_t s, L&& launch) {
  if (!c->prof_on) {
    HCK(launch());
    return AVSEP_OK;
  }
  avsep_ctx::ProfRec r{name, flops, bytes, nullptr, nullptr};
  HCK(hipEventCreate(&r.e0));
  HCK(hipEventCreate(&r.e1));
  HCK(hipEventRecord(r.e0, s));
  hipError_t e = hipSuccess;
  for (int i = 0; i < PROF_REPS && e == hipSuccess; ++i) e = launch();
  HCK(hipEventRecord(r.e1, s));
  c->prof.push_back(r);
  HCK(e);
  return AVSEP_OK;
}

// kalg: algorithmic K (un-padded) for the flop count
int run_gemm(avsep_ctx* c, const GemmParams& p, hipStream_t s, int kalg = 0) {
  const double k = kalg > 0 ? kalg : p.K;
  const double rows = (double)p.M + (p.alt.M > 0 ? p.alt.M : 0);          // both problems of a pair launch
  const double flops = 2.0 * rows * p.N * k;
  double a_bytes = rows * k * 4;
  if (p.amode == AMODE_TAPS3) a_bytes /= 3;                      // each input row feeds 3 taps
  if (p.amode == AMODE_CONV2D) a_bytes = a_bytes / 9 * 4;        // stride-2 3x3: each input pixel read once
  double bytes = a_bytes + (double)p.N * k * 4 * (p.alt.M > 0 ? 2 : 1) + rows * p.N * 4 * (p.C2 ? 2 : 1);
  if (p.R) bytes += (p.rperiod > 0 ? (double)p.rperiod : rows) * p.N * 4;
  if (p.C2) bytes += (double)p.M * p.F * 4;
  static const bool no_taps = dev_env("AVSEP_SPLIT_NO_TAPS") != nullptr, no_mask = dev_env("AVSEP_SPLIT_NO_MASK") != nullptr;   // developer A/B
  if (p.h2) {                                                       // two fp16 terms, three products (gemm_h2.hip): the caller asked h2_site() first
    GemmParams pp = p;
    auto it = c->h2.find(p.W);
    if (it == c->h2.end() || !c->use_h2) return fail(AVSEP_EINTERNAL, "two-term GEMM: not a site of it");
    pp.Wp = it->second.wp; pp.w_rows = p.N; pp.cscale = it->second.cscale;
    pp.split_t2_min = 48;
    if (!gemm_h2_supported(pp)) return fail(AVSEP_EINTERNAL, "two-term GEMM: unsupported problem");
    bytes = rows * k * 4 + (double)p.N * k * 4 + (p.C ? rows * p.N * 4 * (p.C2 ? 2 : 1) : 0.0) + (p.Cp ? rows * p.N * 4 : 0.0) +
            (p.R ? rows * p.N * 4 : 0.0) + (p.C2 ? (double)p.M * p.F * 4 : 0.0);
    return profiled(c, gemm_h2_instance_name(pp), flops, bytes, s, [&] { return launch_gemm_h2(pp, s); });
  }
  if (p.Ap || p.Cp) {                                               // pre-split operands: the caller asked planes_rows() first
    GemmParams pp = p;
    auto it = c->wplanes.find(p.W);
    if (it == c->wplanes.end()) return fail(AVSEP_EINTERNAL, "pre-split GEMM: the weight has no planes");
    pp.Wp = it->second; pp.w_rows = p.N;
    if (!gemm_planes_supported(pp)) return fail(AVSEP_EINTERNAL, "pre-split GEMM: unsupported problem");
    bytes += (p.Ap ? rows * k * 2 : 0.0) + (double)p.N * k * 2 + (p.Cp ? rows * p.N * (p.C ? 6.0 : 2.0) : 0.0);   // 6 bytes per plane element
    return profiled(c, gemm_planes_instance_name(), flops, bytes, s, [&] { return launch_gemm_planes(pp, s); });
  }
  if (c->split_gemm && c->d >= c->split_min && p.N >= c->split_min && p.K >= c->split_min && gemm_split_supported(p) &&       // see avsep_ctx::split_gemm
      !(no_taps && p.amode == AMODE_TAPS3) && !(no_mask && p.C2)) {
    GemmParams ps = p;
    ps.split_t2_min = 48;                                          // several forwards in flight: see gemm_split.hip, "Which kernel"
    return profiled(c, gemm_split_instance_name(ps), flops, bytes, s, [&] { return launch_gemm_split(ps, s); });
  }
  return profiled(c, c->prof_on ? gemm_instance_name(p) : "", flops, bytes, s, [&] { return launch_gemm(p, s); });
}
// Does a stage with M rows run its GEMMs on PRE-SPLIT operands (gemm_planes.hip)?  The planes kernel has the 256 x 128 tile only:
// from 48 of its tiles on at the narrowest weight (N = 512: 4 column tiles) -- the threshold run_gemm gives the in-flight 256 x 128
// kernel (split_t2_min); below it the 64 x 64 in-flight kernel fills the chip better.  Both compute the same bits, so the choice
// may look at the row count.  Debug taps read fp32 tensors: a tapped forward takes the in-flight path.
bool planes_rows(const avsep_ctx* c, const Workspace& w, int M) {
  static const bool off = dev_env("AVSEP_NO_PLANES") != nullptr;                                    // developer A/B
  return c->use_planes && !off && c->split_gemm && c->d >= c->split_min && !(c->d & 31) && w.ln_p && !c->keep_taps && !c->wplanes.empty() &&
         (long)((M + 255) / 256) * 4 >= 48;
}
// The two-term fp16 site of weight W (gemm_h2.hip), or null: the model / the weight is not on that path.  Never looks at a row count.
const avsep_ctx::H2Site* h2_site(const avsep_ctx* c, const Workspace& w, const float* W) {
  if (!c->use_h2 || !c->split_gemm || c->d < c->split_min || !w.ln_p) return nullptr;
  auto it = c->h2.find(W);
  return it == c->h2.end() ? nullptr : &it->second;
}
int run_layernorm_h2(avsep_ctx* c, const float* x, const float* g, const float* b, unsigned short* yp, long long rows, int M, int d,
                     int e, hipStream_t s) {
  return profiled(c, layernorm_planes_instance_name(d, 2), 8.0 * M * d, 2.0 * M * d * 4, s,
                  [&] { return launch_layernorm_h2(x, g, b, yp, rows, M, d, 1e-5f, e, s); });
}
int run_layernorm_planes(avsep_ctx* c, const float* x, const float* g, const float* b, unsigned short* yp, long long rows, int M, int d,
                         hipStream_t s) {
  return profiled(c, layernorm_planes_instance_name(d, 1), 8.0 * M * d, 2.5 * M * d * 4, s,
                  [&] { return launch_layernorm_planes(x, g, b, yp, rows, M, d, 1e-5f, s); });
}
int run_layernorm(avsep_ctx* c, const float* x, const float* g, const float* b, float* y, int M, int d, hipStream_t s) {
  return profiled(c, layernorm_instance_name(d, false), 8.0 * M * d, 2.0 * M * d * 4, s,
                  [&] { return launch_layernorm(x, g, b, y, M, d, 1e-5f, s); });
}
bool attention_is_split(const avsep_ctx* c, int Lq, int Lk);
// op (o_rows): the output as the planes of the out-projection's A operand instead of fp32 `o` (only where attention_is_split())
int run_attention(avsep_ctx* c, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o,
                  int ldo, int B, int Lq, int Lk, hipStream_t s, unsigned short* op = nullptr, long long o_rows = 0, int h2 = 0,
                  int h2_exp = 0, const avsep_ctx::H2Site* qkv = nullptr) {
  const double flops = 4.0 * B * c->h * (double)Lq * Lk * c->dh;
  const double bytes = 4.0 * B * c->d * (2.0 * Lq + 2.0 * Lk);
  // two fp16 terms where the operands have static bounds (self-attention behind the in-projection of a two-term site): the rule looks
  // at the model and the sequence lengths only
  static const bool no_h2_attn = dev_env("AVSEP_NO_H2_ATTN") != nullptr;                             // developer A/B
  if (qkv && qkv->qkv && h2 && op && !no_h2_attn && attention_is_split(c, Lq, Lk) &&
      attention_h2_supported(c->dh, Lq, Lk, qkv->eq, qkv->ek, qkv->ev))
    return profiled(c, "attention_h2_kernel<2>", flops, bytes, s, [&] {
      return launch_attention_h2(q, ldq, k, ldk, v, ldv, o, ldo, B, c->h, c->dh, Lq, Lk, qkv->eq, qkv->ek, qkv->ev, s, op, o_rows, h2_exp);
    });
  // split-precision attention (attention_split.hip): the models whose Linear layers run on the split-precision GEMM
  // (d_model >= 512), head width 64, sequences of 128 keys or more -- the domain of the LDS-staged fp32 kernel it replaces.
  // The rule looks at the model and the sequence lengths only, never at the batch size.
  static const bool no_split_attn = dev_env("AVSEP_NO_SPLIT_ATTN") != nullptr;                       // developer A/B
  if (c->split_gemm && c->d >= c->split_min && Lk >= 128 && attention_split_supported(c->dh, Lq, Lk) && !no_split_attn)
    return profiled(c, "attention_split_kernel<2>", flops, bytes, s, [&] {
      return launch_attention_split(q, ldq, k, ldk, v, ldv, o, ldo, B, c->h, c->dh, Lq, Lk, 1.0f, s, op, o_rows, h2, h2_exp);
    });
  if (op) return fail(AVSEP_EINTERNAL, "plane output asked of the fp32 attention kernel");
  return profiled(c, attention_instance_name(c->dh, Lq, Lk, B, c->h), flops, bytes, s,
                  [&] { return launch_attention(q, ldq, k, ldk, v, ldv, o, ldo, B, c->h, c->dh, Lq, Lk, s); });
}
bool attention_is_split(const avsep_ctx* c, int Lq, int Lk) {
  static const bool no_split_attn = dev_env("AVSEP_NO_SPLIT_ATTN") != nullptr;
  return c->split_gemm && c->d >= c->split_min && Lk >= 128 && attention_split_supported(c->dh, Lq, Lk) && !no_split_attn;
}

GemmParams linear_params(const float* A, int lda, const float* W, int K, const float* bias, float* C, int ldc,
                         int M, int N, int act);

// x += softmax(q k^T) v W_o^T + b_o: one launch for short sequences (attn_proj_kernel), else attention into `att` and the
// projection GEMM with its residual epilogue
// att_p (rows): with planes_rows(), the attention output goes to the projection as planes (split-precision attention only)
int run_attention_proj(avsep_ctx* c, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* att,
                       const float* wo, const float* bo, float* x, int B, int Lq, int Lk, hipStream_t s,
                       unsigned short* att_p = nullptr, long long rows = 0, const avsep_ctx::H2Site* h2 = nullptr,
                       const avsep_ctx::H2Site* qkv = nullptr) {
  const int d = c->d, M = B * Lq;
  if (att_p && attention_is_split(c, Lq, Lk)) {
    RCK(run_attention(c, q, ldq, k, ldk, v, ldv, att, d, B, Lq, Lk, s, att_p, rows, h2 != nullptr, h2 ? h2->eA : 0, qkv));
    GemmParams po = linear_params(nullptr, d, wo, d, bo, x, d, M, d, ACT_NONE);
    po.Ap = att_p; po.a_rows = rows; po.h2 = h2 != nullptr;
    po.R = x; po.ldr = d; po.rperiod = 0;
    return run_gemm(c, po, s);
  }
  if (attn_proj_supported(c->h, c->dh, Lk)) {
    const double flops = 4.0 * B * c->h * (double)Lq * Lk * c->dh + 2.0 * M * (double)d * d;
    const double bytes = 4.0 * B * d * (1.0 * Lq + 2.0 * Lk) + 4.0 * d * d + 8.0 * M * d;
    return profiled(c, attn_proj_instance_name(c->h), flops, bytes, s,
                    [&] { return launch_attn_proj(q, ldq, k, ldk, v, ldv, wo, bo, x, B, c->h, c->dh, Lq, Lk, s); });
  }
  RCK(run_attention(c, q, ldq, k, ldk, v, ldv, att, d, B, Lq, Lk, s));
  GemmParams po = linear_params(att, d, wo, d, bo, x, d, M, d, ACT_NONE);
  po.R = x; po.ldr = d; po.rperiod = 0;
  return run_gemm(c, po, s);
}

// y = act( LayerNorm(x) W^T + b ).  At small M the LayerNorm is folded into the GEMM (one launch less per LayerNorm, 13 per
// forward): algebraically, into the epilogue of a plain GEMM on the raw rows (GemmParams::lnx_c1; round 3) -- the round-2
// form, statistics and normalisation of a whole-K register slab in front of the first MFMA (gemm_ln_kernel), remains as
// avsep_op_ln_linear form 1 and behind AVSEP_NO_LNX.  Once the problem is big enough for 64x64 tiles to fill the chip the
// stand-alone LayerNorm + big-tile GEMM wins.
int run_ln_linear(avsep_ctx* c, const float* x, const float* g, const float* be, float* ln_buf, const float* W,
                  const float* bias, float* y, int M, int N, int act, hipStream_t s) {
  const int d = c->d;
  GemmParams p{};
  p.A = x; p.W = W; p.bias = bias; p.C = y;
  p.M = M; p.N = N; p.K = d;
  p.lda = d; p.ldw = d; p.ldc = N;
  p.amode = AMODE_PLAIN;
  p.act = act;
  const long big_tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
  static const bool no_fuse = dev_env("AVSEP_NO_LN_FUSE") != nullptr;   // developer A/B switch
  if (c->use_lnx && (big_tiles < 1024 || c->lnx_all)) {
    auto it = c->lnx.find(W);
    if (it != c->lnx.end()) {
      p.W = it->second.w; p.bias = nullptr; p.lnx_c1 = it->second.c1; p.lnx_c2 = it->second.c2; p.ln_eps = 1e-5f;
      return run_gemm(c, p, s);
    }
  }
  if (!no_fuse && gemm_ln_supported(d) && big_tiles < 1024) {
    p.ln_gamma = g; p.ln_beta = be; p.ln_eps = 1e-5f;
    return run_gemm(c, p, s);
  }
  // Large M, staged form: one statistics launch (mean, 1/std per row; it reads x once and writes 8 bytes per row), and
  // the GEMM normalises its A tile on the way to LDS -- the normalised tensor is never written or re-read.  ln_buf
  // holds the (M, 2) statistics.
  // Measured NEGATIVE end to end (profiles/r02_ab_layernorm_staged.txt: cfg3 -2.5 %, cfg5 -2 %): the stand-alone
  // LayerNorm already runs at 5.5 TB/s (12 us at M = 16064, d = 512), and the normalisation arithmetic in front of
  // the tile's LDS writes lengthens every K step of the GEMM by more than the 6 us per LayerNorm it saves.  The form
  // stays available (avsep_op_ln_linear form 2, bit-identical) behind this developer switch.
  static const bool staged = dev_env("AVSEP_LN_STAGED") != nullptr;
  if (staged && gemm_ln_staged_supported(d)) {
    RCK(profiled(c, layernorm_instance_name(d, true), 6.0 * M * d, 1.0 * M * d * 4 + 8.0 * M, s,
                 [&] { return launch_layernorm_stats(x, ln_buf, M, d, 1e-5f, s); }));
    p.ln_gamma = g; p.ln_beta = be; p.ln_eps = 1e-5f; p.ln_stats = ln_buf;
    return run_gemm(c, p, s);
  }
  RCK(run_layernorm(c, x, g, be, ln_buf, M, d, s));
  p.A = ln_buf;
  return run_gemm(c, p, s);
}

// run_ln_linear on pre-split operands: LayerNorm writes the planes of its output into ln_p, the GEMM reads them; its result goes
// to y (fp32) or, with yp, to the planes of the NEXT GEMM's operand (FFN-1 -> FFN-2, decoder layer 1 -> mask head)
int run_ln_linear_planes(avsep_ctx* c, const float* x, const float* g, const float* be, unsigned short* ln_p, long long rows,
                         const float* W, const float* bias, float* y, unsigned short* yp, int M, int N, int act, hipStream_t s) {
  const int d = c->d;
  RCK(run_layernorm_planes(c, x, g, be, ln_p, rows, M, d, s));
  GemmParams p = linear_params(nullptr, d, W, d, bias, y, N, M, N, act);
  p.Ap = ln_p; p.a_rows = rows;
  p.Cp = yp; p.c_rows = rows;
  return run_gemm(c, p, s);
}

// ... on two fp16 terms (gemm_h2.hip): `site` = h2_site(W); next = the site of the GEMM that consumes the plane output yp (its
// static exponent scales the planes), null with y
int run_ln_linear_h2(avsep_ctx* c, const float* x, const float* g, const float* be, unsigned short* ln_p, long long rows,
                     const float* W, const float* bias, float* y, unsigned short* yp, const avsep_ctx::H2Site* site,
                     const avsep_ctx::H2Site* next, int M, int N, int act, hipStream_t s) {
  const int d = c->d;
  RCK(run_layernorm_h2(c, x, g, be, ln_p, rows, M, d, site->eA, s));
  GemmParams p = linear_params(nullptr, d, W, d, bias, y, N, M, N, act);
  p.Ap = ln_p; p.a_rows = rows; p.h2 = 1;
  if (yp) { p.Cp = yp; p.c_rows = rows; p.cp_scale = std::ldexp(1.0f, next->eA); }
  return run_gemm(c, p, s);
}

// ------------------------------------------------------------------------------------------ building blocks
GemmParams linear_params(const float* A, int lda, const float* W, int K, const float* bias, float* C, int ldc,
                         int M, int N, int act);

#ifdef AVSEP_DEV   // paired launches: a measured-slower schedule kept as a developer experiment (see forward_paired)
// The second problem of a pair launch (kernels.h GemmParams::alt): p1 differs from p0 in operands and row count only.
GemmParams pair_params(const GemmParams& p0, const GemmParams& p1) {
  GemmParams p = p0;
  p.alt.A = p1.A; p.alt.W = p1.W; p.alt.bias = p1.bias; p.alt.R = p1.R; p.alt.rperiod = p1.rperiod;
  p.alt.ln_gamma = p1.ln_gamma; p.alt.ln_beta = p1.ln_beta; p.alt.C = p1.C; p.alt.M = p1.M;
  return p;
}
int run_gemm_pair(avsep_ctx* c, const GemmParams& p0, const GemmParams& p1, hipStream_t s) {
  const GemmParams p = pair_params(p0, p1);       // for the instance name and the flop / byte accounting
  const double rows = (double)p0.M + p1.M;
  const double flops = 2.0 * rows * p.N * p.K;
  double bytes = rows * p.K * 4 + 2.0 * p.N * p.K * 4 + rows * p.N * 4;
  if (p.R) bytes += rows * p.N * 4;
  return profiled(c, c->prof_on ? gemm_instance_name(p) : "", flops, bytes, s, [&] { return launch_gemm_pair(p0, p1, s); });
}

// y = act(LayerNorm(x) W^T + b) for the audio and the visual instance of the same layer in one launch (two LayerNorm
// launches in front of it where the LayerNorm is not fused, d_model > 256 or a large batch)
struct LnLin { const float *x, *g, *be; float* ln_buf; const float *W, *bias; float* y; int M; };
int run_ln_linear_pair(avsep_ctx* c, const LnLin& a, const LnLin& v, int N, int act, hipStream_t s) {
  const int d = c->d;
  GemmParams pa = linear_params(a.x, d, a.W, d, a.bias, a.y, N, a.M, N, act);
  GemmParams pv = linear_params(v.x, d, v.W, d, v.bias, v.y, N, v.M, N, act);
  const long big_tiles = (long)((a.M + 63) / 64 + (v.M + 63) / 64) * ((N + 63) / 64);
  static const bool no_fuse = dev_env("AVSEP_NO_LN_FUSE") != nullptr;   // developer A/B switch
  if (!no_fuse && gemm_ln_supported(d) && big_tiles < 1024) {
    pa.ln_gamma = a.g; pa.ln_beta = a.be; pa.ln_eps = 1e-5f;
    pv.ln_gamma = v.g; pv.ln_beta = v.be; pv.ln_eps = 1e-5f;
    return run_gemm_pair(c, pa, pv, s);
  }
  RCK(run_layernorm(c, a.x, a.g, a.be, a.ln_buf, a.M, d, s));
  RCK(run_layernorm(c, v.x, v.g, v.be, v.ln_buf, v.M, d, s));
  pa.A = a.ln_buf;
  pv.A = v.ln_buf;
  return run_gemm_pair(c, pa, pv, s);
}
#endif  // AVSEP_DEV

GemmParams linear_params(const float* A, int lda, const float* W, int K, const float* bias, float* C, int ldc,
                         int M, int N, int act) {
  GemmParams p{};
  p.A = A; p.W = W; p.bias = bias; p.C = C;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldw = K; p.ldc = ldc;
  p.amode = AMODE_PLAIN;
  p.act = act;
  return p;
}

// ---- schedule 1: all pre-norm encoder layers of one branch as ONE persistent, dependency-driven launch (chain.hip).
// The ops and their operands are exactly those of encoder_layer() below (LayerNorm-epilogue QKV / FFN-1, short-sequence
// attention, plain out-projection / FFN-2 with the residual in place), so the outputs are the launch-per-op path's bit for bit.
#ifdef AVSEP_DEV
bool chain_usable(const avsep_ctx* c, const std::vector<EncLayerW>& layers, int B, int L) {
  const int d = c->d, M = B * L;
  if (c->schedule == 0 || c->keep_taps || layers.empty() || !c->use_lnx || c->lnx_all) return false;
  if (c->dh != 64 || L <= 48 || L > 64 || (d & 63)) return false;
  // the launch-per-op path takes the LayerNorm-epilogue form below 1024 64x64 tiles only (run_ln_linear): same rule, same bits
  if ((long)((M + 63) / 64) * ((4 * d + 63) / 64) >= 1024) return false;
  for (const auto& Lw : layers)
    if (c->lnx.find(Lw.wqkv) == c->lnx.end() || c->lnx.find(Lw.w1) == c->lnx.end()) return false;
  return true;
}

// the plan of (x, B, L, layers): device tables built once (hipMalloc + blocking copies -- NOT under stream capture: callers
// that capture call prepare_chains() first), kept until the context dies or avsep_set_schedule() changes the order
int get_chain(avsep_ctx* c, const std::vector<EncLayerW>& layers, float* x, float* qkv, float* att, float* ffn, int B, int L,
              ChainPlanImpl** out) {
  for (auto& e : c->chains)
    if (e.x == x && e.B == B && e.L == L && e.layers == (const void*)layers.data()) { *out = e.plan; return AVSEP_OK; }
  const int d = c->d, M = B * L;
  ChainBuilder* b = chain_builder_new();
  if (!b) return fail(AVSEP_ENOMEM, "host allocation failed");
  int prev = -1;
  bool ok = true;
  for (const auto& Lw : layers) {
    GemmParams pq = linear_params(x, d, nullptr, d, nullptr, qkv, 3 * d, M, 3 * d, ACT_NONE);
    const LnxW& lq = c->lnx.at(Lw.wqkv);
    pq.W = lq.w; pq.lnx_c1 = lq.c1; pq.lnx_c2 = lq.c2; pq.ln_eps = 1e-5f;
    const int o_qkv = chain_add_gemm(b, pq, prev, L);
    const AttnProblem pa{qkv, qkv + d, qkv + 2 * d, att, 3 * d, 3 * d, 3 * d, d, B, L, L};
    const int o_att = chain_add_attention(b, pa, c->h, o_qkv);
    GemmParams po = linear_params(att, d, Lw.wo, d, Lw.bo, x, d, M, d, ACT_NONE);
    po.R = x; po.ldr = d; po.rperiod = 0;
    const int o_out = chain_add_gemm(b, po, o_att, L);
    GemmParams p1 = linear_params(x, d, nullptr, d, nullptr, ffn, 4 * d, M, 4 * d, ACT_RELU);
    const LnxW& l1 = c->lnx.at(Lw.w1);
    p1.W = l1.w; p1.lnx_c1 = l1.c1; p1.lnx_c2 = l1.c2; p1.ln_eps = 1e-5f;
    const int o_f1 = chain_add_gemm(b, p1, o_out, L);
    GemmParams p2 = linear_params(ffn, 4 * d, Lw.w2, 4 * d, Lw.b2, x, d, M, d, ACT_NONE);
    p2.R = x; p2.ldr = d; p2.rperiod = 0;
    prev = chain_add_gemm(b, p2, o_f1, L);
    ok = ok && o_qkv >= 0 && o_att >= 0 && o_out >= 0 && o_f1 >= 0 && prev >= 0;
  }
  ChainPlanImpl* plan = nullptr;
  hipError_t e = ok ? chain_build(b, c->schedule == 2 ? 1 : 0, c->chain_group, c->chain_skew, &plan) : hipErrorInvalidValue;
  chain_builder_free(b);
  if (e != hipSuccess) return fail_hip(e, "chain_build (encoder layers)");
  c->chains.push_back({x, B, L, (const void*)layers.data(), plan});
  *out = plan;
  return AVSEP_OK;
}

#endif  // AVSEP_DEV

// plane buffers of one branch (null: the stage has too few rows for the pre-split GEMM, see planes_rows())
struct BranchPlanes { unsigned short *ln, *att, *ffn; long long rows; bool h2; };   // h2: two fp16 terms (gemm_h2.hip)
int encoder_layer(avsep_ctx* c, const EncLayerW& L, float* x, float* ln, float* qkv, float* att, float* ffn, int B,
                  int Lseq, hipStream_t s, const BranchPlanes& bp);

// every encoder layer of one branch, in place on x
int encoder_layers(avsep_ctx* c, const std::vector<EncLayerW>& layers, const Workspace& w, bool audio, int B, int L,
                   hipStream_t s) {
  float* x = audio ? w.a_x : w.v_x;
  float* ln = audio ? w.ln : w.v_ln;
  float* qkv = audio ? w.qkv : w.v_qkv;
  float* att = audio ? w.att : w.v_att;
  float* ffn = audio ? w.ffn : w.v_ffn;
#ifdef AVSEP_DEV
  if (chain_usable(c, layers, B, L)) {
    ChainPlanImpl* plan = nullptr;
    RCK(get_chain(c, layers, x, qkv, att, ffn, B, L, &plan));
    return profiled(c, audio ? "chain_kernel (audio encoder layers)" : "chain_kernel (visual encoder layers)",
                    chain_plan_flops(plan), chain_plan_bytes(plan), s, [&] { return launch_chain(plan, s); });
  }
#endif
  BranchPlanes bp{nullptr, nullptr, nullptr, 0, false};
  const bool h2 = !layers.empty() && h2_site(c, w, layers[0].wqkv) != nullptr;                       // at EVERY row count: other bits
  if (h2 || planes_rows(c, w, B * L))
    bp = audio ? BranchPlanes{w.ln_p, w.att_p, w.ffn_p, w.rows_a, h2} : BranchPlanes{w.v_ln_p, w.v_att_p, w.v_ffn_p, w.rows_v, h2};
  for (size_t i = 0; i < layers.size(); ++i) {
    RCK(encoder_layer(c, layers[i], x, ln, qkv, att, ffn, B, L, s, bp));
    RCK(record_tap(c, w, ((audio ? "a_enc" : "v_enc") + std::to_string(i)).c_str(), x, (size_t)B * L * c->d, s));
  }
  return AVSEP_OK;
}

// one pre-norm encoder layer: x += Wo*Attn(LN1 x); x += W2*relu(W1*LN2 x)   (model.py:48-52, norm_first)
int encoder_layer(avsep_ctx* c, const EncLayerW& L, float* x, float* ln, float* qkv, float* att, float* ffn, int B,
                  int Lseq, hipStream_t s, const BranchPlanes& bp) {
  const int d = c->d, M = B * Lseq;
  if (bp.ln && bp.h2) {   // two fp16 terms, three products: LayerNorm, attention and FFN-1 write the scaled planes their consumers read
    const avsep_ctx::H2Site *sq = &c->h2.at(L.wqkv), *so = &c->h2.at(L.wo), *s1 = &c->h2.at(L.w1), *s2 = &c->h2.at(L.w2);
    RCK(run_ln_linear_h2(c, x, L.g1, L.be1, bp.ln, bp.rows, L.wqkv, L.bqkv, qkv, nullptr, sq, nullptr, M, 3 * d, ACT_NONE, s));
    RCK(run_attention_proj(c, qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, att, L.wo, L.bo, x, B, Lseq, Lseq, s, bp.att, bp.rows, so, sq));
    RCK(run_ln_linear_h2(c, x, L.g2, L.be2, bp.ln, bp.rows, L.w1, L.b1, nullptr, bp.ffn, s1, s2, M, 4 * d, ACT_RELU, s));
    GemmParams p2 = linear_params(nullptr, 4 * d, L.w2, 4 * d, L.b2, x, d, M, d, ACT_NONE);
    p2.Ap = bp.ffn; p2.a_rows = bp.rows; p2.h2 = 1;
    p2.R = x; p2.ldr = d; p2.rperiod = 0;
    return run_gemm(c, p2, s);
  }
  if (bp.ln) {   // pre-split operands: LayerNorm, attention and FFN-1 write the planes their consumer GEMMs read (same bits)
    RCK(run_ln_linear_planes(c, x, L.g1, L.be1, bp.ln, bp.rows, L.wqkv, L.bqkv, qkv, nullptr, M, 3 * d, ACT_NONE, s));
    RCK(run_attention_proj(c, qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, att, L.wo, L.bo, x, B, Lseq, Lseq, s, bp.att, bp.rows));
    RCK(run_ln_linear_planes(c, x, L.g2, L.be2, bp.ln, bp.rows, L.w1, L.b1, nullptr, bp.ffn, M, 4 * d, ACT_RELU, s));
    GemmParams p2 = linear_params(nullptr, 4 * d, L.w2, 4 * d, L.b2, x, d, M, d, ACT_NONE);
    p2.Ap = bp.ffn; p2.a_rows = bp.rows;
    p2.R = x; p2.ldr = d; p2.rperiod = 0;
    return run_gemm(c, p2, s);
  }
  RCK(run_ln_linear(c, x, L.g1, L.be1, ln, L.wqkv, L.bqkv, qkv, M, 3 * d, ACT_NONE, s));   // norm1 -> in_proj
  RCK(run_attention_proj(c, qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, att, L.wo, L.bo, x, B, Lseq, Lseq, s));
  RCK(run_ln_linear(c, x, L.g2, L.be2, ln, L.w1, L.b1, ffn, M, 4 * d, ACT_RELU, s));        // norm2 -> linear1
  GemmParams p2 = linear_params(ffn, 4 * d, L.w2, 4 * d, L.b2, x, d, M, d, ACT_NONE);
  p2.R = x; p2.ldr = d; p2.rperiod = 0;
  RCK(run_gemm(c, p2, s));
  return AVSEP_OK;
}

#ifdef AVSEP_DEV
// Layer i of the audio encoder (rows B*La) and of the visual encoder (rows B*Lv) side by side: the two are independent
// and have the same weight shapes (model.py:48-52 / 97-101), so each of the five kernels of a layer is launched ONCE for
// both -- twice the workgroups per launch and half the launches, instead of two streams whose kernels have to find room
// beside each other (a 32-clip batch is ~760 + ~600 workgroups per QKV launch on a chip with 768 resident slots).
int encoder_layer_pair(avsep_ctx* c, const EncLayerW& A, const EncLayerW& V, const Workspace& w, int B, int La, int Lv,
                       hipStream_t s) {
  const int d = c->d, Ma = B * La, Mv = B * Lv;
  RCK(run_ln_linear_pair(c, LnLin{w.a_x, A.g1, A.be1, w.ln, A.wqkv, A.bqkv, w.qkv, Ma},
                         LnLin{w.v_x, V.g1, V.be1, w.v_ln, V.wqkv, V.bqkv, w.v_qkv, Mv}, 3 * d, ACT_NONE, s));
  if (attention_pair_merges(c->dh, La, Lv)) {
    const AttnProblem pa{w.qkv, w.qkv + d, w.qkv + 2 * d, w.att, 3 * d, 3 * d, 3 * d, d, B, La, La};
    const AttnProblem pv{w.v_qkv, w.v_qkv + d, w.v_qkv + 2 * d, w.v_att, 3 * d, 3 * d, 3 * d, d, B, Lv, Lv};
    const double flops = 4.0 * B * c->h * c->dh * ((double)La * La + (double)Lv * Lv);
    const double bytes = 4.0 * B * c->d * 4.0 * (La + Lv);
    RCK(profiled(c, attention_instance_name(c->dh, La, La, B, c->h), flops, bytes, s,
                 [&] { return launch_attention_pair(pa, pv, c->h, c->dh, s); }));
  } else {   // different kernel instances (e.g. T = 251 audio frames, N = 50 lip frames): one launch each
    RCK(run_attention(c, w.qkv, 3 * d, w.qkv + d, 3 * d, w.qkv + 2 * d, 3 * d, w.att, d, B, La, La, s));
    RCK(run_attention(c, w.v_qkv, 3 * d, w.v_qkv + d, 3 * d, w.v_qkv + 2 * d, 3 * d, w.v_att, d, B, Lv, Lv, s));
  }
  GemmParams oa = linear_params(w.att, d, A.wo, d, A.bo, w.a_x, d, Ma, d, ACT_NONE);
  oa.R = w.a_x; oa.ldr = d; oa.rperiod = 0;
  GemmParams ov = linear_params(w.v_att, d, V.wo, d, V.bo, w.v_x, d, Mv, d, ACT_NONE);
  ov.R = w.v_x; ov.ldr = d; ov.rperiod = 0;
  RCK(run_gemm_pair(c, oa, ov, s));
  RCK(run_ln_linear_pair(c, LnLin{w.a_x, A.g2, A.be2, w.ln, A.w1, A.b1, w.ffn, Ma},
                         LnLin{w.v_x, V.g2, V.be2, w.v_ln, V.w1, V.b1, w.v_ffn, Mv}, 4 * d, ACT_RELU, s));
  GemmParams fa = linear_params(w.ffn, 4 * d, A.w2, 4 * d, A.b2, w.a_x, d, Ma, d, ACT_NONE);
  fa.R = w.a_x; fa.ldr = d; fa.rperiod = 0;
  GemmParams fv = linear_params(w.v_ffn, 4 * d, V.w2, 4 * d, V.b2, w.v_x, d, Mv, d, ACT_NONE);
  fv.R = w.v_x; fv.ldr = d; fv.rperiod = 0;
  RCK(run_gemm_pair(c, fa, fv, s));
  return AVSEP_OK;
}
#endif  // AVSEP_DEV

int check_common(const avsep_ctx* c, int B, int T) {
  if (!c) return fail(AVSEP_EINVAL, "null context");
  if (!c->finalized) return fail(AVSEP_ESTATE, "avsep_finalize_weights() has not been called");
  if (B <= 0 || T <= 0) return fail(AVSEP_EINVAL, "B and T must be positive");
  return AVSEP_OK;
}

// AudioEncoder.input_proj + pos_enc (model.py:56-58): (B,F,T) -> w.a_x (B*T, d).  Also fills w.xt (mixed^T, padded).
int audio_front(avsep_ctx* c, const Workspace& w, const float* mixed, int B, int T, hipStream_t s) {
  if (!c->ok_audio) return fail(AVSEP_ENOWEIGHT, "audio_encoder weights are incomplete");
  if (T > c->pe_len_a)
    return fail(AVSEP_EINVAL, "sequence length exceeds PositionalEncoding max_len (model.py:286,300)");
  const int d = c->d, M = B * T;
  RCK(profiled(c, "transpose_pad_kernel", 0.0, 4.0 * B * T * (c->F + c->Fp), s,
               [&] { return launch_transpose_pad(mixed, w.xt, B, c->F, T, c->Fp, s); }));
  GemmParams p{};
  p.A = w.xt; p.lda = c->Fp; p.W = c->a_w1; p.ldw = 3 * c->Fp; p.bias = c->a_b1; p.C = w.a_h0; p.ldc = d;
  p.M = M; p.N = d; p.K = 3 * c->Fp; p.amode = AMODE_TAPS3; p.T = T; p.Kt = c->Fp; p.act = ACT_RELU; p.zeros = c->zeros;
  RCK(run_gemm(c, p, s, 3 * c->F));
  RCK(record_tap(c, w, "a_conv1", w.a_h0, (size_t)M * d, s));
  GemmParams p2{};
  p2.A = w.a_h0; p2.lda = d; p2.W = c->a_w2; p2.ldw = 3 * d; p2.bias = c->a_b2; p2.C = w.a_x; p2.ldc = d;
  p2.M = M; p2.N = d; p2.K = 3 * d; p2.amode = AMODE_TAPS3; p2.T = T; p2.Kt = d; p2.act = ACT_RELU; p2.zeros = c->zeros;
  p2.R = c->a_pe; p2.ldr = d; p2.rperiod = T;   // x + pe[:, :T]  (model.py:300), fused after the ReLU
  RCK(run_gemm(c, p2, s));
  RCK(record_tap(c, w, "a_pe", w.a_x, (size_t)M * d, s));
  return AVSEP_OK;
}

// AudioEncoder.forward (model.py:54-60); result left in w.a_x.
int audio_branch(avsep_ctx* c, const Workspace& w, const float* mixed, int B, int T, hipStream_t s) {
  RCK(audio_front(c, w, mixed, B, T, s));
  return encoder_layers(c, c->a_layers, w, /*audio=*/true, B, T, s);
}

// VisualEncoder.conv + frame_proj + pos_enc (model.py:106-110): (B,N,H,W) -> w.v_x (B*N, d).
int visual_front(avsep_ctx* c, const Workspace& w, const float* lips, int B, int N, int H, int W, hipStream_t s) {
  if (!c->ok_visual) return fail(AVSEP_ENOWEIGHT, "visual_encoder weights are incomplete");
  if (N <= 0 || H <= 0 || W <= 0) return fail(AVSEP_EINVAL, "N, H, W must be positive");
  if (N > c->pe_len_v) return fail(AVSEP_EINVAL, "frame count exceeds PositionalEncoding max_len");
  const int d = c->d, Mv = B * N;
  const int H1 = conv_out(H), W1 = conv_out(W), H2 = conv_out(H1), W2 = conv_out(W1), H3 = conv_out(H2),
            W3 = conv_out(W2);
  // fused LDS-resident conv stack (conv_stack.hip) unless debug taps want the intermediate activations or the
  // frame size does not fit it
  bool fused = !c->keep_taps && !c->no_fused_conv;
  if (fused) {
    const double fl = 2.0 * Mv * ((double)H1 * W1 * 32 * 9 + (double)H2 * W2 * 64 * 288 + (double)H3 * W3 * 128 * 576);
    hipError_t e = hipSuccess;
    int r = profiled(c, conv_stack_instance_name(Mv, H, W, c->conv_h2.w2h != nullptr), fl, 4.0 * Mv * ((double)H * W + 128), s, [&] {
      e = launch_conv_stack(lips, c->c1_w, c->c1_b, c->c2_w, c->c2_b, c->c3_w, c->c3_b, w.pool, Mv, H, W, s,
                            c->conv_h2.w2h ? &c->conv_h2 : nullptr);
      return e == hipErrorNotSupported ? hipSuccess : e;
    });
    RCK(r);
    if (e == hipErrorNotSupported) fused = false;
  }
  if (!fused) {
    RCK(profiled(c, "conv1_c1_kernel", 2.0 * Mv * H1 * W1 * 32 * 9, 4.0 * Mv * (H * W + H1 * W1 * 32), s,
                 [&] { return launch_conv1_c1(lips, c->c1_w, c->c1_b, w.act1, Mv, H, W, H1, W1, s); }));
    RCK(record_tap(c, w, "v_conv0", w.act1, (size_t)Mv * H1 * W1 * 32, s));
    GemmParams p{};
    p.A = w.act1; p.W = c->c2_w; p.ldw = 9 * 32; p.bias = c->c2_b; p.C = w.act2; p.ldc = 64;
    p.M = Mv * H2 * W2; p.N = 64; p.K = 9 * 32; p.amode = AMODE_CONV2D; p.Kt = 32;
    p.Hin = H1; p.Win = W1; p.Hout = H2; p.Wout = W2; p.act = ACT_RELU;
    RCK(run_gemm(c, p, s));
    RCK(record_tap(c, w, "v_conv1", w.act2, (size_t)Mv * H2 * W2 * 64, s));
    GemmParams p3{};
    p3.A = w.act2; p3.W = c->c3_w; p3.ldw = 9 * 64; p3.bias = c->c3_b; p3.C = w.act3; p3.ldc = 128;
    p3.M = Mv * H3 * W3; p3.N = 128; p3.K = 9 * 64; p3.amode = AMODE_CONV2D; p3.Kt = 64;
    p3.Hin = H2; p3.Win = W2; p3.Hout = H3; p3.Wout = W3; p3.act = ACT_RELU;
    RCK(run_gemm(c, p3, s));
    RCK(record_tap(c, w, "v_conv2", w.act3, (size_t)Mv * H3 * W3 * 128, s));
    RCK(profiled(c, "avgpool_kernel", 1.0 * Mv * H3 * W3 * 128, 4.0 * Mv * 128 * (H3 * W3 + 1), s,
                 [&] { return launch_avgpool(w.act3, w.pool, Mv, H3 * W3, 128, s); }));
  }
  RCK(record_tap(c, w, "v_pool", w.pool, (size_t)Mv * 128, s));
  GemmParams pf = linear_params(w.pool, 128, c->fp_w, 128, c->fp_b, w.v_x, d, Mv, d, ACT_NONE);
  pf.R = c->v_pe; pf.ldr = d; pf.rperiod = N;   // PE indexed by frame position
  RCK(run_gemm(c, pf, s));
  return AVSEP_OK;
}

// F.interpolate(mode="linear") of the frame sequence to the audio length (model.py:114-116): w.v_x -> w.v_up
int visual_upsample(avsep_ctx* c, const Workspace& w, int B, int N, int T, hipStream_t s) {
  const int d = c->d;
  if (c->Lf > 0 && h2_site(c, w, c->wkv_all))       // only the fusion K/V projection reads it: two fp16 terms, one power of two per row
    return profiled(c, "interp_linear_h2_kernel<1>", 3.0 * B * T * d, 4.0 * B * d * (N + T), s,
                    [&] { return launch_interp_linear_h2(w.v_x, w.v_up_p, w.v_up_rs, w.rows_a, B, N, T, d, s); });
  if (planes_rows(c, w, B * T))                    // (bf16 path) its three bf16 planes instead of the fp32 tensor
    return profiled(c, "interp_linear_kernel<true>", 3.0 * B * T * d, 4.0 * B * d * (N + 1.5 * T), s,
                    [&] { return launch_interp_linear_planes(w.v_x, w.v_up_p, w.rows_a, B, N, T, d, s); });
  RCK(profiled(c, "interp_linear_kernel<false>", 3.0 * B * T * d, 4.0 * B * d * (N + T), s,
               [&] { return launch_interp_linear(w.v_x, w.v_up, B, N, T, d, s); }));
  return record_tap(c, w, "v_interp", w.v_up, (size_t)B * T * d, s);
}

// VisualEncoder.forward (model.py:103-117); result (B,T,d) left in w.v_up.
int visual_branch(avsep_ctx* c, const Workspace& w, const float* lips, int B, int N, int H, int W, int T,
                  hipStream_t s) {
  RCK(visual_front(c, w, lips, B, N, H, W, s));
  RCK(encoder_layers(c, c->v_layers, w, /*audio=*/false, B, N, s));
  return visual_upsample(c, w, B, N, T, s);
}

// K/V projections of the (layer-invariant) visual stream for ALL fusion layers in one GEMM.
int fusion_kv(avsep_ctx* c, const Workspace& w, const float* visual, int B, int T, hipStream_t s) {
  if (!c->ok_fusion) return fail(AVSEP_ENOWEIGHT, "fusion weights are incomplete");
  if (c->Lf == 0) return AVSEP_OK;
  const int d = c->d, M = B * T, nkv = c->Lf * 2 * d;
  GemmParams p = linear_params(visual, d, c->wkv_all, d, c->bkv_all, w.kv_all, nkv, M, nkv, ACT_NONE);
  if (visual == w.v_up && h2_site(c, w, c->wkv_all)) {   // the full forward: visual_upsample() wrote row-scaled two-term planes, not w.v_up
    p.A = nullptr; p.Ap = w.v_up_p; p.a_rows = w.rows_a; p.h2 = 1; p.rscale = w.v_up_rs;
  } else if (visual == w.v_up && planes_rows(c, w, M)) {   // (bf16 path) three bf16 planes
    p.A = nullptr; p.Ap = w.v_up_p; p.a_rows = w.rows_a;
  }
  RCK(run_gemm(c, p, s));
  if (p.h2 && c->cross_h2 && w.f_clip_exp)   // the cross-attention's per-clip k / v exponents, from the row scales of the resized visual stream
    RCK(profiled(c, "clip_exp_kernel", 0.0, 4.0 * M, s, [&] { return launch_clip_exp(w.v_up_rs, c->fkv_const, w.f_clip_exp, B, T, c->Lf, s); }));
  return AVSEP_OK;
}

// CrossModalFusion.forward (model.py:145-149) in place on x; final LayerNorm written to w.ln.
int fusion_layer(avsep_ctx* c, const Workspace& w, float* x, int B, int T, int i, hipStream_t s) {
  const int d = c->d, M = B * T, nkv = c->Lf * 2 * d;
  const FusLayerW& L = c->f_layers[i];
  const float* kk = w.kv_all + (size_t)i * 2 * d;
  if (const avsep_ctx::H2Site* sq = h2_site(c, w, L.wq)) {   // two fp16 terms, see encoder_layer()
    const avsep_ctx::H2Site *s1 = &c->h2.at(L.w1), *s2 = &c->h2.at(L.w2);
    RCK(run_ln_linear_h2(c, x, L.g1, L.be1, w.ln_p, w.rows_a, L.wq, L.bq, w.f_q, nullptr, sq, nullptr, M, d, ACT_NONE, s));
    // the cross-attention output (a combination of rows of the VISUAL stream's projection) has no static bound: its projection stays
    // on three bf16 terms -- as planes from the 256 x 128 kernel's row count on, else cut in flight (same bits)
    static const bool no_cross_h2 = dev_env("AVSEP_NO_H2_CROSS") != nullptr;                          // developer A/B
    if (c->cross_h2 && !no_cross_h2 && sq->qkv && w.f_clip_exp && attention_is_split(c, T, T) && attention_h2_supported(c->dh, T, T, sq->eq, 0, 0)) {
      // ... unless its K and V carry a bound per CLIP (clip_exp_kernel): two fp16 terms, the output scaled by the clip's power of two, the
      // projection descaled per row -- at every row count
      const avsep_ctx::H2Site* so = &c->h2.at(L.wo);
      const int* ce = w.f_clip_exp + (size_t)i * 2;
      const double flops = 4.0 * B * c->h * (double)T * T * c->dh, bytes = 4.0 * B * c->d * 4.0 * T;
      RCK(profiled(c, "attention_h2_kernel<2>", flops, bytes, s, [&] {
        return launch_attention_h2(w.f_q, d, kk, nkv, kk + d, nkv, nullptr, d, B, c->h, c->dh, T, T, sq->eq, 0, 0, s, w.att_p, w.rows_a, 0, ce,
                                   c->Lf * 2, w.f_att_rs);
      }));
      GemmParams po = linear_params(nullptr, d, L.wo, d, L.bo, x, d, M, d, ACT_NONE);
      po.Ap = w.att_p; po.a_rows = w.rows_a; po.h2 = 1; po.rscale = w.f_att_rs;
      po.R = x; po.ldr = d; po.rperiod = 0;
      (void)so;
      RCK(run_gemm(c, po, s));
    } else if (planes_rows(c, w, M)) RCK(run_attention_proj(c, w.f_q, d, kk, nkv, kk + d, nkv, w.att, L.wo, L.bo, x, B, T, T, s, w.att_p, w.rows_a));
    else RCK(run_attention_proj(c, w.f_q, d, kk, nkv, kk + d, nkv, w.att, L.wo, L.bo, x, B, T, T, s));
    RCK(run_ln_linear_h2(c, x, L.g2, L.be2, w.ln_p, w.rows_a, L.w1, L.b1, nullptr, w.ffn_p, s1, s2, M, 4 * d, ACT_GELU, s));
    GemmParams p2 = linear_params(nullptr, 4 * d, L.w2, 4 * d, L.b2, x, d, M, d, ACT_NONE);
    p2.Ap = w.ffn_p; p2.a_rows = w.rows_a; p2.h2 = 1;
    p2.R = x; p2.ldr = d;
    RCK(run_gemm(c, p2, s));
    return record_tap(c, w, ("f_layer" + std::to_string(i)).c_str(), x, (size_t)M * d, s);
  }
  if (planes_rows(c, w, M)) {                       // pre-split operands, see encoder_layer()
    RCK(run_ln_linear_planes(c, x, L.g1, L.be1, w.ln_p, w.rows_a, L.wq, L.bq, w.f_q, nullptr, M, d, ACT_NONE, s));
    RCK(run_attention_proj(c, w.f_q, d, kk, nkv, kk + d, nkv, w.att, L.wo, L.bo, x, B, T, T, s, w.att_p, w.rows_a));
    RCK(run_ln_linear_planes(c, x, L.g2, L.be2, w.ln_p, w.rows_a, L.w1, L.b1, nullptr, w.ffn_p, M, 4 * d, ACT_GELU, s));
    GemmParams p2 = linear_params(nullptr, 4 * d, L.w2, 4 * d, L.b2, x, d, M, d, ACT_NONE);
    p2.Ap = w.ffn_p; p2.a_rows = w.rows_a;
    p2.R = x; p2.ldr = d;
    RCK(run_gemm(c, p2, s));
    return record_tap(c, w, ("f_layer" + std::to_string(i)).c_str(), x, (size_t)M * d, s);
  }
  RCK(run_ln_linear(c, x, L.g1, L.be1, w.ln, L.wq, L.bq, w.f_q, M, d, ACT_NONE, s));       // norm1 -> q proj
  RCK(run_attention_proj(c, w.f_q, d, kk, nkv, kk + d, nkv, w.att, L.wo, L.bo, x, B, T, T, s));
  RCK(run_ln_linear(c, x, L.g2, L.be2, w.ln, L.w1, L.b1, w.ffn, M, 4 * d, ACT_GELU, s));   // norm2 -> ff.0
  GemmParams p2 = linear_params(w.ffn, 4 * d, L.w2, 4 * d, L.b2, x, d, M, d, ACT_NONE);
  p2.R = x; p2.ldr = d;
  RCK(run_gemm(c, p2, s));
  return record_tap(c, w, ("f_layer" + std::to_string(i)).c_str(), x, (size_t)M * d, s);
}

int fusion_layers(avsep_ctx* c, const Workspace& w, float* x, int B, int T, hipStream_t s, bool final_norm) {
  if (!c->ok_fusion) return fail(AVSEP_ENOWEIGHT, "fusion weights are incomplete");
  const int d = c->d, M = B * T;
  for (int i = 0; i < c->Lf; ++i) RCK(fusion_layer(c, w, x, B, T, i, s));
  if (final_norm) {   // stand-alone CrossModalFusion: materialise fusion.norm; the full forward fuses it into the decoder
    RCK(run_layernorm(c, x, c->fn_g, c->fn_b, w.ln, M, d, s));
    RCK(record_tap(c, w, "f_norm", w.ln, (size_t)M * d, s));
  }
  return AVSEP_OK;
}

// SeparationDecoder.forward + .separate (model.py:201-220): masks = sigmoid(W2 gelu(W1 x)), output
// channel n = s*F + f, written as (B,T,S,F); separated = masks * mixed (xt holds mixed^T) in the epilogue.
// `fuse_norm`: `fused` is the un-normalised fusion output and fusion.norm (model.py:149) is applied as the
// LayerNorm prologue of the first decoder GEMM.
int decoder_stage(avsep_ctx* c, const Workspace& w, const float* fused, float* masks, float* sep, int B, int T,
                  hipStream_t s, bool fuse_norm) {
  if (!c->ok_decoder) return fail(AVSEP_ENOWEIGHT, "decoder weights are incomplete");
  const int d = c->d, M = B * T, SF = c->S * c->F;
  // (an odd S * F -- three speakers -- has no two-output epilogue on the bf16 / fp16 pipes: the mask head then runs the fp32 MFMA kernel
  // and the decoder keeps fp32 tensors)
  if (fuse_norm && !(SF & 1) && h2_site(c, w, c->d_w1) && h2_site(c, w, c->d_w2)) {                      // two fp16 terms
    if (!c->ok_fusion) return fail(AVSEP_ENOWEIGHT, "fusion weights are incomplete");
    const avsep_ctx::H2Site *s1 = h2_site(c, w, c->d_w1), *s2 = h2_site(c, w, c->d_w2);
    RCK(run_ln_linear_h2(c, fused, c->fn_g, c->fn_b, w.ln_p, w.rows_a, c->d_w1, c->d_b1, nullptr, w.ffn_p, s1, s2, M, 2 * d, ACT_GELU, s));
    GemmParams p = linear_params(nullptr, 2 * d, c->d_w2, 2 * d, c->d_b2, masks, SF, M, SF, ACT_SIGMOID);
    p.Ap = w.ffn_p; p.a_rows = w.rows_a; p.h2 = 1;
    if (sep) { p.C2 = sep; p.X = w.xt; p.ldx = c->Fp; p.F = c->F; }
    return run_gemm(c, p, s);
  }
  if (fuse_norm && !(SF & 1) && planes_rows(c, w, M) && c->wplanes.count(c->d_w1) && c->wplanes.count(c->d_w2)) {   // pre-split operands
    if (!c->ok_fusion) return fail(AVSEP_ENOWEIGHT, "fusion weights are incomplete");
    RCK(run_ln_linear_planes(c, fused, c->fn_g, c->fn_b, w.ln_p, w.rows_a, c->d_w1, c->d_b1, nullptr, w.ffn_p, M, 2 * d, ACT_GELU, s));
    GemmParams p = linear_params(nullptr, 2 * d, c->d_w2, 2 * d, c->d_b2, masks, SF, M, SF, ACT_SIGMOID);
    p.Ap = w.ffn_p; p.a_rows = w.rows_a;
    if (sep) { p.C2 = sep; p.X = w.xt; p.ldx = c->Fp; p.F = c->F; }
    return run_gemm(c, p, s);
  }
  if (fuse_norm) {
    if (!c->ok_fusion) return fail(AVSEP_ENOWEIGHT, "fusion weights are incomplete");
    RCK(run_ln_linear(c, fused, c->fn_g, c->fn_b, w.ln, c->d_w1, c->d_b1, w.ffn, M, 2 * d, ACT_GELU, s));
  } else {
    RCK(run_gemm(c, linear_params(fused, d, c->d_w1, d, c->d_b1, w.ffn, 2 * d, M, 2 * d, ACT_GELU), s));
  }
  GemmParams p = linear_params(w.ffn, 2 * d, c->d_w2, 2 * d, c->d_b2, masks, SF, M, SF, ACT_SIGMOID);
  if (sep) {
    p.C2 = sep; p.X = w.xt; p.ldx = c->Fp; p.F = c->F;
  }
  RCK(run_gemm(c, p, s));
  return AVSEP_OK;
}

// Workspace view of the clips [b0, B) for the row-local stages after the audio/visual join.
Workspace shift_rows(const avsep_ctx* c, const Workspace& w, int b0, int T) {
  Workspace v = w;
  const size_t rows = (size_t)b0 * T, d = c->d;
  v.xt += rows * c->Fp;
  v.a_x += rows * d;
  v.ln += rows * d;
  v.f_q += rows * d;
  v.att += rows * d;
  v.ffn += rows * 4 * d;
  v.kv_all += rows * (size_t)c->Lf * 2 * d;
  // plane buffers: a row-range view is the same slab layout 32 * row0 elements further on, with the same slab height
  if (v.ln_p) { v.ln_p += rows * 32; v.att_p += rows * 32; v.ffn_p += rows * 32; v.v_up_p += rows * 32; }
  if (v.f_clip_exp) { v.f_clip_exp += (size_t)b0 * c->Lf * 2; v.f_att_rs += rows; }
  return v;
}

// One shard of the batch: audio encoder on `sa`, visual encoder (+ fusion K/V projection) on `sv`, join, then
// fusion + decoder.  Clips are independent in eval mode, so shards never exchange data.
// After the join the visual stream would idle while fusion + decoder run on the audio stream (measured: one
// hardware queue 84 % busy, the other 40 %, profiles/r01c_*), so the tail is cut in two halves of the batch,
// one per stream -- same kernels, half the rows each, running concurrently.
inline void stamp(avsep_ctx* c, int idx, hipStream_t s) {
  if (c->stamps) (void)launch_stamp(c->stamps, idx, s);
}

#ifdef AVSEP_DEV
// Paired schedule (developer experiment, AVSEP_SCHEDULE=paired): only the two front-ends run side by side -- the
// LDS-resident conv stack + frame projection on `sv`, the Conv1d pair on `sa` -- then ONE chain on `sa`: every encoder
// layer as five launches that serve the audio and the visual sequence at once (encoder_layer_pair), the resize, the K/V
// projection, the fusion layers and the decoder on the whole batch: ~30 launches per step instead of 56.
// MEASURED SLOWER than the two-stream schedule (profiles/r03_ab_paired_schedule.txt, same-run A/B on one MI355X, cfg2):
// 0.490 vs 0.439 ms one step at a time, 0.365 vs 0.353 with two steps in flight; cfg3 / cfg5 equal.  A pair launch takes
// ~85 % of the time of its two halves launched one after the other (24.4 us for the 3616-row QKV against 14.3 + 13.9), but
// two streams overlap the ramp, prologue and drain of DIFFERENT kernels, which is worth more (the two chains' 364 us of
// kernel time finish in 290), and the full-batch tail on one stream (149 us) loses to two half-batch tails (139).
int forward_paired(avsep_ctx* c, const Workspace& w, const float* mixed, const float* lips, float* masks, float* sep,
                   int B, int T, int N, int H, int W, hipStream_t sa, hipStream_t sv) {
  stamp(c, 1, sv);
  int rv = visual_front(c, w, lips, B, N, H, W, sv);
  stamp(c, 2, sv);
  hipError_t ej = hipEventRecord(c->ev_vdone, sv);          // always join, even on error: never leave a capture forked
  stamp(c, 0, sa);
  int ra = audio_front(c, w, mixed, B, T, sa);
  stamp(c, 4, sa);
  hipError_t ew = hipStreamWaitEvent(sa, c->ev_vdone, 0);
  if (rv != AVSEP_OK) return rv;
  if (ra != AVSEP_OK) return ra;
  HCK(ej);
  HCK(ew);
  if (N > c->pe_len_v) return fail(AVSEP_EINVAL, "frame count exceeds PositionalEncoding max_len");
  const int d = c->d;
  for (int i = 0; i < c->Le; ++i) {
    RCK(encoder_layer_pair(c, c->a_layers[i], c->v_layers[i], w, B, T, N, sa));
    RCK(record_tap(c, w, ("a_enc" + std::to_string(i)).c_str(), w.a_x, (size_t)B * T * d, sa));
    RCK(record_tap(c, w, ("v_enc" + std::to_string(i)).c_str(), w.v_x, (size_t)B * N * d, sa));
  }
  stamp(c, 5, sa);
  RCK(visual_upsample(c, w, B, N, T, sa));
  RCK(fusion_kv(c, w, w.v_up, B, T, sa));
  stamp(c, 3, sa);
  RCK(fusion_layers(c, w, w.a_x, B, T, sa, /*final_norm=*/c->keep_taps));
  stamp(c, 7, sa);
  RCK(decoder_stage(c, w, w.a_x, masks, sep, B, T, sa, /*fuse_norm=*/true));
  stamp(c, 9, sa);
  return AVSEP_OK;
}

#endif  // AVSEP_DEV

int forward_part(avsep_ctx* c, const Workspace& w, const float* mixed, const float* lips, float* masks, float* sep,
                 int B, int T, int N, int H, int W, hipStream_t sa, hipStream_t sv) {
#ifdef AVSEP_DEV
  if (c->paired) return forward_paired(c, w, mixed, lips, masks, sep, B, T, N, H, W, sa, sv);
#endif
  stamp(c, 1, sv);
  int rv = visual_branch(c, w, lips, B, N, H, W, T, sv);
  stamp(c, 2, sv);
  if (rv == AVSEP_OK) rv = fusion_kv(c, w, w.v_up, B, T, sv);
  stamp(c, 3, sv);
  // always join, even on error, so a capture in progress is not left forked
  hipError_t ej = hipEventRecord(c->ev_vdone, sv);
  stamp(c, 0, sa);
  int ra = audio_branch(c, w, mixed, B, T, sa);
  stamp(c, 4, sa);
  hipError_t ew = hipStreamWaitEvent(sa, c->ev_vdone, 0);
  stamp(c, 5, sa);
  const bool tail_split = c->tail_split && B >= 2 && !c->keep_taps && !c->prof_on;
  hipError_t ea = hipSuccess, eb = hipSuccess;
  if (tail_split) {
    ea = hipEventRecord(c->ev_adone, sa);
    eb = hipStreamWaitEvent(sv, c->ev_adone, 0);
  }
  if (rv != AVSEP_OK) return rv;
  if (ra != AVSEP_OK) return ra;
  HCK(ej);
  HCK(ew);
  HCK(ea);
  HCK(eb);
  const size_t SF = (size_t)c->S * c->F;
  if (!tail_split) {
    RCK(fusion_layers(c, w, w.a_x, B, T, sa, /*final_norm=*/c->keep_taps));
    RCK(decoder_stage(c, w, w.a_x, masks, sep, B, T, sa, /*fuse_norm=*/true));
    return AVSEP_OK;
  }
  // clips [0,b1) on sa, [b1,B) on sv.  The two halves are enqueued layer by layer, the side stream's first: a graph
  // replay submits its nodes in capture order (~0.8 us each), and with one half captured after the other the second
  // one started ~10 us late and finished last (device stamps).
  const int b1 = (B + 1) / 2;
  if (!c->ok_fusion) return fail(AVSEP_ENOWEIGHT, "fusion weights are incomplete");
  const Workspace w1 = shift_rows(c, w, b1, T);
  stamp(c, 6, sv);
  int r0 = AVSEP_OK, r1 = AVSEP_OK;
  for (int i = 0; i < c->Lf; ++i) {
    if (r1 == AVSEP_OK) r1 = fusion_layer(c, w1, w1.a_x, B - b1, T, i, sv);
    if (r0 == AVSEP_OK) r0 = fusion_layer(c, w, w.a_x, b1, T, i, sa);
  }
  if (r1 == AVSEP_OK)
    r1 = decoder_stage(c, w1, w1.a_x, masks + (size_t)b1 * T * SF, sep + (size_t)b1 * T * SF, B - b1, T, sv, true);
  if (r0 == AVSEP_OK) r0 = decoder_stage(c, w, w.a_x, masks, sep, b1, T, sa, true);
  stamp(c, 7, sa);
  stamp(c, 8, sv);
  hipError_t et = hipEventRecord(c->ev_tdone, sv);
  hipError_t eu = hipStreamWaitEvent(sa, c->ev_tdone, 0);
  stamp(c, 9, sa);
  if (r0 != AVSEP_OK) return r0;
  if (r1 != AVSEP_OK) return r1;
  HCK(et);
  HCK(eu);
  return AVSEP_OK;
}

int forward_impl(avsep_ctx* c, const float* mixed, const float* lips, float* masks, float* sep, void* ws,
                 size_t ws_bytes, int B, int T, int N, int H, int W, hipStream_t s) {
  RCK(check_common(c, B, T));
  if (!mixed || !lips || !masks || !sep || !ws) return fail(AVSEP_EINVAL, "null tensor pointer");
  Workspace w;
  const size_t need = carve(c, &w, reinterpret_cast<float*>(ws), B, T, N, H, W) * sizeof(float);
  if (ws_bytes < need) return fail(AVSEP_ENOMEM, "workspace too small: see avsep_workspace_bytes()");
  c->taps.clear();
  c->tap_cursor = 0;
  // profiling: park the stream so the launches below are already queued when the GPU reaches them
  if (c->prof_on) HCK(launch_delay(5000, s));
  // fork the side stream off the caller's stream; forward_part joins it back on every path (also on errors), so a
  // capture in progress is never left with an un-joined stream
  HCK(hipEventRecord(c->ev_fork, s));
  hipStream_t sv = c->side;
  static const bool serial = dev_env("AVSEP_SERIAL") != nullptr;   // developer A/B: everything on one stream
  // the live profiler times every kernel alone on the chip: with two streams the 20x repeated launches of one
  // branch would overlap the other branch's and inflate both (conv_stack's LDS footprint stalls the audio GEMMs)
  if (serial || c->prof_on) sv = s;
  HCK(hipStreamWaitEvent(sv, c->ev_fork, 0));
  return forward_part(c, w, mixed, lips, masks, sep, B, T, N, H, W, s, sv);
}

// ---- static exponents of the two-term fp16 GEMM sites (gemm_h2.hip) from the packed weights alone
// exponent e with bound * 2^e <= 2^14 (one binade of slack for the roundings inside the bound itself)
int h2_exponent(double bound) {
  if (!(bound > 0.0)) return 0;
  int ex;
  (void)std::frexp(bound * (1.0 + 1e-5), &ex);                          // bound <= 2^ex
  const int e = 14 - ex;
  return e > 100 ? 100 : e < -100 ? -100 : e;
}
// conv2 / conv3 of the fused conv stack on two fp16 terms: weight planes + the constants of the per-pass activation bounds
int conv_h2_prepare(avsep_ctx* c, hipStream_t s) {
  c->conv_h2 = ConvH2{};
  if (!c->ok_visual || dev_env("AVSEP_CONV_FP32")) return AVSEP_OK;
  HCK(launch_h2_row_stats(c->c2_w, 64, 9 * 32, c->c2_ew, c->c2_l2, s));
  HCK(launch_h2_row_stats(c->c3_w, 128, 9 * 64, c->c3_ew, c->c3_l2, s));
  HCK(launch_pack_conv_h2(c->c2_w, c->c2_ew, c->c2_h, c->c2_sc, 64, 32, s));
  HCK(launch_pack_conv_h2(c->c3_w, c->c3_ew, c->c3_h, c->c3_sc, 128, 64, s));
  HCK(hipStreamSynchronize(s));
  std::vector<float> w1(9 * 32), b1(32), l2(64), b2(64);
  HCK(hipMemcpy(w1.data(), c->c1_w, w1.size() * 4, hipMemcpyDeviceToHost));
  HCK(hipMemcpy(b1.data(), c->c1_b, b1.size() * 4, hipMemcpyDeviceToHost));
  HCK(hipMemcpy(l2.data(), c->c2_l2, l2.size() * 4, hipMemcpyDeviceToHost));
  HCK(hipMemcpy(b2.data(), c->c2_b, b2.size() * 4, hipMemcpyDeviceToHost));
  double s1 = 0, b1m = 0, l2m = 0, b2m = 0;
  for (int ch = 0; ch < 32; ++ch) {
    double t = 0;
    for (int k = 0; k < 9; ++k) t += std::fabs((double)w1[k * 32 + ch]);   // packed [9][32]
    s1 = std::max(s1, t);
    b1m = std::max(b1m, std::fabs((double)b1[ch]));
  }
  for (int n = 0; n < 64; ++n) { l2m = std::max(l2m, (double)l2[n]); b2m = std::max(b2m, std::fabs((double)b2[n])); }
  if (!std::isfinite(s1) || !std::isfinite(b1m) || !std::isfinite(l2m) || !std::isfinite(b2m)) return AVSEP_OK;   // fp32 kernel
  ConvH2 h{};
  h.w2h = c->c2_h; h.w3h = c->c3_h; h.sc2 = c->c2_sc; h.sc3 = c->c3_sc;
  h.s1max = (float)(s1 * (1.0 + 1e-6)); h.b1max = (float)(b1m * (1.0 + 1e-6));
  h.l2max2 = (float)(l2m * (1.0 + 1e-6)); h.b2max = (float)(b2m * (1.0 + 1e-6));
  c->conv_h2 = h;
  return AVSEP_OK;
}

int h2_prepare(avsep_ctx* c, hipStream_t s) {
  RCK(conv_h2_prepare(c, s));
  if (c->h2.empty()) return AVSEP_OK;
  const int d = c->d;
  for (auto& kv : c->h2) HCK(launch_h2_row_stats(kv.first, kv.second.N, kv.second.K, kv.second.ew, kv.second.l2, s));
  HCK(hipStreamSynchronize(s));
  auto host = [&](const float* p, size_t n) {
    std::vector<float> v(n);
    if (hipMemcpy(v.data(), p, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) v.clear();
    return v;
  };
  bool finite = true;
  // LayerNorm(x) = xhat o gamma + beta with |xhat_k| <= sqrt(d - 1), ||xhat||_2 <= sqrt(d): element bound and 2-norm bound
  struct LnB { double elem, l2; };
  auto ln_bound = [&](const float* g, const float* be) {
    const std::vector<float> gv = host(g, d), bv = host(be, d);
    LnB r{0.0, 0.0};
    double gmax = 0.0, b2 = 0.0;
    for (int k = 0; k < d && k < (int)gv.size() && k < (int)bv.size(); ++k) {
      r.elem = std::max(r.elem, std::sqrt((double)d - 1.0) * std::fabs((double)gv[k]) + std::fabs((double)bv[k]));
      gmax = std::max(gmax, std::fabs((double)gv[k]));
      b2 += (double)bv[k] * bv[k];
    }
    r.l2 = std::sqrt((double)d) * gmax + std::sqrt(b2);
    if (gv.empty() || bv.empty() || !std::isfinite(r.elem) || !std::isfinite(r.l2)) finite = false;
    return r;
  };
  // max over the rows [n0, n1) of a projection of (l2_in ||w_n||_2 + |b_n|): bounds act(x W^T + b) for ||x||_2 <= l2_in (|relu|, |gelu| <= |.|)
  auto proj_bound = [&](const float* w, const float* b, int n0, int n1, double l2_in) {
    auto it = c->h2.find(w);
    if (it == c->h2.end()) { finite = false; return 0.0; }
    const std::vector<float> l2 = host(it->second.l2, it->second.N), bv = b ? host(b, it->second.N) : std::vector<float>(it->second.N, 0.0f);
    double r = 0.0;
    for (int n = n0; n < n1 && n < (int)l2.size() && n < (int)bv.size(); ++n) r = std::max(r, l2_in * (double)l2[n] + std::fabs((double)bv[n]));
    if (l2.empty() || bv.empty() || !std::isfinite(r)) finite = false;
    return r;
  };
  auto set = [&](const float* w, double bound) {
    auto it = c->h2.find(w);
    if (it != c->h2.end()) it->second.eA = h2_exponent(bound);
  };
  auto encoder = [&](std::vector<EncLayerW>& layers, bool ok) {
    if (!ok) return;
    for (auto& L : layers) {
      const LnB n1 = ln_bound(L.g1, L.be1), n2 = ln_bound(L.g2, L.be2);
      set(L.wqkv, n1.elem);
      const double bq = proj_bound(L.wqkv, L.bqkv, 0, d, n1.l2), bk = proj_bound(L.wqkv, L.bqkv, d, 2 * d, n1.l2),
                   bv = proj_bound(L.wqkv, L.bqkv, 2 * d, 3 * d, n1.l2);
      set(L.wo, bv);                                                        // a convex combination of value rows
      auto it = c->h2.find(L.wqkv);
      if (it != c->h2.end()) {                                              // the self-attention's operand bounds (packed q rows carry 1 / sqrt(dh))
        it->second.eq = h2_exponent(bq); it->second.ek = h2_exponent(bk); it->second.ev = h2_exponent(bv);
        it->second.qkv = true;
      }
      set(L.w1, n2.elem);
      set(L.w2, proj_bound(L.w1, L.b1, 0, 4 * d, n2.l2));
    }
  };
  encoder(c->a_layers, c->ok_audio);
  encoder(c->v_layers, c->ok_visual);
  if (c->ok_fusion) {
    c->cross_h2 = c->Lf > 0;
    std::vector<float> kc((size_t)std::max(c->Lf, 1) * 4, 0.0f);
    const std::vector<float> kvb = c->bkv_all ? host(c->bkv_all, (size_t)c->Lf * 2 * d) : std::vector<float>();
    auto kvs = c->h2.find(c->wkv_all);
    const std::vector<float> kvl2 = kvs != c->h2.end() ? host(kvs->second.l2, (size_t)c->Lf * 2 * d) : std::vector<float>();
    if (kvl2.size() != (size_t)c->Lf * 2 * d || kvb.size() != kvl2.size()) c->cross_h2 = false;
    for (size_t li = 0; li < c->f_layers.size(); ++li) {
      auto& L = c->f_layers[li];
      const LnB n1 = ln_bound(L.g1, L.be1), n2 = ln_bound(L.g2, L.be2);
      set(L.wq, n1.elem);
      set(L.w1, n2.elem);
      set(L.w2, proj_bound(L.w1, L.b1, 0, 4 * d, n2.l2));
      auto it = c->h2.find(L.wq);                                         // the cross-attention's q bound (packed q rows carry 1 / sqrt(dh))
      if (it != c->h2.end()) {
        it->second.eq = h2_exponent(proj_bound(L.wq, L.bq, 0, d, n1.l2));
        it->second.qkv = true;
        if (it->second.eq < 0) c->cross_h2 = false;
      } else {
        c->cross_h2 = false;
      }
      if (c->cross_h2)
        for (int kv = 0; kv < 2; ++kv) {                                  // K rows [li 2d, li 2d + d), V rows behind them
          double wmax = 0.0, bmax = 0.0;
          for (int n = 0; n < d; ++n) {
            const size_t r = li * 2 * d + (size_t)kv * d + n;
            wmax = std::max(wmax, (double)kvl2[r]);
            bmax = std::max(bmax, std::fabs((double)kvb[r]));
          }
          kc[li * 4 + kv * 2] = (float)(std::sqrt((double)d) * wmax * (1.0 + 1e-6));
          kc[li * 4 + kv * 2 + 1] = (float)(bmax * (1.0 + 1e-6));
          if (!std::isfinite(kc[li * 4 + kv * 2]) || !std::isfinite(kc[li * 4 + kv * 2 + 1])) finite = false;
        }
    }
    if (c->cross_h2 && c->fkv_const) HCK(hipMemcpy(c->fkv_const, kc.data(), kc.size() * sizeof(float), hipMemcpyHostToDevice));
    if (c->ok_decoder) {
      const LnB nf = ln_bound(c->fn_g, c->fn_b);
      set(c->d_w1, nf.elem);
      set(c->d_w2, proj_bound(c->d_w1, c->d_b1, 0, 2 * d, nf.l2));
    }
  }
  c->use_h2 = finite && dev_env("AVSEP_NO_H2") == nullptr;               // non-finite weights: the three-term bf16 kernels (full fp32 range)
  if (!c->use_h2) return AVSEP_OK;
  for (auto& kv : c->h2) {
    avsep_ctx::H2Site& t = kv.second;
    std::vector<int> ew(t.N);
    if (hipMemcpy(ew.data(), t.ew, t.N * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(AVSEP_EHIP, "hipMemcpy(h2 exponents)");
    std::vector<float> cs(t.N);
    for (int n = 0; n < t.N; ++n) cs[n] = std::ldexp(1.0f, -(t.eA + ew[n]));
    HCK(hipMemcpy(t.cscale, cs.data(), t.N * sizeof(float), hipMemcpyHostToDevice));
    HCK(launch_split_h2(kv.first, t.K, t.wp, t.N, t.N, t.K, t.ew, 0, s));
  }
  return AVSEP_OK;
}

// Guard for entry points that own device-wide state: makes the context's device current for the scope.
struct DeviceScope {
  int prev = -1;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
    else prev = -1;
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace

// ============================================================================================ C ABI
extern "C" void avsep_set_error_(const char* msg) { g_err = msg ? msg : ""; }   // used by train_api.hip

extern "C" {

int avsep_abi_version(void) { return AVSEP_ABI_VERSION; }
const char* avsep_last_error(void) { return g_err.c_str(); }

int avsep_create(const avsep_config* cfg, avsep_ctx** out) try {
  if (!cfg || !out) return fail(AVSEP_EINVAL, "null argument");
  *out = nullptr;
  if (cfg->freq_bins <= 0 || cfg->d_model <= 0 || cfg->nhead <= 0 || cfg->num_encoder_layers < 0 ||
      cfg->num_fusion_layers < 0 || cfg->num_speakers <= 0)
    return fail(AVSEP_EINVAL, "non-positive model dimension");
  if (cfg->d_model % 32) return fail(AVSEP_EINVAL, "d_model must be a multiple of 32 (K chunks of the fp32 MFMA GEMM)");
  if (cfg->d_model % cfg->nhead) return fail(AVSEP_EINVAL, "embed_dim must be divisible by num_heads");
  const int dh = cfg->d_model / cfg->nhead;
  if ((dh & 3) || dh > 128)
    return fail(AVSEP_EINVAL, "head dim d_model/nhead must be a multiple of 4 and <= 128");
  avsep_ctx* c = new (std::nothrow) avsep_ctx();
  if (!c) return fail(AVSEP_ENOMEM, "host allocation failed");
  c->cfg = *cfg;
  c->F = cfg->freq_bins; c->Fp = (int)align_up(cfg->freq_bins, 32); c->d = cfg->d_model; c->h = cfg->nhead;
  c->dh = dh; c->Le = cfg->num_encoder_layers; c->Lf = cfg->num_fusion_layers; c->S = cfg->num_speakers;
  // LayerNorm -> Linear sites of the d_model <= 256 models run as LayerNorm-in-the-epilogue GEMMs below 1024 64x64 tiles
  // (+5 % on the cfg2 step, profiles/r03_ab_ln_epilogue.txt).  Large problems and d_model = 512 keep LayerNorm launch +
  // big-tile GEMM: measured 2.7 % faster there, and every batch size of such a model then computes the same bits.
  // Developer A/B: AVSEP_NO_LNX=1 restores the in-kernel LayerNorm form of round 2, AVSEP_LNX=all takes every site.
  c->split_gemm = dev_env("AVSEP_NO_SPLIT") == nullptr;            // developer A/B: the fp32 MFMA GEMM everywhere
  if (const char* e = dev_env("AVSEP_SPLIT_MIN")) c->split_min = atoi(e) >= 64 ? atoi(e) : 512;   // developer A/B: the rule's floor
  c->use_lnx = dev_env("AVSEP_NO_LNX") == nullptr && (gemm_ln_supported(c->d) || dev_env("AVSEP_LNX"));
  c->lnx_all = dev_env("AVSEP_LNX") && !strcmp(dev_env("AVSEP_LNX"), "all");
  size_t off = 0;
  layout_arena(c, [&](size_t n) { off += align_up(n ? n : 1, 64); return (float*)nullptr; });
  c->arena_floats = off;
  (void)hipGetDevice(&c->device);   // the caller's current device owns this context (model.py enters torch.cuda.device)
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->arena), off * sizeof(float));
  if (e != hipSuccess) { delete c; return fail_hip(e, "hipMalloc(weight arena)"); }
  off = 0;
  layout_arena(c, [&](size_t n) { float* p = c->arena + off; off += align_up(n ? n : 1, 64); return p; });
  e = hipMemset(c->zeros, 0, ((size_t)std::max(c->Fp, c->d) + 64) * sizeof(float));
  if (e != hipSuccess) { (void)hipFree(c->arena); delete c; return fail_hip(e, "hipMemset(zero row)"); }
  bool ok = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_vdone, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_adone, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_tdone, hipEventDisableTiming) == hipSuccess;
  // developer A/B switches (libavsep_hip_dev.so only)
  if (const char* e = dev_env("AVSEP_SCHEDULE")) c->paired = strcmp(e, "paired") == 0;
  if (const char* e = dev_env("AVSEP_TAIL_SPLIT")) c->tail_split = atoi(e) != 0;
  c->no_fused_conv = dev_env("AVSEP_NO_FUSED_CONV") != nullptr;
  if (dev_env("AVSEP_STAMPS")) {
    if (hipMalloc(reinterpret_cast<void**>(&c->stamps), 16 * sizeof(unsigned long long)) != hipSuccess) c->stamps = nullptr;
    else (void)hipMemset(c->stamps, 0, 16 * sizeof(unsigned long long));
  }
  if (!ok) { avsep_destroy(c); return fail(AVSEP_EHIP, "stream/event creation failed"); }
  *out = c;
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

int avsep_create_ex(const avsep_config* cfg, uint32_t flags, avsep_ctx** out) try {
  if (flags & ~(uint32_t)AVSEP_CREATE_FP32_MATRIX) return fail(AVSEP_EINVAL, "unknown creation flag");
  RCK(avsep_create(cfg, out));
  if (flags & AVSEP_CREATE_FP32_MATRIX) (*out)->split_gemm = false;
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

void avsep_destroy(avsep_ctx* c) {
  if (!c) return;
  // Work of this context may still be in flight: on its own two streams, and on every caller stream a graph was
  // replayed on.  Drain exactly those, on the context's own device (the caller's current device may be another GPU).
  DeviceScope guard(c->device);
  if (c->side) (void)hipStreamSynchronize(c->side);
  for (auto& g : c->graphs) (void)hipStreamSynchronize(g.last_stream);
  (void)hipDeviceSynchronize();   // eager forwards ran on caller streams this context keeps no record of
  for (auto& g : c->graphs) {
    (void)hipGraphExecDestroy(g.exec);
    (void)hipStreamDestroy(g.cap);
    (void)hipStreamDestroy(g.side);
  }
  for (auto& r : c->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
#ifdef AVSEP_DEV
  for (auto& e : c->chains) chain_plan_free(e.plan);
#endif
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_vdone) (void)hipEventDestroy(c->ev_vdone);
  if (c->ev_adone) (void)hipEventDestroy(c->ev_adone);
  if (c->ev_tdone) (void)hipEventDestroy(c->ev_tdone);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->arena) (void)hipFree(c->arena);
  if (c->stamps) (void)hipFree(c->stamps);
  delete c;
}

// Split-precision kernels on (the default) or off for this context: see include/avsep.h.  Captured graphs hold the kernels of
// the previous setting: drain and drop them.
int avsep_set_split_precision(avsep_ctx* c, int enable) try {
  if (!c) return fail(AVSEP_EINVAL, "null context");
  const bool on = enable != 0;
  if (on == c->split_gemm) return AVSEP_OK;
  DeviceScope guard(c->device);
  if (c->side) (void)hipStreamSynchronize(c->side);
  for (auto& g : c->graphs) (void)hipStreamSynchronize(g.last_stream);
  (void)hipDeviceSynchronize();
  for (auto& g : c->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipStreamDestroy(g.cap); (void)hipStreamDestroy(g.side); }
  c->graphs.clear();
  c->split_gemm = on;
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

#ifdef AVSEP_DEV
int avsep_set_schedule(avsep_ctx* c, int schedule, int group, float skew) try {
  if (!c) return fail(AVSEP_EINVAL, "null context");
  if (schedule < 0 || schedule > 2)
    return fail(AVSEP_EINVAL, "schedule: 0 = one launch per op, 1 = chained encoder layers (one queue), 2 = chained, XCD-local queues");
  if (group < 0 || !(skew >= 0.0f) || skew > 64.0f) return fail(AVSEP_EINVAL, "bad work-list order");
  DeviceScope guard(c->device);
  // plans and captured graphs of the previous schedule may be in flight: drain, then drop them
  if (c->side) (void)hipStreamSynchronize(c->side);
  for (auto& g : c->graphs) (void)hipStreamSynchronize(g.last_stream);
  (void)hipDeviceSynchronize();
  for (auto& g : c->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipStreamDestroy(g.cap); (void)hipStreamDestroy(g.side); }
  c->graphs.clear();
  for (auto& e : c->chains) chain_plan_free(e.plan);
  c->chains.clear();
  c->schedule = schedule; c->chain_group = group; c->chain_skew = skew;
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

int avsep_chain_status(avsep_ctx* c, void* stream) try {
  if (!c) return fail(AVSEP_EINVAL, "null context");
  int bad = 0;
  for (auto& e : c->chains) {
    unsigned word = 0;
    HCK(chain_plan_error(e.plan, reinterpret_cast<hipStream_t>(stream), &word));
    if (word) { bad = (int)word; g_err = "a chained launch gave up waiting for a producer tile (ticket " + std::to_string(word - 1) + ")"; }
  }
  return bad ? AVSEP_EINTERNAL : AVSEP_OK;
} catch (...) {
  return on_exception();
}

// developer aid: the first n state words (ticket head, error word, 2 unused, counters...) of chained plan `idx`, copied on a
// stream of its own while the launch may still be running
int avsep_chain_peek(avsep_ctx* c, int idx, unsigned* out, int n) try {
  if (!c || idx < 0 || idx >= (int)c->chains.size() || !out || n <= 0) return fail(AVSEP_EINVAL, "bad argument");
  hipStream_t s = nullptr;
  HCK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipError_t e = chain_plan_peek(c->chains[idx].plan, s, out, n);
  (void)hipStreamDestroy(s);
  HCK(e);
  return (int)c->chains.size();
} catch (...) {
  return on_exception();
}
#endif  // AVSEP_DEV

int avsep_profile_begin(avsep_ctx* c) try {
  if (!c) return fail(AVSEP_EINVAL, "null context");
  for (auto& r : c->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  c->prof.clear();
  c->prof_on = true;
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

// Stops profiling, waits for the recorded events and writes one JSON array, aggregated per kernel name in
// first-launch order: [{"name":..,"calls":n,"ms":total,"flops":total,"bytes":total}, ...].
// Returns the number of bytes written (excluding the NUL) or a negative error.
int64_t avsep_profile_end(avsep_ctx* c, char* json, size_t cap) try {
  if (!c || !json || cap < 3) return fail(AVSEP_EINVAL, "bad argument");
  c->prof_on = false;
  struct Agg { std::string name; long calls; double ms, flops, bytes; };
  std::vector<Agg> agg;
  int rc = AVSEP_OK;
  for (auto& r : c->prof) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(r.e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.e0, r.e1);
    if (e != hipSuccess && rc == AVSEP_OK) rc = fail_hip(e, "hipEventElapsedTime");
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
    Agg* a = nullptr;
    for (auto& x : agg) if (x.name == r.name) a = &x;
    if (!a) { agg.push_back({r.name, 0, 0, 0, 0}); a = &agg.back(); }
    a->calls++; a->ms += ms / PROF_REPS; a->flops += r.flops; a->bytes += r.bytes;
  }
  c->prof.clear();
  if (rc != AVSEP_OK) return rc;
  std::string out = "[";
  char line[384];
  for (size_t i = 0; i < agg.size(); ++i) {
    snprintf(line, sizeof line, "%s{\"name\":\"%s\",\"calls\":%ld,\"ms\":%.6f,\"flops\":%.0f,\"bytes\":%.0f}",
             i ? "," : "", agg[i].name.c_str(), agg[i].calls, agg[i].ms, agg[i].flops, agg[i].bytes);
    out += line;
  }
  out += "]";
  if (out.size() + 1 > cap) return fail(AVSEP_ENOMEM, "profile buffer too small");
  memcpy(json, out.c_str(), out.size() + 1);
  return (int64_t)out.size();
} catch (...) {
  return on_exception();
}

int avsep_read_stamps(avsep_ctx* c, uint64_t* out, int n) try {
  if (!c || !out || n <= 0) return fail(AVSEP_EINVAL, "bad argument");
  if (!c->stamps) return fail(AVSEP_EINVAL, "stamps are off: set AVSEP_STAMPS=1 before avsep_create()");
  if (n > 16) n = 16;
  HCK(hipMemcpy(out, c->stamps, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return n;
} catch (...) {
  return on_exception();
}

int avsep_set_debug_taps(avsep_ctx* c, int on) try {
  if (!c) return fail(AVSEP_EINVAL, "null context");
  c->keep_taps = on != 0;
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

int avsep_set_weight(avsep_ctx* c, const char* key, const float* dev_ptr, const int64_t* shape, int ndim) try {
  if (!c || !key || !dev_ptr || ndim < 0 || ndim > 4 || (ndim > 0 && !shape)) return fail(AVSEP_EINVAL, "bad argument");
  RawW w;
  w.ptr = dev_ptr;
  w.shape.assign(shape, shape + ndim);
  c->raw[key] = std::move(w);
  c->finalized = false;
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

int avsep_finalize_weights(avsep_ctx* c, void* stream) try {
  if (!c) return fail(AVSEP_EINVAL, "null context");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int d = c->d, F = c->F, S = c->S;
  // replayed graphs bake in nothing about weights (the arena address is stable), so they stay valid
  {  // ---- audio encoder
    Packer P{c, s};
    if (const float* w = P.need("audio_encoder.input_proj.0.weight", {d, F, 3}))
      P.keep(launch_pack_conv1d(w, c->a_w1, d, F, c->Fp, s));
    P.copy("audio_encoder.input_proj.0.bias", c->a_b1, d);
    if (const float* w = P.need("audio_encoder.input_proj.2.weight", {d, d, 3}))
      P.keep(launch_pack_conv1d(w, c->a_w2, d, d, d, s));
    P.copy("audio_encoder.input_proj.2.bias", c->a_b2, d);
    RCK(pack_pe(P, "audio_encoder.pos_enc.pe", c->a_pe, &c->pe_len_a));
    pack_encoder(P, "audio_encoder.", c->a_layers);
    HCK(P.err);
    c->ok_audio = !P.missing;
  }
  {  // ---- visual encoder
    Packer P{c, s};
    const int chans[4] = {1, 32, 64, 128};
    float* wp[3] = {c->c1_w, c->c2_w, c->c3_w};
    float* bp[3] = {c->c1_b, c->c2_b, c->c3_b};
    for (int i = 0; i < 3; ++i) {
      const std::string cv = "visual_encoder.conv." + std::to_string(3 * i) + ".";
      const std::string bn = "visual_encoder.conv." + std::to_string(3 * i + 1) + ".";
      const int co = chans[i + 1], ci = chans[i];
      const float* w = P.need(cv + "weight", {co, ci, 3, 3});
      const float* b = P.need(cv + "bias", {co});
      const float* g = P.need(bn + "weight", {co});
      const float* be = P.need(bn + "bias", {co});
      const float* mu = P.need(bn + "running_mean", {co});
      const float* var = P.need(bn + "running_var", {co});
      if (w && b && g && be && mu && var)
        P.keep(launch_pack_conv2d_bn(w, b, g, be, mu, var, wp[i], bp[i], co, ci, 1e-5f, s));
    }
    P.rows("visual_encoder.frame_proj.weight", c->fp_w, d, 128);
    P.copy("visual_encoder.frame_proj.bias", c->fp_b, d);
    RCK(pack_pe(P, "visual_encoder.pos_enc.pe", c->v_pe, &c->pe_len_v));
    pack_encoder(P, "visual_encoder.", c->v_layers);
    HCK(P.err);
    c->ok_visual = !P.missing;
  }
  {  // ---- fusion
    Packer P{c, s};
    const float qs = 1.0f / std::sqrt((float)c->dh);
    for (int i = 0; i < c->Lf; ++i) {
      const std::string p = "fusion.layers." + std::to_string(i) + ".";
      FusLayerW& L = c->f_layers[i];
      if (const float* w = P.need(p + "cross_attn.in_proj_weight", {3 * d, d})) {
        P.keep(launch_pack_rows(w, L.wq, d, d, d, qs, d, s));                                  // Wq * 1/sqrt(dh)
        P.keep(launch_pack_rows(w + (size_t)d * d, c->wkv_all + (size_t)i * 2 * d * d, 2 * d, d, d, 1.0f, 0, s));
      }
      if (const float* b = P.need(p + "cross_attn.in_proj_bias", {3 * d})) {
        P.keep(launch_scale_copy(b, L.bq, d, qs, d, s));
        P.keep(launch_scale_copy(b + d, c->bkv_all + (size_t)i * 2 * d, 2 * d, 1.0f, 0, s));
      }
      P.rows(p + "cross_attn.out_proj.weight", L.wo, d, d);
      P.copy(p + "cross_attn.out_proj.bias", L.bo, d);
      P.rows(p + "ff.0.weight", L.w1, 4 * d, d);
      P.copy(p + "ff.0.bias", L.b1, 4 * d);
      P.rows(p + "ff.3.weight", L.w2, d, 4 * d);
      P.copy(p + "ff.3.bias", L.b2, d);
      P.copy(p + "norm1.weight", L.g1, d);
      P.copy(p + "norm1.bias", L.be1, d);
      P.copy(p + "norm2.weight", L.g2, d);
      P.copy(p + "norm2.bias", L.be2, d);
    }
    P.copy("fusion.norm.weight", c->fn_g, d);
    P.copy("fusion.norm.bias", c->fn_b, d);
    HCK(P.err);
    c->ok_fusion = !P.missing;
  }
  {  // ---- decoder
    Packer P{c, s};
    P.rows("decoder.decoder.0.weight", c->d_w1, 2 * d, d);
    P.copy("decoder.decoder.0.bias", c->d_b1, 2 * d);
    P.rows("decoder.decoder.3.weight", c->d_w2, (int64_t)S * F, 2 * d);
    P.copy("decoder.decoder.3.bias", c->d_b2, (int64_t)S * F);
    HCK(P.err);
    c->ok_decoder = !P.missing;
  }
  if (!(c->ok_audio || c->ok_visual || c->ok_fusion || c->ok_decoder))
    return fail(AVSEP_ENOWEIGHT, "no stage has a complete set of weights");
  if (c->use_lnx) {   // W o gamma, its row sums, W beta + b for every LayerNorm -> Linear site (after the packs above, same stream)
    auto site = [&](const float* w, const float* b, const float* g, const float* be, int n) {
      auto it = c->lnx.find(w);
      if (it == c->lnx.end()) return hipErrorInvalidValue;
      return launch_pack_lnx(w, b, g, be, it->second.w, it->second.c1, it->second.c2, n, d, s);
    };
    auto enc = [&](std::vector<EncLayerW>& v) {
      for (auto& L : v) {
        HCK(site(L.wqkv, L.bqkv, L.g1, L.be1, 3 * d));
        HCK(site(L.w1, L.b1, L.g2, L.be2, 4 * d));
      }
      return (int)AVSEP_OK;
    };
    if (c->ok_audio) RCK(enc(c->a_layers));
    if (c->ok_visual) RCK(enc(c->v_layers));
    if (c->ok_fusion) {
      for (auto& L : c->f_layers) {
        HCK(site(L.wq, L.bq, L.g1, L.be1, d));
        HCK(site(L.w1, L.b1, L.g2, L.be2, 4 * d));
      }
      if (c->ok_decoder) HCK(site(c->d_w1, c->d_b1, c->fn_g, c->fn_b, 2 * d));
    }
  }
  // the bf16 planes of every weight the split-precision GEMM takes (after the packs above, same stream): gemm_planes.hip
  for (const auto& site : c->wplane_sites)
    HCK(launch_split_planes(site.first, site.second.second, c->wplanes[site.first], site.second.first, site.second.first,
                            site.second.second, s));
  RCK(h2_prepare(c, s));
  c->finalized = true;
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

size_t avsep_workspace_bytes(const avsep_ctx* c, int B, int T, int N, int H, int W) {
  if (!c || B <= 0 || T <= 0 || N <= 0 || H <= 0 || W <= 0) return 0;
  return carve(c, nullptr, nullptr, B, T, N, H, W) * sizeof(float);
}

int avsep_forward(avsep_ctx* c, const float* mixed, const float* lips, float* masks, float* sep, void* ws,
                  size_t ws_bytes, int B, int T, int N, int H, int W, void* stream) try {
  return forward_impl(c, mixed, lips, masks, sep, ws, ws_bytes, B, T, N, H, W, reinterpret_cast<hipStream_t>(stream));
} catch (...) {
  return on_exception();
}

int avsep_forward_graph(avsep_ctx* c, const float* mixed, const float* lips, float* masks, float* sep, void* ws,
                        size_t ws_bytes, int B, int T, int N, int H, int W, void* stream) try {
  RCK(check_common(c, B, T));
  if (c->keep_taps || c->prof_on) return fail(AVSEP_EINVAL, "debug taps / profiler are not available under graph replay");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  GraphEntry* hit = nullptr;
  for (auto& g : c->graphs)
    if (g.mixed == mixed && g.lips == lips && g.masks == masks && g.sep == sep && g.ws == ws &&
        g.ws_bytes == ws_bytes && g.B == B && g.T == T && g.N == N && g.H == H && g.W == W)
      hit = &g;
  if (!hit) {
    // capture on streams of this graph's own (the caller's may be the legacy stream, which cannot capture; and see
    // GraphEntry for why not the context's)
    GraphEntry g{mixed, lips, masks, sep, ws, ws_bytes, B, T, N, H, W, nullptr, s, nullptr, nullptr};
    HCK(hipStreamCreateWithFlags(&g.cap, hipStreamNonBlocking));
    hipError_t e = hipStreamCreateWithFlags(&g.side, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipStreamDestroy(g.cap); return fail_hip(e, "hipStreamCreate(capture side stream)"); }
    auto drop_streams = [&] { (void)hipStreamDestroy(g.cap); (void)hipStreamDestroy(g.side); };
    hipGraph_t graph = nullptr;
#ifdef AVSEP_DEV
    if (c->schedule != 0) {      // the chain plans allocate device tables: build them BEFORE the capture starts
      Workspace w;
      if (ws_bytes >= carve(c, &w, reinterpret_cast<float*>(ws), B, T, N, H, W) * sizeof(float)) {
        ChainPlanImpl* plan = nullptr;
        int r = AVSEP_OK;
        if (chain_usable(c, c->a_layers, B, T)) r = get_chain(c, c->a_layers, w.a_x, w.qkv, w.att, w.ffn, B, T, &plan);
        if (r == AVSEP_OK && chain_usable(c, c->v_layers, B, N)) r = get_chain(c, c->v_layers, w.v_x, w.v_qkv, w.v_att, w.v_ffn, B, N, &plan);
        if (r != AVSEP_OK) { drop_streams(); return r; }
      }
    }
#endif
    e = hipStreamBeginCapture(g.cap, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { drop_streams(); return fail_hip(e, "hipStreamBeginCapture"); }
    hipStream_t ctx_side = c->side;
    c->side = g.side;
    int r = forward_impl(c, mixed, lips, masks, sep, ws, ws_bytes, B, T, N, H, W, g.cap);
    c->side = ctx_side;
    e = hipStreamEndCapture(g.cap, &graph);
    if (r != AVSEP_OK) { if (graph) (void)hipGraphDestroy(graph); drop_streams(); return r; }
    if (e != hipSuccess) { drop_streams(); return fail_hip(e, "hipStreamEndCapture"); }
    e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { drop_streams(); return fail_hip(e, "hipGraphInstantiate"); }
    if (c->graphs.size() >= 8) {
      // the evicted graph may still be replaying on the stream it was last launched on (not necessarily this
      // caller's): let it drain there before its nodes are freed
      GraphEntry& old = c->graphs.front();
      (void)hipStreamSynchronize(old.last_stream);
      (void)hipGraphExecDestroy(old.exec);
      (void)hipStreamDestroy(old.cap);
      (void)hipStreamDestroy(old.side);
      c->graphs.erase(c->graphs.begin());
    }
    c->graphs.push_back(g);
    hit = &c->graphs.back();
  }
  // Replay on the caller's own stream: no event fences between consecutive steps (the fenced hand-off to the
  // capture stream left a ~40-60 us bubble per step).  Only capture needs a non-legacy stream, launch does not.
  HCK(hipGraphLaunch(hit->exec, s));
  hit->last_stream = s;
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

int avsep_audio_encoder(avsep_ctx* c, const float* mixed, float* out, void* ws, size_t ws_bytes, int B, int T,
                        void* stream) try {
  RCK(check_common(c, B, T));
  if (!mixed || !out || !ws) return fail(AVSEP_EINVAL, "null tensor pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Workspace w;
  if (ws_bytes < carve(c, &w, reinterpret_cast<float*>(ws), B, T, 1, 1, 1) * sizeof(float))
    return fail(AVSEP_ENOMEM, "workspace too small");
  c->taps.clear(); c->tap_cursor = 0;
  RCK(audio_branch(c, w, mixed, B, T, s));
  HCK(hipMemcpyAsync(out, w.a_x, (size_t)B * T * c->d * sizeof(float), hipMemcpyDeviceToDevice, s));
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

int avsep_visual_encoder(avsep_ctx* c, const float* lips, float* out, void* ws, size_t ws_bytes, int B, int N, int H,
                         int W, int T, void* stream) try {
  RCK(check_common(c, B, T));
  if (!lips || !out || !ws) return fail(AVSEP_EINVAL, "null tensor pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Workspace w;
  if (ws_bytes < carve(c, &w, reinterpret_cast<float*>(ws), B, T, N, H, W) * sizeof(float))
    return fail(AVSEP_ENOMEM, "workspace too small");
  c->taps.clear(); c->tap_cursor = 0;
  RCK(visual_branch(c, w, lips, B, N, H, W, T, s));
  HCK(hipMemcpyAsync(out, w.v_up, (size_t)B * T * c->d * sizeof(float), hipMemcpyDeviceToDevice, s));
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

int avsep_fusion(avsep_ctx* c, const float* audio, const float* visual, float* out, void* ws, size_t ws_bytes, int B,
                 int T, void* stream) try {
  RCK(check_common(c, B, T));
  if (!audio || !visual || !out || !ws) return fail(AVSEP_EINVAL, "null tensor pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Workspace w;
  if (ws_bytes < carve(c, &w, reinterpret_cast<float*>(ws), B, T, 1, 1, 1) * sizeof(float))
    return fail(AVSEP_ENOMEM, "workspace too small");
  c->taps.clear(); c->tap_cursor = 0;
  HCK(hipMemcpyAsync(w.a_x, audio, (size_t)B * T * c->d * sizeof(float), hipMemcpyDeviceToDevice, s));
  RCK(fusion_kv(c, w, visual, B, T, s));
  RCK(fusion_layers(c, w, w.a_x, B, T, s, /*final_norm=*/true));
  HCK(hipMemcpyAsync(out, w.ln, (size_t)B * T * c->d * sizeof(float), hipMemcpyDeviceToDevice, s));
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

int avsep_decoder(avsep_ctx* c, const float* fused, const float* mixed, float* masks, float* sep, void* ws,
                  size_t ws_bytes, int B, int T, void* stream) try {
  RCK(check_common(c, B, T));
  if (!fused || !masks || !ws) return fail(AVSEP_EINVAL, "null tensor pointer");
  if ((mixed == nullptr) != (sep == nullptr)) return fail(AVSEP_EINVAL, "mixed and separated must be given together");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Workspace w;
  if (ws_bytes < carve(c, &w, reinterpret_cast<float*>(ws), B, T, 1, 1, 1) * sizeof(float))
    return fail(AVSEP_ENOMEM, "workspace too small");
  if (mixed) HCK(launch_transpose_pad(mixed, w.xt, B, c->F, T, c->Fp, s));
  RCK(decoder_stage(c, w, fused, masks, sep, B, T, s, /*fuse_norm=*/false));
  return AVSEP_OK;
} catch (...) {
  return on_exception();
}

int64_t avsep_read_tap(avsep_ctx* c, const char* name, float* dst, int64_t max_floats, void* ws, int B, int T,
                          int N, int H, int W, void* stream) try {
  if (!c || !name || !dst || !ws) return fail(AVSEP_EINVAL, "bad argument");
  if (!c->keep_taps) return fail(AVSEP_ESTATE, "debug taps are off (avsep_set_debug_taps)");
  Workspace w;
  carve(c, &w, reinterpret_cast<float*>(ws), B, T, N, H, W);
  for (const auto& t : c->taps)
    if (t.name == name) {
      if ((int64_t)t.n > max_floats) return fail(AVSEP_ENOMEM, "destination too small for tap");
      hipError_t e = hipMemcpyAsync(dst, w.taps + t.off, t.n * sizeof(float), hipMemcpyDeviceToDevice,
                                    reinterpret_cast<hipStream_t>(stream));
      if (e != hipSuccess) return fail_hip(e, "hipMemcpyAsync(tap)");
      return (int64_t)t.n;
    }
  return fail(AVSEP_EINVAL, std::string("unknown tap: ") + name);
} catch (...) {
  return on_exception();
}

// ---------------------------------------------------------------------------------- single-kernel entry points
int avsep_op_linear(const float* x, const float* w, const float* bias, const float* residual, float* y, int M, int N,
                    int K, int act, void* stream) {
  if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0) return fail(AVSEP_EINVAL, "bad argument");
  if (K % 32) return fail(AVSEP_EINVAL, "K must be a multiple of 32");
  if (act < 0 || act > 3) return fail(AVSEP_EINVAL, "unknown activation");
  GemmParams p = linear_params(x, K, w, K, bias, y, N, M, N, act);
  if (residual) { p.R = residual; p.ldr = N; }
  HCK(launch_gemm(p, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_linear_split(const float* x, const float* w, const float* bias, const float* residual, float* y, int M, int N,
                          int K, int act, void* stream) {
  if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0) return fail(AVSEP_EINVAL, "bad argument");
  if (K % 32 || N % 4) return fail(AVSEP_EINVAL, "K must be a multiple of 32 and N of 4");
  if (act < 0 || act > 3) return fail(AVSEP_EINVAL, "unknown activation");
  GemmParams p = linear_params(x, K, w, K, bias, y, N, M, N, act);
  if (residual) { p.R = residual; p.ldr = N; }
  if (!gemm_split_supported(p)) return fail(AVSEP_EINVAL, "shape not supported by the split-precision GEMM");
  HCK(launch_gemm_split(p, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_split_planes(const float* x, int ld, uint16_t* planes, int64_t rows, int M, int K, void* stream) {
  if (!x || !planes || M <= 0 || K <= 0) return fail(AVSEP_EINVAL, "bad argument");
  if (K % 32 || ld % 4 || ld < K || rows < M) return fail(AVSEP_EINVAL, "K must be a multiple of 32, ld of 4 and >= K, rows >= M");
  HCK(launch_split_planes(x, ld, planes, rows, M, K, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_linear_planes(const uint16_t* xp, int64_t x_rows, const uint16_t* wp, int64_t w_rows, const float* bias,
                           const float* residual, float* y, uint16_t* yp, int64_t y_rows, int M, int N, int K, int act,
                           void* stream) {
  if (!xp || !wp || (!y && !yp) || M <= 0 || N <= 0 || K <= 0) return fail(AVSEP_EINVAL, "bad argument");
  if (K % 32 || N % 4) return fail(AVSEP_EINVAL, "K must be a multiple of 32 and N of 4");
  if (act < 0 || act > 3) return fail(AVSEP_EINVAL, "unknown activation");
  GemmParams p = linear_params(nullptr, K, nullptr, K, bias, y, N, M, N, act);
  p.Ap = xp; p.a_rows = x_rows; p.Wp = wp; p.w_rows = w_rows; p.Cp = yp; p.c_rows = y_rows;
  if (residual) { p.R = residual; p.ldr = N; }
  if (!gemm_planes_supported(p)) return fail(AVSEP_EINVAL, "shape / epilogue not supported by the pre-split GEMM");
  HCK(launch_gemm_planes(p, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_layernorm_planes(const float* x, const float* gamma, const float* beta, uint16_t* yp, int64_t rows, int M, int d, float eps,
                              void* stream) {
  if (!x || !gamma || !beta || !yp) return fail(AVSEP_EINVAL, "null pointer");
  if (M <= 0 || d <= 0 || d % 32 || d > 2048 || rows < M) return fail(AVSEP_EINVAL, "d must be a multiple of 32 and <= 2048, rows >= M");
  HCK(launch_layernorm_planes(x, gamma, beta, yp, rows, M, d, eps, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_interp_linear_planes(const float* x, uint16_t* yp, int64_t rows, int B, int N, int T, int d, void* stream) {
  if (!x || !yp || B <= 0 || N <= 0 || T <= 0 || d <= 0 || d % 32 || rows < (int64_t)B * T) return fail(AVSEP_EINVAL, "bad argument");
  HCK(launch_interp_linear_planes(x, yp, rows, B, N, T, d, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_attention_split_planes(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, uint16_t* op, int64_t rows,
                                    int B, int nhead, int dh, int Lq, int Lk, void* stream) {
  if (!q || !k || !v || !op) return fail(AVSEP_EINVAL, "null pointer");
  if (!attention_split_supported(dh, Lq, Lk) || B <= 0 || nhead <= 0 || rows < (int64_t)B * Lq) return fail(AVSEP_EINVAL, "unsupported shape");
  HCK(launch_attention_split(q, ldq, k, ldk, v, ldv, nullptr, 0, B, nhead, dh, Lq, Lk, 1.0f, reinterpret_cast<hipStream_t>(stream), op, rows));
  return AVSEP_OK;
}

int avsep_op_h2_row_stats(const float* w, int N, int K, int32_t* ew, float* l2, void* stream) {
  if (!w || !ew || !l2 || N <= 0 || K <= 0) return fail(AVSEP_EINVAL, "bad argument");
  HCK(launch_h2_row_stats(w, N, K, reinterpret_cast<int*>(ew), l2, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_split_h2(const float* x, int ld, uint16_t* planes, int64_t rows, int M, int K, const int32_t* row_exp, int e, void* stream) {
  if (!x || !planes || M <= 0 || K <= 0) return fail(AVSEP_EINVAL, "bad argument");
  if (K % 32 || ld % 4 || ld < K || rows < M || e < -120 || e > 120) return fail(AVSEP_EINVAL, "K must be a multiple of 32, ld of 4 and >= K, rows >= M, |e| <= 120");
  HCK(launch_split_h2(x, ld, planes, rows, M, K, reinterpret_cast<const int*>(row_exp), e, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_interp_linear_h2(const float* x, uint16_t* yp, float* rscale, int64_t rows, int B, int N, int T, int d, void* stream) {
  if (!x || !yp || !rscale || B <= 0 || N <= 0 || T <= 0 || d <= 0 || d % 32 || d > 2048 || rows < (int64_t)B * T) return fail(AVSEP_EINVAL, "bad argument");
  HCK(launch_interp_linear_h2(x, yp, rscale, rows, B, N, T, d, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_linear_h2(const uint16_t* xp, int64_t x_rows, const uint16_t* wp, int64_t w_rows, const float* cscale, const float* rscale,
                       const float* bias, const float* residual, float* y, uint16_t* yp, int64_t y_rows, int yp_exp, int M, int N, int K,
                       int act, void* stream) {
  if (!xp || !wp || !cscale || (!y && !yp) || M <= 0 || N <= 0 || K <= 0) return fail(AVSEP_EINVAL, "bad argument");
  if (K % 32 || N % 2) return fail(AVSEP_EINVAL, "K must be a multiple of 32 and N even");
  if (act < 0 || act > 3 || yp_exp < -120 || yp_exp > 120) return fail(AVSEP_EINVAL, "unknown activation / exponent out of range");
  GemmParams p = linear_params(nullptr, K, nullptr, K, bias, y, N, M, N, act);
  p.Ap = xp; p.a_rows = x_rows; p.Wp = wp; p.w_rows = w_rows; p.Cp = yp; p.c_rows = y_rows; p.cscale = cscale; p.rscale = rscale; p.h2 = 1;
  p.cp_scale = std::ldexp(1.0f, yp_exp);
  if (residual) { p.R = residual; p.ldr = N; }
  if (!gemm_h2_supported(p)) return fail(AVSEP_EINVAL, "shape / epilogue not supported by the two-term GEMM");
  HCK(launch_gemm_h2(p, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_layernorm_h2(const float* x, const float* gamma, const float* beta, uint16_t* yp, int64_t rows, int M, int d, float eps, int e,
                          void* stream) {
  if (!x || !gamma || !beta || !yp) return fail(AVSEP_EINVAL, "null pointer");
  if (M <= 0 || d <= 0 || d % 32 || d > 2048 || rows < M || e < -120 || e > 120) return fail(AVSEP_EINVAL, "bad argument");
  HCK(launch_layernorm_h2(x, gamma, beta, yp, rows, M, d, eps, e, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_attention_split_h2(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, uint16_t* op, int64_t rows,
                                int e, int B, int nhead, int dh, int Lq, int Lk, void* stream) {
  if (!q || !k || !v || !op) return fail(AVSEP_EINVAL, "null pointer");
  if (!attention_split_supported(dh, Lq, Lk) || B <= 0 || nhead <= 0 || rows < (int64_t)B * Lq || e < -120 || e > 120)
    return fail(AVSEP_EINVAL, "unsupported shape");
  HCK(launch_attention_split(q, ldq, k, ldk, v, ldv, nullptr, 0, B, nhead, dh, Lq, Lk, 1.0f, reinterpret_cast<hipStream_t>(stream), op, rows,
                             1, e));
  return AVSEP_OK;
}

#ifdef AVSEP_DEV
int avsep_op_linear_pair(const float* x0, const float* w0, const float* b0, const float* r0, const float* gamma0,
                         const float* beta0, float* y0, int M0, const float* x1, const float* w1, const float* b1,
                         const float* r1, const float* gamma1, const float* beta1, float* y1, int M1, int N, int K,
                         int act, float eps, void* stream) {
  if (!x0 || !w0 || !y0 || !x1 || !w1 || !y1 || M0 <= 0 || M1 <= 0 || N <= 0 || K <= 0) return fail(AVSEP_EINVAL, "bad argument");
  if (K % 32) return fail(AVSEP_EINVAL, "K must be a multiple of 32");
  if (act < 0 || act > 3) return fail(AVSEP_EINVAL, "unknown activation");
  if ((b0 == nullptr) != (b1 == nullptr) || (r0 == nullptr) != (r1 == nullptr) || (gamma0 == nullptr) != (gamma1 == nullptr) ||
      (gamma0 == nullptr) != (beta0 == nullptr) || (gamma1 == nullptr) != (beta1 == nullptr))
    return fail(AVSEP_EINVAL, "bias / residual / LayerNorm vectors must be given for both problems or for neither");
  if (gamma0 && !gemm_ln_supported(K)) return fail(AVSEP_EINVAL, "in-kernel LayerNorm form: K is not supported");
  GemmParams p0 = linear_params(x0, K, w0, K, b0, y0, N, M0, N, act);
  GemmParams p1 = linear_params(x1, K, w1, K, b1, y1, N, M1, N, act);
  if (r0) { p0.R = r0; p0.ldr = N; p1.R = r1; p1.ldr = N; }
  if (gamma0) {
    p0.ln_gamma = gamma0; p0.ln_beta = beta0; p0.ln_eps = eps;
    p1.ln_gamma = gamma1; p1.ln_beta = beta1; p1.ln_eps = eps;
  }
  HCK(launch_gemm_pair(p0, p1, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_attention_pair(const float* q0, const float* k0, const float* v0, float* o0, int ldqkv0, int ldo0, int B0,
                            int L0, const float* q1, const float* k1, const float* v1, float* o1, int ldqkv1, int ldo1,
                            int B1, int L1, int nhead, int dh, void* stream) {
  if (!q0 || !k0 || !v0 || !o0 || !q1 || !k1 || !v1 || !o1) return fail(AVSEP_EINVAL, "null pointer");
  const AttnProblem a{q0, k0, v0, o0, ldqkv0, ldqkv0, ldqkv0, ldo0, B0, L0, L0};
  const AttnProblem b{q1, k1, v1, o1, ldqkv1, ldqkv1, ldqkv1, ldo1, B1, L1, L1};
  HCK(launch_attention_pair(a, b, nhead, dh, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}
#endif  // AVSEP_DEV

int avsep_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int M, int d, float eps,
                       void* stream) {
  if (!x || !gamma || !beta || !y) return fail(AVSEP_EINVAL, "null pointer");
  HCK(launch_layernorm(x, gamma, beta, y, M, d, eps, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_ln_linear(const float* x, const float* gamma, const float* beta, const float* w, const float* bias,
                       float* y, float* scratch, int M, int N, int K, int act, float eps, int form, void* stream) {
  if (!x || !gamma || !beta || !w || !y || M <= 0 || N <= 0 || K <= 0) return fail(AVSEP_EINVAL, "bad argument");
  if (K % 32) return fail(AVSEP_EINVAL, "K must be a multiple of 32");
  if (act < 0 || act > 3) return fail(AVSEP_EINVAL, "unknown activation");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  GemmParams p = linear_params(x, K, w, K, bias, y, N, M, N, act);
  if (form == 0) {                       // LayerNorm launch, then the plain GEMM on the normalised rows
    if (!scratch) return fail(AVSEP_EINVAL, "form 0 needs M*K floats of scratch");
    HCK(launch_layernorm(x, gamma, beta, scratch, M, K, eps, s));
    p.A = scratch;
  } else if (form == 1) {                // statistics and normalisation inside the GEMM (short K)
    if (!gemm_ln_supported(K)) return fail(AVSEP_EINVAL, "in-kernel LayerNorm form: K is not supported");
    p.ln_gamma = gamma; p.ln_beta = beta; p.ln_eps = eps;
  } else if (form == 2) {                // statistics launch, normalisation while the GEMM stages A
#ifndef AVSEP_DEV
    return fail(AVSEP_EINVAL, "form 2 (LayerNorm applied while the GEMM stages A) is a developer instance: measured "
                              "slower than form 0, built only into libavsep_hip_dev.so");
#endif
    if (!scratch) return fail(AVSEP_EINVAL, "form 2 needs 2*M floats of scratch");
    if (!gemm_ln_staged_supported(K)) return fail(AVSEP_EINVAL, "staged LayerNorm form: K is not supported");
    HCK(launch_layernorm_stats(x, scratch, M, K, eps, s));
    p.ln_gamma = gamma; p.ln_beta = beta; p.ln_eps = eps; p.ln_stats = scratch;
  } else if (form == 3) {                // LayerNorm in the epilogue: GEMM on the raw rows against W o gamma
    if (!scratch) return fail(AVSEP_EINVAL, "form 3 needs N*K + 2*N floats of scratch");
    if (N % 4) return fail(AVSEP_EINVAL, "form 3: N must be a multiple of 4");
    float* wp = scratch;
    float* c1 = scratch + (size_t)N * K;
    HCK(launch_pack_lnx(w, bias, gamma, beta, wp, c1, c1 + N, N, K, s));
    p.W = wp; p.bias = nullptr; p.lnx_c1 = c1; p.lnx_c2 = c1 + N; p.ln_eps = eps;
  } else {
    return fail(AVSEP_EINVAL, "form must be 0, 1, 2 or 3");
  }
  HCK(launch_gemm(p, s));
  return AVSEP_OK;
}

int avsep_op_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                       int B, int nhead, int dh, int Lq, int Lk, void* stream) {
  if (!q || !k || !v || !out) return fail(AVSEP_EINVAL, "null pointer");
  HCK(launch_attention(q, ldq, k, ldk, v, ldv, out, ldo, B, nhead, dh, Lq, Lk, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_attention_split(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo,
                             int B, int nhead, int dh, int Lq, int Lk, void* stream) {
  if (!q || !k || !v || !out) return fail(AVSEP_EINVAL, "null pointer");
  if (B <= 0 || nhead <= 0 || Lq <= 0 || Lk <= 0) return fail(AVSEP_EINVAL, "bad argument");
  if (!attention_split_supported(dh, Lq, Lk)) return fail(AVSEP_EINVAL, "the split-precision attention needs dh = 64");
  if ((ldq | ldk | ldv | ldo) & 3) return fail(AVSEP_EINVAL, "rows must be 16-byte aligned");
  HCK(launch_attention_split(q, ldq, k, ldk, v, ldv, out, ldo, B, nhead, dh, Lq, Lk, 1.0f, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_attention_h2(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, int B,
                          int nhead, int dh, int Lq, int Lk, int eq, int ek, int ev, void* stream) {
  if (!q || !k || !v || !out) return fail(AVSEP_EINVAL, "null pointer");
  if (B <= 0 || nhead <= 0 || Lq <= 0 || Lk <= 0) return fail(AVSEP_EINVAL, "bad argument");
  if (!attention_h2_supported(dh, Lq, Lk, eq, ek, ev)) return fail(AVSEP_EINVAL, "the two-term attention needs dh = 64 and |exponents| <= 60, |eq + ek| <= 60");
  if ((ldq | ldk | ldv | ldo) & 3) return fail(AVSEP_EINVAL, "rows must be 16-byte aligned");
  HCK(launch_attention_h2(q, ldq, k, ldk, v, ldv, out, ldo, B, nhead, dh, Lq, Lk, eq, ek, ev, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

// ---------------------------------------------------------------------------------- STFT front-end
int64_t avsep_stft_basis_floats(int n_fft) {
  if (n_fft <= 0 || (n_fft & 31)) return fail(AVSEP_EINVAL, "n_fft must be a positive multiple of 32");
  return (int64_t)2 * (n_fft / 2 + 1) * n_fft;
}

int avsep_stft_basis(float* basis, int n_fft, void* stream) {
  if (!basis) return fail(AVSEP_EINVAL, "null pointer");
  if (n_fft <= 0 || (n_fft & 31)) return fail(AVSEP_EINVAL, "n_fft must be a positive multiple of 32");
  HCK(launch_stft_basis(basis, n_fft, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

// spec[b][f][t] = | sum_k hann[k] audio[b][t*hop + k] e^{-2 pi i f k / n_fft} |, samples beyond L read as zero,
// T = 1 + L / hop frames (dataset.py:63-65, 122-135) -- one GEMM over overlapping rows of the waveform against the
// windowed DFT basis, magnitude taken in the epilogue, output in the reference's (B, F, T) layout.
int avsep_op_stft_mag(const float* audio, const float* basis, float* spec, int B, int L, int n_fft, int hop,
                      void* stream) {
  if (!audio || !basis || !spec) return fail(AVSEP_EINVAL, "null pointer");
  if (B <= 0 || L <= 0 || hop <= 0) return fail(AVSEP_EINVAL, "B, L and hop must be positive");
  if (n_fft <= 0 || (n_fft & 31)) return fail(AVSEP_EINVAL, "n_fft must be a positive multiple of 32");
  if ((L & 3) || (hop & 3)) return fail(AVSEP_EINVAL, "clip length and hop must be multiples of 4 samples (16-byte loads)");
  const int T = 1 + L / hop, F = n_fft / 2 + 1;
  if ((long long)B * T > 0x7fffffffLL) return fail(AVSEP_EINVAL, "too many frames");
  GemmParams p{};
  p.A = audio; p.W = basis; p.C = spec;
  p.M = B * T; p.N = 2 * F; p.K = n_fft;
  p.lda = hop; p.ldw = n_fft; p.ldc = 2 * F;
  p.amode = AMODE_FRAMES; p.T = T; p.frame_hop = hop; p.frame_len = L; p.mag_F = F;
  p.act = ACT_NONE;
  HCK(launch_gemm(p, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

#ifdef AVSEP_DEV
int avsep_op_mask_head(const float* x, const float* w, const float* bias, const float* xt, float* masks, float* sep, int M,
                       int N, int K, int F, int ldx, int act, int general, void* stream) {
  if (!x || !w || !xt || !masks || !sep || M <= 0 || N <= 0 || K <= 0 || F <= 0 || ldx < F) return fail(AVSEP_EINVAL, "bad argument");
  GemmParams p = linear_params(x, K, w, K, bias, masks, N, M, N, act);
  p.C2 = sep; p.X = xt; p.ldx = ldx; p.F = F;
  p.epi_general = general ? 1 : 0;
  HCK(launch_gemm(p, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

int avsep_op_attention_proj(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* wo,
                            const float* bo, float* x, int B, int nhead, int dh, int Lq, int Lk, void* stream) {
  if (!q || !k || !v || !wo || !x) return fail(AVSEP_EINVAL, "null pointer");
  const hipError_t e = launch_attn_proj(q, ldq, k, ldk, v, ldv, wo, bo, x, B, nhead, dh, Lq, Lk, reinterpret_cast<hipStream_t>(stream));
  if (e == hipErrorNotSupported)
    return fail(AVSEP_EINVAL, "fused attention + projection needs dh = 64, 49..64 keys and at most 8 heads");
  HCK(e);
  return AVSEP_OK;
}
#endif  // AVSEP_DEV

int avsep_op_interp_linear(const float* x, float* y, int B, int N, int T, int d, void* stream) {
  if (!x || !y || B <= 0 || N <= 0 || T <= 0 || d <= 0) return fail(AVSEP_EINVAL, "bad argument");
  HCK(launch_interp_linear(x, y, B, N, T, d, reinterpret_cast<hipStream_t>(stream)));
  return AVSEP_OK;
}

}  // extern "C"
