/*
 * avsep.h -- C ABI of the MI355X-native AV-Separation-Transformer forward path (libavsep_hip.so).
 *
 * This is the drop-in boundary of DESIGN.md §(b): plain pointers and sizes, no torch types.  The
 * reference has NO native/FFI layer (8 pure-Python files); its "interface" for this
 * path is the set of torch.nn module calls in /root/reference/src/av_separation/model.py.  Each entry
 * point below names the reference call it replaces.  The Python host mirror
 * (av-separation-transformer_amd/av_separation/) binds these with ctypes; INTEGRATION.md shows the stub
 * a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to float32 unless noted; tensors are dense row-major
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises the host
 *   - functions return 0 on success, a negative AVSEP_E* code otherwise, and never throw;
 *     avsep_last_error() gives a thread-local message for the last failure
 *   - inputs are borrowed and never written; outputs/workspace are caller-owned
 */
#ifndef AVSEP_H_
#define AVSEP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVSEP_ABI_VERSION 1

enum {
  AVSEP_OK = 0,
  AVSEP_EINVAL = -1,     /* bad argument / unsupported shape            */
  AVSEP_ENOWEIGHT = -2,  /* a weight needed by the called stage is unset */
  AVSEP_EHIP = -3,       /* a HIP runtime call failed                   */
  AVSEP_ENOMEM = -4,     /* workspace too small / allocation failed     */
  AVSEP_ESTATE = -5,     /* weights not finalized                       */
  AVSEP_EINTERNAL = -6   /* unexpected internal error (a C++ exception was caught at the boundary) */
};

/* Constructor arguments of AVSeparationTransformer (model.py:240-249).  dropout is not part of the
 * inference path (identity in eval, model.py:301). */
typedef struct avsep_config {
  int32_t freq_bins;           /* F   */
  int32_t d_model;             /* d   (multiple of 32) */
  int32_t nhead;               /* h   (d/h a multiple of 4, <= 128) */
  int32_t num_encoder_layers;  /* Le  */
  int32_t num_fusion_layers;   /* Lf  */
  int32_t num_speakers;        /* S   */
} avsep_config;

typedef struct avsep_ctx avsep_ctx;

int avsep_abi_version(void);
const char* avsep_last_error(void);
/* 16 hex digits: hash of the kernel sources this library was built from (csrc/Makefile).  Measurement hygiene only:
 * profiles/pmc_hbm_traffic.json is stamped with it and bench.py reports PMC traffic only for a matching library. */
const char* avsep_build_id(void);

/* One context per device & model.  Allocates (hipMalloc) the packed-weight arena. */
int avsep_create(const avsep_config* cfg, avsep_ctx** out);
/* avsep_create with creation-time options (round 5): AVSEP_CREATE_FP32_MATRIX = the d_model >= 512 forward on the fp32
 * MFMA kernels from the start (what avsep_set_split_precision(ctx, 0) switches to later) -- the choice for a deployment that
 * calls the model with ONE or two clips at a time, where those kernels are still ~20 % faster (config 3, one clip: 0.84 ms against
 * 1.05 ms per forward); from 2-4 clips on the default split-precision kernels win (64 clips: 1.9x).  Same results within the
 * reference tolerance either way, and under either setting the same bits at every batch size. */
#define AVSEP_CREATE_FP32_MATRIX 1u
int avsep_create_ex(const avsep_config* cfg, uint32_t flags, avsep_ctx** out);
void avsep_destroy(avsep_ctx* ctx);

/* The weight ABI is the reference's state_dict: `key` is a reference state_dict key
 * ("audio_encoder.input_proj.0.weight", ...), `dev_ptr` a float32 device tensor of `shape`.
 * The pointer is only recorded; avsep_finalize_weights() reads all recorded tensors on `stream` and
 * re-packs them (BN folding model.py:83-89, q-scale folding, conv tap-major layouts, K padding) into
 * the context's arena, after which the caller's tensors are no longer referenced.
 * `*.pe` buffers (model.py:297) are accepted; `*.num_batches_tracked` is ignored. */
int avsep_set_weight(avsep_ctx* ctx, const char* key, const float* dev_ptr, const int64_t* shape, int ndim);
int avsep_finalize_weights(avsep_ctx* ctx, void* stream);

/* Bytes of scratch avsep_forward()/stage calls need for these sizes (pure host arithmetic). */
size_t avsep_workspace_bytes(const avsep_ctx* ctx, int B, int T, int N, int H, int W);

/* AVSeparationTransformer.forward (model.py:268-276).
 *   mixed   (B,F,T)      lips (B,N,H,W)
 *   masks_btsf, separated_btsf: (B,T,S,F) row-major -- the memory order of the reference's outputs,
 *   whose logical (B,S,F,T) tensors are views with strides (S*F*T, F, 1, S*F).
 * The audio and visual encoders run concurrently on `stream` and one context-owned side stream. */
int avsep_forward(avsep_ctx* ctx, const float* mixed, const float* lips, float* masks_btsf,
                  float* separated_btsf, void* workspace, size_t workspace_bytes, int B, int T, int N,
                  int H, int W, void* stream);

/* Same call, replayed from a hipGraph captured on first use for this exact argument tuple
 * (launch-bound at small batch: DESIGN.md "graphs").  Pointers must stay valid between calls. */
int avsep_forward_graph(avsep_ctx* ctx, const float* mixed, const float* lips, float* masks_btsf,
                        float* separated_btsf, void* workspace, size_t workspace_bytes, int B, int T,
                        int N, int H, int W, void* stream);

/* Stage entry points = the reference's sub-modules, usable stand-alone like in its tests.
 *   AudioEncoder.forward      model.py:54-60     mixed (B,F,T)            -> out (B,T,d)
 *   VisualEncoder.forward     model.py:103-117   lips (B,N,H,W), T        -> out (B,T,d)
 *   CrossModalFusion.forward  model.py:145-149   audio,visual (B,T,d)     -> out (B,T,d)
 *   SeparationDecoder.forward+separate  model.py:201-220  fused (B,T,d), mixed (B,F,T) (may be NULL,
 *                             then separated_btsf must be NULL too)        -> masks/separated (B,T,S,F) */
int avsep_audio_encoder(avsep_ctx* ctx, const float* mixed, float* out, void* ws, size_t ws_bytes, int B,
                        int T, void* stream);
int avsep_visual_encoder(avsep_ctx* ctx, const float* lips, float* out, void* ws, size_t ws_bytes, int B,
                         int N, int H, int W, int T, void* stream);
int avsep_fusion(avsep_ctx* ctx, const float* audio, const float* visual, float* out, void* ws,
                 size_t ws_bytes, int B, int T, void* stream);
int avsep_decoder(avsep_ctx* ctx, const float* fused, const float* mixed, float* masks_btsf,
                  float* separated_btsf, void* ws, size_t ws_bytes, int B, int T, void* stream);

/* Split-precision kernels (csrc/gemm_h2.hip, gemm_planes.hip, gemm_split.hip, attention_split.hip) of this context's forwards on
 * (1, the default) or off (0).  They apply to d_model >= 512 models only (weights with N, K >= 512; attention at 128 keys or more)
 * and are fp32-equivalent (see avsep_op_linear_h2 / avsep_op_linear_split): 1.9x the fp32 MFMA kernels' throughput at config 3's
 * 64 clips per forward, ahead from 2-4 clips on -- and still ~20 % SLOWER for ONE clip at a time (profiles/r05_ab_small_batches.txt:
 * 1.05 ms against 0.84 ms for one clip of config 3; round 4: 1.40 ms).  The choice never looks at the batch size by itself (a model
 * computes the same bits at every batch size under either setting): a latency deployment creates its context with
 * AVSEP_CREATE_FP32_MATRIX (avsep_create_ex) or turns them off here.  Captured graphs of the other setting are dropped. */
int avsep_set_split_precision(avsep_ctx* ctx, int enable);
/* Debug/parity taps.  After avsep_set_debug_taps(ctx, 1), avsep_workspace_bytes() reserves a tap area and
 * every eager forward/stage call copies its stage-boundary activations there; avsep_read_tap() copies one
 * of them (names follow oracle/numpy_forward.py: "a_conv1","a_pe","a_enc0","v_conv0".."v_conv2" (channels-
 * last), "v_pool","v_enc0","v_interp","f_layer0","f_norm") from the same workspace/sizes into `dst`.
 * Returns the number of floats written, or a negative error.  Not available under graph replay. */
int avsep_set_debug_taps(avsep_ctx* ctx, int on);
int64_t avsep_read_tap(avsep_ctx* ctx, const char* name, float* dst, int64_t max_floats, void* workspace, int B,
                       int T, int N, int H, int W, void* stream);

/* Live per-kernel profile: between avsep_profile_begin() and avsep_profile_end() every kernel the EAGER
 * entry points launch is issued 20x back to back between one pair of HIP events on the stream it runs on
 * (an event record costs microseconds here, a kernel may take less) and its MEAN duration is kept; outputs
 * of a profiled call are therefore meaningless (residual updates applied 20x).  avsep_profile_end() waits
 * for the events and writes a JSON array aggregated per kernel (template instance) in first-launch order:
 *   [{"name":"gemm_kernel<64, 32, 0>","calls":n,"ms":total,"flops":algorithmic,"bytes":algorithmic},...]
 * Returns the JSON length or a negative error.  bench.py prices its roofline from this. */
int avsep_profile_begin(avsep_ctx* ctx);
/* Timeline aid (developer tool tools/stamps.py): with AVSEP_STAMPS=1 in the environment at avsep_create(), every
 * forward also runs one-lane kernels that store the device's 100 MHz wall clock at stage boundaries (0 audio start,
 * 1 visual start, 2 visual encoder done, 3 K/V projection done, 4 audio done, 5 tail start, 6/8 second-half tail
 * start/end, 7 first-half tail end, 9 step end); this copies the last forward's first n (<= 16) stamps to the host
 * (synchronous).  Works under graph replay, where no profiler-free timeline exists otherwise. */
int avsep_read_stamps(avsep_ctx* ctx, uint64_t* out, int n);
int64_t avsep_profile_end(avsep_ctx* ctx, char* json, size_t capacity);

/* Single-kernel entry points (parity tests drive every kernel through the ABI).
 * y = act(LN?(x) W^T + b) (+ residual); act: 0 none, 1 relu, 2 gelu(erf), 3 sigmoid. */
int avsep_op_linear(const float* x, const float* w, const float* bias, const float* residual, float* y, int M,
                    int N, int K, int act, void* stream);
/* The same Linear on the split-precision GEMM (csrc/gemm_split.hip; what the forward runs for weights with N >= 512 and
 * K >= 512, i.e. every nn.Linear of the d_model >= 512 configurations, /root/reference/src/av_separation/model.py:48-52, 93,
 * 155-161, 195): each fp32 operand is cut into three bf16 terms (exact truncation splits), the six significant bf16 x bf16
 * products are accumulated in fp32 on the bf16 matrix cores.  fp32 in, fp32 out, error against float64 at the fp32 GEMM's own level
 * (tests/test_gpu_parity.py::test_op_linear_split_precision); NOT bit-identical to avsep_op_linear. */
int avsep_op_linear_split(const float* x, const float* w, const float* bias, const float* residual, float* y, int M, int N,
                          int K, int act, void* stream);
/* The same GEMM on PRE-SPLIT operands (csrc/gemm_planes.hip, round 5).  avsep_op_split_planes cuts x (M, K; row stride ld floats)
 * into its three bf16 terms ("planes": bf16 [K/32][3][rows][32], element (m, k) of term t at ((k/32 * 3 + t) * rows + m) * 32 + k % 32;
 * rows >= M is the buffer's row count, 6 * rows * K bytes) -- what the forward's producers (weight packer, LayerNorm, GEMM and
 * attention epilogues) write once per element.  avsep_op_linear_planes multiplies two such operands, staging them by LDS-DMA:
 * y = act(x w^T + bias) + residual as fp32 (y, may be null) and / or as the planes of the next GEMM's operand (yp / y_rows, may be
 * null; N % 32 == 0, no residual).  Same six products in the same order as avsep_op_linear_split: bit-identical to it on the
 * fp32 operands the planes were cut from.  A term of +-inf is NaN (inf - inf): non-finite inputs give non-finite outputs, but an
 * inf may come out as NaN.  nn.Linear, /root/reference/src/av_separation/model.py:48-52,155-161,194-199. */
int avsep_op_split_planes(const float* x, int ld, uint16_t* planes, int64_t rows, int M, int K, void* stream);
int avsep_op_linear_planes(const uint16_t* xp, int64_t x_rows, const uint16_t* wp, int64_t w_rows, const float* bias,
                           const float* residual, float* y, uint16_t* yp, int64_t y_rows, int M, int N, int K, int act,
                           void* stream);
/* The producers of such planes inside the forward, as ops (each writes exactly avsep_op_split_planes of what its fp32 twin writes,
 * bit for bit): nn.LayerNorm (avsep_op_layernorm; d % 32 == 0), F.interpolate(mode="linear") (avsep_op_interp_linear; d % 32 == 0) and
 * the split-precision attention (avsep_op_attention_split; dh = 64; output row b * Lq + q, column h * 64 + c of a (B * Lq, nhead * 64)
 * matrix).  /root/reference/src/av_separation/model.py:48-52,114-116,143,155,162-163. */
int avsep_op_layernorm_planes(const float* x, const float* gamma, const float* beta, uint16_t* yp, int64_t rows, int M, int d, float eps,
                              void* stream);
int avsep_op_interp_linear_planes(const float* x, uint16_t* yp, int64_t rows, int B, int N, int T, int d, void* stream);
int avsep_op_attention_split_planes(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, uint16_t* op, int64_t rows,
                                    int B, int nhead, int dh, int Lq, int Lk, void* stream);
/* TWO fp16 terms, THREE products per fp32 product (csrc/gemm_h2.hip, round 5): what the d_model >= 512 forward runs wherever a GEMM's
 * input has a static bound from the weights alone (LayerNorm outputs, act(LayerNorm(x) W^T + b), self-attention outputs) -- half the
 * matrix work of the three-term bf16 kernels at an error at or below the fp32 MFMA GEMM's own.  Operands are stored scaled by powers
 * of two (exact) that put them under 2^14 (fp16's range): H2 planes = fp16 [K/32][2][rows][32], element (m, k) of term t (0 = hi =
 * rn16(x 2^e), 1 = lo = rn16(x 2^e - hi)) at ((k/32 * 2 + t) * rows + m) * 32 + k % 32.
 *   avsep_op_h2_row_stats   per row n of w (N, K): ew[n] = the exponent that puts max|w[n][:]| into [2^13, 2^14) (0 for a zero row),
 *                           l2[n] >= ||w[n][:]||_2 (for the bounds avsep_finalize_weights derives the activations' exponents from)
 *   avsep_op_split_h2       x (M, K; row stride ld) -> H2 planes of x 2^e, e = row_exp[m], or the one exponent `e` with row_exp null.
 *                           The caller guarantees |x| 2^e <= 65504 (a value beyond becomes inf).
 *   avsep_op_linear_h2      y = act((sum_k x'_k w'_k) * cscale[n] (* rscale[m]) + bias[n]) + residual, cscale[n] = 2^-(ex + ew[n]); as fp32 (y) and / or
 *                           as the H2 planes of y 2^yp_exp (yp / y_rows; N % 32 == 0, no residual, not sigmoid).  N even (N % 4 != 0: the mask
 *                           head's epilogue needs avsep_op_mask_head's operands and is reached through the forward only).
 *                           A row computed alone has the bits it has inside any batch (64 x 64 and 256 x 128 kernels, same products, same order).
 *   avsep_op_layernorm_h2, avsep_op_attention_split_h2     the forward's producers of such planes: exactly avsep_op_split_h2(e) of what
 *                           avsep_op_layernorm / avsep_op_attention_split write.
 * NORMWISE accuracy: an element below 2^-40 of its tensor's bound is lost (the three-term bf16 kernels keep every element to 24 bits).
 * Non-finite inputs give non-finite outputs.  nn.Linear, nn.LayerNorm, nn.MultiheadAttention: /root/reference/src/av_separation/model.py:48-52,155-164,194-199. */
int avsep_op_h2_row_stats(const float* w, int N, int K, int32_t* ew, float* l2, void* stream);
int avsep_op_split_h2(const float* x, int ld, uint16_t* planes, int64_t rows, int M, int K, const int32_t* row_exp, int e, void* stream);
int avsep_op_linear_h2(const uint16_t* xp, int64_t x_rows, const uint16_t* wp, int64_t w_rows, const float* cscale, const float* rscale,
                       const float* bias, const float* residual, float* y, uint16_t* yp, int64_t y_rows, int yp_exp, int M, int N, int K,
                       int act, void* stream);
/* rscale (may be null): an operand WITHOUT a static bound carries one power of two per ROW, taken from the row itself by its producer --
 * avsep_op_interp_linear_h2 = avsep_op_interp_linear as two-term planes of each row scaled into [2^13, 2^14), rscale[row] = the inverse
 * (what the forward runs between the visual encoder and the fusion K/V projection) --, and the GEMM multiplies row m of its
 * accumulators by rscale[m] (exact) beside cscale[n] = 2^-ew[n].  fp32 output only. */
int avsep_op_interp_linear_h2(const float* x, uint16_t* yp, float* rscale, int64_t rows, int B, int N, int T, int d, void* stream);
int avsep_op_layernorm_h2(const float* x, const float* gamma, const float* beta, uint16_t* yp, int64_t rows, int M, int d, float eps, int e,
                          void* stream);
int avsep_op_attention_split_h2(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, uint16_t* op, int64_t rows,
                                int e, int B, int nhead, int dh, int Lq, int Lk, void* stream);
int avsep_op_layernorm(const float* x, const float* gamma, const float* beta, float* y, int M, int d,
                       float eps, void* stream);
/* y = act(LayerNorm(x) W^T + b), the pair every pre-norm block of the model is made of (model.py:48-52 norm_first
 * layers, 145-149, 201): x (M,K), w (N,K).  `form` picks how the engine may run it -- all four are the same
 * function: 0 = LayerNorm launch into scratch (M*K floats) + GEMM (what large batches and d_model > 256 use);
 * 1 = statistics and normalisation inside the GEMM, in front of its first MFMA (K <= 256; rounds 1-2);
 * 2 = a statistics launch into scratch (2*M floats: mean, 1/std per row) and a GEMM that normalises its A tile on the
 * way to LDS (K <= 512; bit-identical to form 0; developer library only, measured slower);
 * 3 = LayerNorm in the EPILOGUE (what the model's d_model <= 256 sites use since round 3): the GEMM runs on the raw rows
 * against W o gamma, the row statistics are summed on the side, y = act(rstd (acc - mean c1) + c2) with c1 = rowsum(W o gamma),
 * c2 = W beta + b, which the call packs into scratch (N*K + 2*N floats; avsep_finalize_weights does it once per weight
 * update for the model); N % 4 == 0.  Same function as form 0, not the same bits. */
int avsep_op_ln_linear(const float* x, const float* gamma, const float* beta, const float* w, const float* bias,
                       float* y, float* scratch, int M, int N, int K, int act, float eps, int form, void* stream);
/* softmax(q k^T) v per (batch, head); q is expected pre-scaled.  q (B,Lq,ldq) etc. with head h at
 * column offset h*dh; out (B,Lq,ldo). */
int avsep_op_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out,
                       int ldo, int B, int nhead, int dh, int Lq, int Lk, void* stream);
/* The same function with both matrix products (q k^T and p v) on the bf16 matrix pipe: every fp32 operand -- q, k, v and the
 * probabilities -- cut into three bf16 terms, six bf16 MFMA products per fp32 product, fp32 accumulation and softmax
 * (csrc/attention_split.hip); dh = 64.  Not the bits of avsep_op_attention: the same distance from float64
 * (tests/test_gpu_parity.py::test_op_attention_split_precision).  What the forward runs for d_model >= 512 models at 128
 * keys or more (nn.MultiheadAttention, /root/reference/src/av_separation/model.py:46-60,159-172). */
int avsep_op_attention_split(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out,
                             int ldo, int B, int nhead, int dh, int Lq, int Lk, void* stream);
/* The same function on TWO fp16 terms per operand and three fp16 MFMA products per fp32 product (csrc/attention_split.hip,
 * attention_h2_kernel; the scheme of avsep_op_linear_h2): the caller states a bound on its operands as powers of two -- every |q| 2^eq,
 * |k| 2^ek, |v| 2^ev it can produce is below 2^14 (larger values saturate to +-inf: a bound, not a measurement) -- the probabilities
 * are in [0, 1] by construction.  22 significant bits for every operand entry within 17 binades of its bound, an absolute error of
 * 2^-39 of the bound below that.  The forward runs it for the attention of d_model >= 512 models at 128 keys or more: self-attention
 * with the bounds avsep_finalize_weights derives from the in-projection behind its LayerNorm; cross-attention with the q bound of its
 * q-projection and k / v bounds PER CLIP, computed on the device in every forward from the row scales of the resized visual stream
 * (its magnitude is data, not weights) -- a clip's exponents depend on that clip alone, so a clip has the same bits alone and inside
 * any batch (nn.MultiheadAttention, /root/reference/src/av_separation/model.py:46-60,159-172). */
int avsep_op_attention_h2(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out, int ldo, int B,
                          int nhead, int dh, int Lq, int Lk, int eq, int ek, int ev, void* stream);
#ifdef AVSEP_DEV
/* Developer build only (measured slower than the two launches it replaces).
 * x += softmax(q k^T) v  W_o^T + b_o in ONE launch: the attention core of a pre-norm block with its out_proj and the
 * residual add (nn.MultiheadAttention inside nn.TransformerEncoderLayer, model.py:48-52, and CrossAttentionLayer 168-170)
 * for short sequences -- dh = 64, 49 <= Lk <= 64, nhead <= 8, d = 64 nhead (the 1 s clips of BASELINE configs 1 / 2).
 * q (B*Lq, ldq), k / v (B*Lk, ldk / ldv) with head h at column 64 h, q pre-scaled; wo (d, d), bo (d) or null; x (B*Lq, d)
 * updated in place.  Bit-identical to avsep_op_attention followed by avsep_op_linear(residual = x). */
int avsep_op_attention_proj(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* wo,
                            const float* bo, float* x, int B, int nhead, int dh, int Lq, int Lk, void* stream);
/* The mask head alone (decoder.decoder.3 + Sigmoid + separate(), model.py:195-220): masks = act(x w^T + bias) (M, N),
 * sep = masks * xt[m][n % F] with xt (M, ldx); general != 0 forces the block-by-block epilogue the straight-line one is
 * tested against bit for bit. */
int avsep_op_mask_head(const float* x, const float* w, const float* bias, const float* xt, float* masks, float* sep, int M,
                       int N, int K, int F, int ldx, int act, int general, void* stream);
/* DEVELOPER BUILD ONLY (measured slower than the launch-per-op schedule: DESIGN.md (d), profiles/r04_chain_*).  Launch schedule of
 * the eval forward (round 4).  0 (default, the only one of the product library): one launch per op, two streams.  1: the pre-norm encoder layers of each branch (nn.TransformerEncoder of AudioEncoder / VisualEncoder,
 * /root/reference/src/av_separation/model.py:48-52,59 and 97-101,111) run as ONE dependency-driven persistent launch per
 * branch (csrc/chain.hip): tiles of the same kernels' code, started as soon as the producer tiles of THEIR rows / clip have
 * finished instead of after a device-wide launch boundary; outputs are bit-identical to schedule 0.  Schedule 1 keeps ONE work
 * queue and hands tiles over with write-through stores (any workgroup may run any tile); schedule 2 keeps one queue per XCD
 * (clip groups; a workgroup reads the XCD it runs on from the hardware) and hands over through that XCD's L2 with plain stores.
 * `group` / `skew` order a queue: sub-groups of `group` clips run `skew` ops apart (skew 0 = op-major).  Applies to models whose
 * layers take the LayerNorm-in-the-epilogue GEMM and the short-sequence attention (d_model <= 256, head dim 64, 49..64
 * positions); other shapes keep schedule 0 silently.  Drops the context's captured graphs.  Not an environment switch: the
 * product library reads none. */
int avsep_set_schedule(avsep_ctx* ctx, int schedule, int group, float skew);
/* After a forward under schedule 1: waits for `stream` and returns AVSEP_OK when every dependency wait of the chained launches
 * was satisfied, AVSEP_EINTERNAL (with avsep_last_error()) when a bounded spin gave up (outputs are then invalid). */
int avsep_chain_status(avsep_ctx* ctx, void* stream);
/* developer aid: copies the first n state words (ticket head, error word, two unused, arrival counters ...) of chained plan
 * `idx` on a stream of its own, also while the launch is running; returns the number of plans */
int avsep_chain_peek(avsep_ctx* ctx, int idx, unsigned* out, int n);

#endif  /* AVSEP_DEV */
int avsep_op_interp_linear(const float* x, float* y, int B, int N, int T, int d, void* stream);
#ifdef AVSEP_DEV
/* Developer build only (libavsep_hip_dev.so): the paired-launch experiment of round 3, measured slower than the two-stream
 * schedule and therefore not part of the product ABI. */
/* The two instances of one encoder-layer kernel -- AudioEncoder's and VisualEncoder's layer i, /root/reference/src/
 * av_separation/model.py:48-52 and 97-101: same weight shapes, M0 = B*T and M1 = B*N rows -- as ONE launch (what the
 * forward path does for every encoder layer).  Problem j: y_j = act(LN?(x_j) w_j^T + b_j) (+ r_j); gamma_j / beta_j
 * null = no LayerNorm (both or neither; K <= 256 when given), b_j and r_j both given or both null across the pair.
 * Bit-identical to two avsep_op_linear / avsep_op_ln_linear(form 1) calls. */
int avsep_op_linear_pair(const float* x0, const float* w0, const float* b0, const float* r0, const float* gamma0,
                         const float* beta0, float* y0, int M0, const float* x1, const float* w1, const float* b1,
                         const float* r1, const float* gamma1, const float* beta1, float* y1, int M1, int N, int K,
                         int act, float eps, void* stream);
/* Two attention problems with the same head count and head size (the audio and the visual self-attention of one layer)
 * in one launch when both run the short-sequence kernel (dh = 64, 49..64 keys), else two launches on `stream`. */
int avsep_op_attention_pair(const float* q0, const float* k0, const float* v0, float* o0, int ldqkv0, int ldo0, int B0,
                            int L0, const float* q1, const float* k1, const float* v1, float* o1, int ldqkv1, int ldo1,
                            int B1, int L1, int nhead, int dh, void* stream);
#endif  /* AVSEP_DEV */

/* ------------------------------------------------------------------------------------------------------------
 * STFT magnitude front-end: replaces SyntheticAVDataset._stft
 * (/root/reference/src/av_separation/dataset.py:122-135) -- Hann(n_fft) window (np.hanning, symmetric), hop `hop`,
 * T = 1 + L / hop frames (dataset.py:63-65), tail frames zero-padded, |rfft| -- for a batch of waveforms resident in
 * HBM.  One fp32-MFMA GEMM of overlapping waveform rows against a windowed real-DFT basis, magnitude in the epilogue.
 *   avsep_stft_basis_floats(n_fft)      floats the caller allocates for the basis (2*(n_fft/2+1) * n_fft)
 *   avsep_stft_basis(basis, n_fft, s)   fills it (once per n_fft)
 *   avsep_op_stft_mag(audio (B,L), basis, spec (B, n_fft/2+1, T), B, L, n_fft, hop, s)
 * n_fft % 32 == 0, L % 4 == 0, hop % 4 == 0 (every configuration the reference uses); otherwise AVSEP_EINVAL.
 * ------------------------------------------------------------------------------------------------------------ */
int64_t avsep_stft_basis_floats(int n_fft);
int avsep_stft_basis(float* basis, int n_fft, void* stream);
int avsep_op_stft_mag(const float* audio, const float* basis, float* spec, int B, int L, int n_fft, int hop,
                      void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Training ops (what demo.py:83-113 / tests/test_model.py:210-217,332-353 exercise through
 * torch autograd).  Stateless; activations are row tensors [rows][C] (channels last); scratch is caller-owned.
 * The Python layer (av_separation/_train.py) composes them under torch.autograd.Function wrappers.  Dense
 * contractions of the backward pass reuse avsep_op_linear_ex on transposed operands (dX = dY W, dW = dY^T X).
 * ------------------------------------------------------------------------------------------------------------ */
/* y[m][n] = act(x W^T + b) (+ residual[(m % rperiod)][n], rperiod <= 0: row m); general leading dimensions */
int avsep_op_linear_ex(const float* x, int lda, const float* w, int ldw, const float* bias, const float* residual,
                       int ldr, int rperiod, float* y, int ldc, int M, int N, int K, int act, void* stream);
/* attention forward that also returns the per-query log-sum-exp (B*nhead*Lq) and scales q by qscale on load;
 * drop_p > 0: dropout on the attention probabilities (nn.MultiheadAttention(dropout=p) in training), stateless
 * mask from (drop_seed, element index) that the backward regenerates */
/* Weight gradient of a linear/conv layer: dw[N][K] = dyt[N][R] . xt[K][R]^T, the row index R (a multiple of 32,
 * zero padded) being the contraction -- both operands are the transposes avsep_op_transpose produces.  When N*K is
 * small and R long (conv layers) the contraction is split over workgroups and summed in a fixed order; `scratch`
 * must then hold avsep_op_wgrad_scratch_floats(N, K, R) floats (0 = not needed).  Replaces the autograd of
 * nn.Linear / nn.Conv1d / nn.Conv2d weights (model.py:38-40, 82-93, ...). */
int64_t avsep_op_wgrad_scratch_floats(int N, int K, int R);
int avsep_op_wgrad(const float* dyt, const float* xt, float* dw, float* scratch, int N, int K, int R, void* stream);
/* The same weight gradient straight from the row-major tensors autograd holds: dy [R][ldy] (N columns used),
 * x [R][ldx] (K columns used) -- no transposed copies; rows are the contraction and may be any count.  Needs N, K,
 * ldy, ldx to be multiples of 4 (float4 columns); otherwise use avsep_op_wgrad on padded transposes. */
int64_t avsep_op_wgrad_direct_scratch_floats(int N, int K, int R);
int avsep_op_wgrad_direct(const float* dy, int ldy, const float* x, int ldx, float* dw, float* scratch, int N, int K,
                          int R, void* stream);
/* The same launch also producing the bias gradient db[n] = sum_r dy[r][n] (the workgroups of the first K tile add up the
 * dy chunks they stage anyway -- two launches per layer less than a separate column reduction): dwb holds N*K weight
 * gradients followed by N bias gradients. */
int64_t avsep_op_wgrad_bias_direct_scratch_floats(int N, int K, int R);
int avsep_op_wgrad_bias_direct(const float* dy, int ldy, const float* x, int ldx, float* dwb, float* scratch, int N,
                               int K, int R, void* stream);
#ifdef AVSEP_DEV
/* Developer build only: measured 12 % slower on the training step than the two-launch form (every one of a launch's ~1000
 * workgroups pays an agent-scope release of 16 KB of freshly written partials, 256 KB of slices per tile for the reducer).
 * The same gradients (with_bias = 0: dwb holds only the N*K weight gradients) with the split contraction's slices summed
 * INSIDE the launch: the workgroup that arrives last at a tile's ticket counter adds the slices in slice order -- the values
 * of avsep_op_wgrad_(bias_)direct bit for bit, one launch instead of two.  scratch: the floats
 * avsep_op_wgrad_bias_direct_scratch_floats() asks for (0 = no split, counters unused); counters: avsep_op_wgrad_tiles()
 * 32-bit words, ZERO before the first call and left at zero by every call; one launch at a time may use a counter buffer. */
int64_t avsep_op_wgrad_tiles(int N, int K, int R);
int avsep_op_wgrad_merged(const float* dy, int ldy, const float* x, int ldx, float* dwb, float* scratch, uint32_t* counters,
                          int N, int K, int R, int with_bias, void* stream);
#endif  /* AVSEP_DEV */

int avsep_op_attention_train(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* out,
                             int ldo, float* lse, int B, int nhead, int dh, int Lq, int Lk, float qscale, float drop_p,
                             uint64_t drop_seed, void* stream);
/* inverted dropout y = keep ? x/(1-p) : 0 with the same stateless mask; its backward is the same call on dy */
int avsep_op_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream);
/* y = residual + dropout(x) in one launch (dropout1 / dropout2 + residual add of a transformer block) */
int avsep_op_dropout_add(const float* x, const float* residual, float* y, int64_t n, float p, uint64_t seed,
                         void* stream);
/* gradients of softmax((qscale q) k^T) v w.r.t. q, k, v; dvec: B*nhead*Lq floats of scratch; dh % 16 == 0 */
int avsep_op_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                           int ldo, const float* d_out, int lddo, const float* lse, float* dvec, float* dq, int lddq,
                           float* dk, int lddk, float* dv, int lddv, int B, int nhead, int dh, int Lq, int Lk,
                           float qscale, float drop_p, uint64_t drop_seed, void* stream);
/* y[c][r] = x[r][c] (x [R][C] -> y [C][Rp], zero for r >= R) */
int avsep_op_transpose(const float* x, float* y, int R, int C, int Rp, void* stream);
/* The same for n matrices in one launch: `table` is a DEVICE array of n descriptors {src, dst, R, C, Rp, pad} (32 bytes
 * each, below); max_rp / max_c = the largest Rp / C in the table (grid size).  The training path makes the W^T operands
 * of all Linear layers' activation-gradient GEMMs with one call per step. */
typedef struct avsep_transpose_desc { const float* src; float* dst; int R, C, Rp, pad; } avsep_transpose_desc;
int avsep_op_transpose_many(const void* table, int n, int max_rp, int max_c, void* stream);
/* (B,F,T) -> (B,T,Fp), zero padded: the layout pass in front of the first Conv1d */
int avsep_op_transpose_pad(const float* x, float* y, int B, int F, int T, int Fp, void* stream);
/* Conv1d k3 p1 / Conv2d k3 s2 p1 as im2col (+ GEMM) and their adjoints */
int avsep_op_im2col1d(const float* x, float* col, int M, int T, int C, void* stream);
int avsep_op_col2im1d(const float* dcol, float* dx, int M, int T, int C, void* stream);
int avsep_op_im2col2d(const float* x, float* col, int I, int H, int W, int C, int Kp, void* stream);
int avsep_op_col2im2d(const float* dcol, float* dx, int I, int H, int W, int C, int Kp, void* stream);
/* deterministic two-stage column sums: out0[c] = sum_r a[r][c]; with b: out1[c] = sum_r a[r][c] b[r][c] */
int64_t avsep_op_colreduce_scratch_floats(int M, int C);
int avsep_op_colreduce(const float* a, const float* b, float* scratch, float* out0, float* out1, int M, int C,
                       void* stream);
/* BatchNorm2d in training mode on rows [M][C] (+ optional fused ReLU): batch mean / biased var out, xhat saved,
 * running statistics updated in place (unbiased variance, momentum) when given */
int avsep_op_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* mean, float* var, float* xhat,
                          float* y, float* running_mean, float* running_var, float* scratch, int M, int C, float eps,
                          float momentum, int relu, void* stream);
int avsep_op_bn_train_bwd(const float* dy, const float* y, const float* xhat, const float* gamma, const float* var,
                          float* dx, float* dgamma, float* dbeta, float* dyr_scratch, float* scratch, int M, int C,
                          float eps, int relu, void* stream);

/* Split BatchNorm for data-parallel training ("BN statistics all-reduce ... to keep parity with the
 * reference's full-batch BatchNorm", model.py:83-89 in train mode).  The host combines the C-length vectors across
 * ranks between the halves: stats -> [all-gather mean/var/count, combine] -> apply;  bwd_sums -> [all-reduce] -> bwd_dx
 * with inv_count = 1 / (global row count).  var is the biased variance of this rank's M rows. */
int avsep_op_bn_stats(const float* x, float* mean, float* var, float* scratch, int M, int C, void* stream);
int avsep_op_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                      float* xhat, float* y, int M, int C, float eps, int relu, void* stream);
int avsep_op_bn_bwd_sums(const float* dy, const float* y, const float* xhat, float* dyr, float* sum_dy, float* sum_dyx,
                         float* scratch, int M, int C, int relu, void* stream);
int avsep_op_bn_bwd_dx(const float* dyr, const float* xhat, const float* gamma, const float* var, const float* sum_dy,
                       const float* sum_dyx, float* dx, int M, int C, float inv_count, float eps, void* stream);
/* activations (1 relu, 2 gelu-erf, 3 sigmoid); backward aux = output (relu, sigmoid) or pre-activation (gelu) */
int avsep_op_act_fwd(const float* x, float* y, int64_t n, int act, void* stream);
int avsep_op_act_bwd(const float* dy, const float* aux, float* dx, int64_t n, int act, void* stream);
/* out[m][s*F+f] = a[m][s*F+f] * xt[m][f]: SeparationDecoder.separate and its adjoint w.r.t. the masks */
int avsep_op_mul_mixed(const float* a, const float* xt, float* out, int64_t M, int S, int F, int ldx, void* stream);
/* y[m][c] = x[m][c] + r[m % period][c]  (PositionalEncoding add, model.py:300) */
int avsep_op_add_rows(const float* x, const float* r, float* y, int64_t M, int C, int period, void* stream);
int avsep_op_avgpool_fwd(const float* x, float* y, int M, int P, int C, void* stream);
int avsep_op_avgpool_bwd(const float* dy, float* dx, int M, int P, int C, void* stream);
int avsep_op_interp_linear_bwd(const float* dy, float* dx, int B, int N, int T, int d, void* stream);
/* xhat_scratch: unused since round 3 (kept in the signature for ABI stability); pass NULL */
int avsep_op_layernorm_bwd(const float* dy, const float* x, const float* gamma, float* dx, float* dgamma, float* dbeta,
                           float* xhat_scratch, float* scratch, int M, int d, float eps, void* stream);
/* The same with the gradient that reaches x along the RESIDUAL path of a pre-norm block (nn.TransformerEncoderLayer
 * norm_first=True, CrossAttentionLayer, model.py:48-52,168-172) added inside the kernel: dx = LayerNorm'(dy) + dres
 * (dres may be null) -- instead of the elementwise add autograd inserts for a tensor with two consumers. */
int avsep_op_layernorm_bwd_res(const float* dy, const float* x, const float* gamma, const float* dres, float* dx,
                               float* dgamma, float* dbeta, float* xhat_scratch, float* scratch, int M, int d, float eps,
                               void* stream);
/* y = residual + dropout(act(x w^T + b)) with the dropout of a train-mode transformer block (dropout1 / dropout2 / the FFN's
 * inner dropout, model.py:48-52,159,164) applied in the GEMM epilogue: inverted dropout, keep-mask = stateless hash of
 * (seed, element index m*N + n) -- the values of avsep_op_linear_ex followed by avsep_op_dropout / _dropout_add, bit for
 * bit, in one launch.  y is (M, N) contiguous. */
int avsep_op_linear_drop(const float* x, int lda, const float* w, int ldw, const float* bias, const float* residual,
                         int ldr, int rperiod, float* y, int M, int N, int K, int act, float drop_p, uint64_t drop_seed,
                         void* stream);
/* avsep_op_linear_ex / avsep_op_linear_drop (drop_p = 0: no dropout) on the split-precision GEMM of avsep_op_linear_split:
 * y = residual + dropout(act(x w^T + b)), same keep-mask as avsep_op_linear_drop, values equal to the fp32-MFMA ops' within
 * the fp32 GEMM's own rounding (tests/test_train_gpu.py).  The training step (av_separation/_train.py) runs every Linear
 * forward and activation-gradient GEMM whose weight has N >= 512 and K >= 512 through it -- the rule of the inference forward
 * (nn.Linear, /root/reference/src/av_separation/model.py:38-60).  N, ldc and ldr multiples of 4; dropout needs ldc == N. */
int avsep_op_linear_split_ex(const float* x, int lda, const float* w, int ldw, const float* bias, const float* residual,
                             int ldr, int rperiod, float* y, int ldc, int M, int N, int K, int act, float drop_p,
                             uint64_t drop_seed, void* stream);
/* avsep_op_wgrad_direct (with_bias = 0: dwb = N*K floats) / avsep_op_wgrad_bias_direct (with_bias = 1: dwb = N*K + N floats) on
 * the split-precision kernels (csrc/wgrad_split.hip): dW = dY^T X with both operands cut into three bf16 terms on their way to
 * LDS and transposed by the LDS read; the bias gradient is the fp32 op's, bit for bit.  Scratch: the fp32 op's
 * (avsep_op_wgrad_direct_scratch_floats / avsep_op_wgrad_bias_direct_scratch_floats).  Problems whose fp32 plan takes the
 * 32 x 32 tile run the fp32 kernel.  The training step uses it for weights with N, K >= 512 (av_separation/_train.py;
 * nn.Linear backward, /root/reference/src/av_separation/model.py:38-60). */
int avsep_op_wgrad_direct_split(const float* dy, int ldy, const float* x, int ldx, float* dwb, float* scratch, int N, int K, int R,
                                int with_bias, void* stream);
/* backward of y = dropout(relu(z)) from y alone: dx = y > 0 ? dy / (1 - p) : 0 */
int avsep_op_relu_dropout_bwd(const float* dy, const float* y, float* dx, int64_t n, float p, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AVSEP_H_ */
