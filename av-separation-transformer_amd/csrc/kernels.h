// kernels.h -- internal launcher declarations shared by the HIP translation units of libavsep_hip.so.
// Everything here is gfx950 (CDNA4) only: 64-wide wavefronts, v_mfma_f32_16x16x4_f32, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// Developer switches (tile overrides, A/B toggles, diagnostics) exist only in the developer build of the library
// (`make dev` -> libavsep_hip_dev.so, -DAVSEP_DEV): the product library reads no environment variable and carries none
// of the kernel instances that were measured slower and are kept for bit-identity tests and hardware sweeps.
#ifdef AVSEP_DEV
inline const char* dev_env(const char* name) { return getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Counter-based dropout mask (training path): a 32-bit hash of (seed, element index), kept when hash >= p * 2^32.
// Stateless, so the backward regenerates exactly the forward's mask.  Rounds 1-2 used splitmix64: three 64-bit multiplies =
// twelve quarter-rate 32-bit multiplies, ~240 VALU cycles per ELEMENT -- and VALU cycles are matrix-pipe cycles on this chip
// (profiles/r03_mfma_valu_exclusive.txt): 560 M mask elements per cfg4 training step were ~4 % of the step.  Now: the
// two-round xorshift-multiply integer hash "lowbias32" on (index ^ seed_lo), the seed's high word (and an index's, beyond
// 2^32 elements) folded in before a third multiply: three 32-bit multiplies, ~85 cycles.
__device__ __forceinline__ unsigned dropout_hash(unsigned long long seed, unsigned long long idx) {
  unsigned x = (unsigned)idx ^ (unsigned)seed;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  x ^= (unsigned)(seed >> 32) + (unsigned)(idx >> 32);
  x *= 0x9e3779b1u;
  x ^= x >> 15;
  return x;
}
// The seed a caller hands to the C ABI is mixed ONCE on the host (murmur3's 64-bit finaliser) before it reaches a kernel:
// dropout_hash() xors the raw low word into the index, so two user seeds with equal high words (consecutive seeds, small
// integers) would otherwise give masks that are xor-permutations of each other, with identical drop counts over aligned
// index ranges.  After the mix both words differ for any two seeds; no per-element cost.
inline unsigned long long dropout_mix_seed(unsigned long long s) {
  s ^= s >> 33; s *= 0xff51afd7ed558ccdULL;
  s ^= s >> 33; s *= 0xc4ceb9fe1a85ec53ULL;
  s ^= s >> 33;
  return s;
}
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned long long idx, float p) {
  return dropout_hash(seed, idx) >= (unsigned)(p * 4294967296.0f);     // 0 <= p < 1 (checked by the launchers)
}
// the kept elements' factor: ONE rounding of 1 / (1 - p), then a multiply per element at every site (a division per element
// is ~10 VALU instructions)
__device__ __forceinline__ float dropout_scale(float p) { return 1.0f / (1.0f - p); }

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SIGMOID = 3 };
enum { AMODE_PLAIN = 0, AMODE_TAPS3 = 1, AMODE_CONV2D = 2, AMODE_FRAMES = 3, AMODE_LN = 4, AMODE_LNX = 5 };
enum { LN_KMAX = 512 };   // widest LayerNorm the staged form (AMODE_LN) keeps gamma / beta in LDS for

// C[M,N] = epilogue( A'[M,K] * W[N,K]^T ), fp32 in / fp32 accumulate on the matrix cores.
// A' is A itself (PLAIN), a 3-tap shifted view of a (B,T,Kt) sequence tensor (TAPS3: Conv1d k=3 p=1 as one
// GEMM with K = 3*Kt, model.py:38,40), or an on-the-fly im2col gather of a channels-last image tensor
// (CONV2D: Conv2d k=3 s=2 p=1, K = 9*Cin, model.py:85,88).
struct GemmParams {
  const float* A;
  const float* W;      // [N][K] row-major, K contiguous (== nn.Linear.weight layout), K % 32 == 0
  const float* bias;   // [N] or nullptr
  float* C;            // [M][ldc]
  int M, N, K;
  int lda, ldw, ldc;
  int amode;
  int T;               // TAPS3: sequence length (rows per batch element)
  int Kt;              // TAPS3: per-tap K;  CONV2D: Cin
  int Hin, Win, Hout, Wout;  // CONV2D
  int act;             // ACT_*
  const float* R;      // residual / positional-encoding rows, or nullptr:  C += R[(m % rperiod)][n]
  int ldr, rperiod;    // rperiod <= 0: plain residual (row m)
  float* C2;           // mask mode (SeparationDecoder.separate, model.py:220): C2 = C * X[m][n % F]
  const float* X;
  int ldx, F;
  // FRAMES mode (STFT front-end, dataset.py:122-135): row m = (clip b, frame t) of T frames per clip is the window
  // audio[b][t*frame_hop .. + K) of a (B, frame_len) signal -- overlapping rows read straight from the waveform, samples
  // at or beyond frame_len read as zero (the reference zero-pads tail frames).  frame_hop % 4 == frame_len % 4 == 0.
  // With mag_F > 0 the epilogue takes columns (2f, 2f+1) as (re, im) of bin f and stores |.| to C[(b*mag_F + f)*T + t],
  // i.e. the reference's (B, F, T) spectrogram layout; bias / act / R / C2 are not used then.
  int frame_hop, frame_len, mag_F;
  int no_xcd_remap;    // developer switch: 1 = launch-order tiles (A/B measurements)
  int split_t2_min;    // split-precision GEMM: 256 x 128 kernel from this many of its tiles on (0: 128, the single-stream optimum)
  int epi_general;           // developer A/B switch (AVSEP_EPI_GENERAL): 1 = block-by-block epilogue everywhere
  unsigned long long* dbg;   // developer diagnostics (AVSEP_GEMM_DBG): per-workgroup phase stamps, null otherwise
  // Fused LayerNorm prologue (PLAIN mode, K == normalised width): A' = (A - mean_row) * rstd_row * gamma + beta,
  // row statistics computed in-kernel by a pre-pass over the block's rows (nn.LayerNorm, eps ln_eps).
  // With ln_stats (row m: mean at [2m], 1/sqrt(var+eps) at [2m+1], from launch_layernorm_stats) the statistics are not
  // recomputed: A' is formed while the tile is staged to LDS (AMODE_LN, any tile, K <= LN_KMAX) -- the form for large M,
  // where the in-kernel pre-pass of every column tile over the same rows would cost more than one statistics launch.
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  const float* ln_stats;
  // LayerNorm in the EPILOGUE (AMODE_LNX, round 3).  With W' = W o gamma, c1[n] = sum_k W'[n][k], c2[n] = sum_k W[n][k] beta[k]
  // + bias[n] (all made once by the weight packer):
  //     LayerNorm(x) W^T + b  =  rstd_m ( x W'^T )[m][n]  -  rstd_m mean_m c1[n]  +  c2[n]
  // so the GEMM runs on the RAW rows with the ring-pipelined staging of the plain kernel -- no whole-K register slab, no
  // statistics pass in front of the first MFMA -- and the row statistics are summed on the side from the A chunks the
  // workgroup stages anyway (shifted one-pass sums, the formula of gemm_ln_kernel) and applied after the K loop.
  // W = W' here, bias = null, K = the normalised width.  The same function as LayerNorm launch + GEMM, not the same bits: the
  // rounding error grows with |mean| / std of the rows (0.4-0.9 at the model's LayerNorm sites; against float64 the two
  // forms measure the same error on every fixture).  Every instance has BK = 32: the side sums then run over
  // the same 8 lanes per row in the same order on every tile, so the result does not depend on the tile or the batch size.
  const float* lnx_c1;
  const float* lnx_c2;
  // Split-K (weight gradients: K = rows >> M, N): gridDim.y = ksplit slices of kchunk columns each (kchunk % 64 == 0),
  // slice z reads A/W columns [z*kchunk, ...) and writes its partial product to C + z*cstride; the caller sums the
  // slices with a column reduction (deterministic, no atomics).  PLAIN mode, no bias/act/residual.  0/1 = off.
  int ksplit, kchunk;
  long long cstride;
  // Pair launch (launch_gemm_pair): a second problem with the SAME N, K, leading dimensions, A mode, activation and
  // epilogue kind rides in the same launch -- the audio and the visual instance of an encoder-layer GEMM
  // (nn.TransformerEncoderLayer of AudioEncoder / VisualEncoder, model.py:48-52 / 97-101: same shapes of weights, M = B*T
  // and B*N rows).  Virtual tiles [0, g_tiles0) belong to the problem above, the others to `alt`; g_tiles0 is filled in by
  // the launcher once the tile is chosen (alt.M > 0 marks a pair).
  // Training: inverted dropout in the epilogue, y = residual + (keep ? act(acc + bias) / (1 - p) : 0) with the stateless
  // mask of dropout_keep() on the element index m*N + n -- the same values, bit for bit, as a dropout / dropout_add launch
  // behind the GEMM (the backward regenerates the mask from the seed).  drop_p = 0: off.  Needs ldc == N.
  float drop_p;
  unsigned long long drop_seed;
  int g_tiles0;
  // Set by the launcher (fp32 MFMA and VALU instructions do not overlap on a SIMD, profiles/r03_mfma_valu_exclusive.txt, so
  // what can leave the VALU leaves it): nbn_magic = floor(2^32 / column tiles) + 1, the workgroup's tile row is
  // mulhi(tile, nbn_magic) on the scalar unit instead of an integer division (a ~25-instruction VALU sequence even for
  // uniform operands); 0 = divide (one column tile, or tiles x column tiles >= 2^32).  ln_inv_k = 1 / K (AMODE_LNX).
  unsigned nbn_magic;
  float ln_inv_k;
  // TAPS3: at least Kt floats of zeros.  A tap outside its sequence (t - 1 < 0, t + 1 >= T) reads THIS row instead of being
  // loaded from a valid address and zeroed by four selects per float4 on the way to LDS: the same zeros without the VALU.
  const float* zeros;
  // Pre-split operands (gemm_planes.hip, round 5): the three bf16 terms of A / W as "P32" plane buffers -- bf16 [K/32][3][rows][32],
  // element (m, k) of term t at ((k/32 * 3 + t) * rows + m) * 32 + k % 32 -- cut ONCE by the operand's producer instead of in every
  // column tile of every GEMM; a_rows / w_rows = the buffers' row counts (>= M / N).  Cp (c_rows): the result written as the
  // planes of the NEXT GEMM's A operand (bias + activation only), instead of or beside the fp32 C.
  const unsigned short* Ap;
  const unsigned short* Wp;
  unsigned short* Cp;
  long long a_rows, w_rows, c_rows;
  // Two-term fp16 planes (gemm_h2.hip): operands stored scaled by powers of two -- cscale[n] = 2^-(eA + ew[n]) brings an accumulator
  // back to true scale (exact), cp_scale = 2^eC is the scale of the plane OUTPUT (the consumer's static exponent).
  const float* cscale;
  const float* rscale;   // per-ROW descale of an operand that carries one power of two per row (the resized visual stream), or null
  float cp_scale;
  int h2;              // 1: Ap / Wp / Cp hold two fp16 terms (gemm_h2.hip), 0: three bf16 terms (gemm_planes.hip)
#ifdef AVSEP_DEV
  struct Alt {
    const float *A, *W, *bias, *R, *ln_gamma, *ln_beta;
    float* C;
    int M, rperiod;
  } alt;
#else
  // product library: no pair launches (measured slower, profiles/r03_ab_paired_schedule.txt) -- no second problem in the
  // kernel arguments, select_pair() compiles to nothing, launch_gemm_pair() does not exist
  struct Alt { static constexpr int M = 0; } alt;
#endif
};

hipError_t launch_gemm(const GemmParams& p, hipStream_t s);
// p0 and p1 in ONE launch; they must agree in everything but A, W, bias, C, R (+ rperiod), the LayerNorm vectors and M
#ifdef AVSEP_DEV
hipError_t launch_gemm_pair(const GemmParams& p0, const GemmParams& p1, hipStream_t s);
#endif
// split-precision GEMM (gemm_split.hip): fp32 operands as three bf16 terms each, six bf16 MFMA products, fp32 accumulation;
// fp32-equivalent results at 6 / 16 of the fp32 MFMA's matrix time.  PLAIN A operand, fast epilogue.
bool gemm_split_supported(const GemmParams& p);
hipError_t launch_gemm_split(GemmParams p, hipStream_t s);
const char* gemm_split_instance_name(const GemmParams& p);
// the same GEMM on PRE-SPLIT operands (gemm_planes.hip): GemmParams::Ap / Wp (/ Cp), staged by LDS-DMA; same bits as the kernels above
bool gemm_planes_supported(const GemmParams& p);
hipError_t launch_gemm_planes(GemmParams p, hipStream_t s);
const char* gemm_planes_instance_name();
const char* layernorm_planes_instance_name(int d, int kind);   // kind 1: three bf16 terms, 2: two fp16 terms
// the same on TWO fp16 terms and THREE products per fp32 product (gemm_h2.hip): GemmParams::Ap / Wp as H2 planes, cscale (/ Cp, cp_scale)
bool gemm_h2_supported(const GemmParams& p);
hipError_t launch_gemm_h2(GemmParams p, hipStream_t s);
const char* gemm_h2_instance_name(const GemmParams& p);
// x [M][ld] fp32 -> H2 planes (fp16 [K/32][2][rows][32]) of x * 2^e, e = row_exp[m] (or the one exponent `e` when row_exp is null)
hipError_t launch_split_h2(const float* x, int ld, unsigned short* planes, long long rows, int M, int K, const int* row_exp, int e,
                           hipStream_t s);
// per row n of w [N][K]: ew[n] = the exponent that puts max|w[n][:]| into [2^13, 2^14) (0 for a zero row), l2[n] >= ||w[n][:]||_2
hipError_t launch_h2_row_stats(const float* w, int N, int K, int* ew, float* l2, hipStream_t s);
// x [M][ld] fp32 -> its three bf16 terms in P32 plane format (rows >= M: the buffer's row count); K % 32 == 0, ld % 4 == 0
hipError_t launch_split_planes(const float* x, int ld, unsigned short* planes, long long rows, int M, int K, hipStream_t s);
// split-precision attention (attention_split.hip): dh = 64, fp32 in / out, QK^T and PV as six bf16 MFMA products per fp32 product
bool attention_split_supported(int dh, int Lq, int Lk);
// op != null: the output is written as the bf16 planes of the out-projection GEMM's A operand (GemmParams::Ap; row b * Lq + q, column
// h * 64 + c of a buffer with o_rows rows) INSTEAD of fp32 o
hipError_t launch_attention_split(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                  int B, int nhead, int dh, int Lq, int Lk, float qscale, hipStream_t s,
                                  unsigned short* op = nullptr, long long o_rows = 0, int h2 = 0, int h2_exp = 0);
bool attention_h2_supported(int dh, int Lq, int Lk, int eq, int ek, int ev);
// clip_exp (stride clip_stride ints per clip: {ek, ev}): per-clip exponents of k and v instead of ek / ev; the plane output is then
// scaled by the clip's 2^ev and rs_out[b Lq + q] = 2^-ev is written for the out-projection (GemmParams::rscale)
hipError_t launch_attention_h2(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo, int B,
                               int nhead, int dh, int Lq, int Lk, int eq, int ek, int ev, hipStream_t s, unsigned short* op = nullptr,
                               long long o_rows = 0, int h2_exp = 0, const int* clip_exp = nullptr, int clip_stride = 0,
                               float* rs_out = nullptr);
hipError_t launch_clip_exp(const float* rscale, const float* kc, int* out, int B, int T, int Lf, hipStream_t s);   // h2: two fp16 terms scaled by 2^h2_exp
// weight gradient dW[N][K] = dY^T X from row-major dY [R][ldy], X [R][ldx] (gemm.hip wgrad_kernel)
int wgrad_slices(int N, int K, int R);
// with_bias: each slice is N*K + N floats, the last N = column sums of dy (the bias gradient)
// merged + counters (slices > 1): the slices are summed inside the launch by each tile's last-arriving workgroup into
// `merged` ([N][K] (+ [N])); counters: wgrad_tiles() zero-initialised words, left at zero
hipError_t launch_wgrad(const float* dy, int ldy, const float* x, int ldx, float* out, int N, int K, int R, int slices,
                        bool with_bias, hipStream_t s, float* merged = nullptr, unsigned* counters = nullptr);
int wgrad_tiles(int N, int K, int R);
// the same on the split-precision bf16 pipe (wgrad_split.hip): 64 x 64 tiles, the slices / output layout of launch_wgrad
hipError_t launch_wgrad_split(const float* dy, int ldy, const float* x, int ldx, float* out, int N, int K, int R, int slices,
                              bool with_bias, hipStream_t s);
const char* gemm_instance_name(const GemmParams& p);
bool gemm_ln_supported(int K);                          // can launch_gemm() fuse a LayerNorm over K columns?   // template instance launch_gemm() will pick

// ---- dependency-driven persistent launch of a chain of small ops (chain.hip)
struct AttnProblem;
struct ChainBuilder;
struct ChainPlanImpl;
ChainBuilder* chain_builder_new();
void chain_builder_free(ChainBuilder* b);
// ops are added in execution order; `dep` = index (return value) of the op that produces this op's input rows, or -1 when they
// come from an earlier launch; L = rows per clip of the sequence tensor.  -1 = the op cannot run as a chain tile.
int chain_add_gemm(ChainBuilder* b, const GemmParams& p, int dep, int L);
int chain_add_attention(ChainBuilder* b, const AttnProblem& a, int nhead, int dep);
// device tables of the work list (hipMalloc + blocking copies: never call it while a stream capture is in progress)
hipError_t chain_build(ChainBuilder* b, int xcd_local, int order_group, float order_skew, ChainPlanImpl** out);
void chain_plan_free(ChainPlanImpl* p);
double chain_plan_flops(const ChainPlanImpl* p);
double chain_plan_bytes(const ChainPlanImpl* p);
int chain_plan_items(const ChainPlanImpl* p);
hipError_t launch_chain(const ChainPlanImpl* p, hipStream_t s);                       // memset of the counters + ONE kernel
hipError_t chain_plan_error(const ChainPlanImpl* p, hipStream_t s, unsigned* word);    // word 0 = no wait timed out (synchronises s)
hipError_t chain_plan_peek(const ChainPlanImpl* p, hipStream_t s, unsigned* out, int n);   // developer aid

hipError_t launch_layernorm(const float* x, const float* g, const float* b, float* y, int M, int d, float eps,
                            hipStream_t s);
// row statistics of nn.LayerNorm only: stats[2m] = mean, stats[2m+1] = 1/sqrt(biased var + eps), the values
// layernorm_kernel normalises with (same reduction order)
hipError_t launch_layernorm_stats(const float* x, float* stats, int M, int d, float eps, hipStream_t s);
bool gemm_ln_staged_supported(int K);                   // can launch_gemm() take ln_stats for a LayerNorm over K columns?
hipError_t launch_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                            float* o, int ldo, int B, int nhead, int dh, int Lq, int Lk, hipStream_t s);
// two attention problems (same head count / head size) in one launch when both take the same kernel instance, else two
// launches on `s`
struct AttnProblem { const float *q, *k, *v; float* o; int ldq, ldk, ldv, ldo, B, Lq, Lk; };
hipError_t launch_attention_pair(const AttnProblem& a, const AttnProblem& b, int nhead, int dh, hipStream_t s);
// x += softmax(q k^T) v W_o^T + b_o in ONE launch (short sequences: dh = 64, 49..64 keys, nhead <= 8, d = 64 nhead);
// hipErrorNotSupported otherwise (callers then launch attention and the projection GEMM separately)
bool attn_proj_supported(int nhead, int dh, int Lk);
const char* attn_proj_instance_name(int nhead);
hipError_t launch_attn_proj(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* wo,
                            const float* bo, float* x, int B, int nhead, int dh, int Lq, int Lk, hipStream_t s);
bool attention_pair_merges(int dh, int Lk_a, int Lk_b);                          // does the pair become ONE launch?
// the kernel instance launch_attention() picks for an inference call, spelled as rocprofv3 prints it
const char* attention_instance_name(int dh, int Lq, int Lk, int B, int nhead);
hipError_t launch_attention_ex(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                               float* o, int ldo, int B, int nhead, int dh, int Lq, int Lk, float qscale, float* lse,
                               float drop_p, unsigned long long drop_seed, hipStream_t s);
hipError_t launch_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                const float* o, int ldo, const float* dO, int lddo, const float* lse, float* dvec,
                                float* dq, int lddq, float* dk, int lddk, float* dv, int lddv, int B, int nhead, int dh,
                                int Lq, int Lk, float qscale, float drop_p, unsigned long long drop_seed, hipStream_t s);
hipError_t launch_dropout(const float* x, float* y, size_t n, float p, unsigned long long seed, hipStream_t s);
hipError_t launch_dropout_add(const float* x, const float* r, float* y, size_t n, float p, unsigned long long seed,
                              hipStream_t s);   // y = r + dropout(x)
// training-path kernels (train_ops.hip)
hipError_t launch_transpose2d(const float* x, float* y, int R, int C, int Rp, hipStream_t s);
// one launch for a device-resident table of transpositions (layout = avsep_transpose_desc of include/avsep.h)
struct TransposeDesc { const float* src; float* dst; int R, C, Rp, pad; };
hipError_t launch_transpose_many(const TransposeDesc* table_dev, int n, int max_rp, int max_c, hipStream_t s);
hipError_t launch_im2col1d(const float* x, float* col, int M, int T, int C, hipStream_t s);
hipError_t launch_col2im1d(const float* dcol, float* dx, int M, int T, int C, hipStream_t s);
hipError_t launch_im2col2d(const float* x, float* col, int I, int H, int W, int C, int Ho, int Wo, int Kp, hipStream_t s);
hipError_t launch_col2im2d(const float* dcol, float* dx, int I, int H, int W, int C, int Ho, int Wo, int Kp, hipStream_t s);
hipError_t launch_colreduce(const float* a, const float* b, float* part, float* out0, float* out1, int M, int C,
                            float scale, hipStream_t s);
int colreduce_part_floats(int M, int C);
hipError_t launch_sum_slices(const float* part, float* out, int S, size_t n, hipStream_t s);
hipError_t launch_bn_var(const float* x, const float* mean, float* part, float* out, int M, int C, hipStream_t s);
hipError_t launch_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                           float* xhat, float* y, size_t n, int C, int relu, float eps, hipStream_t s);
hipError_t launch_bn_bwd(const float* dyr, const float* xhat, const float* gamma, const float* var,
                         const float* sum_dy, const float* sum_dyx, float* dx, size_t n, int C, float invM, float eps,
                         hipStream_t s);
hipError_t launch_bn_running(float* rmean, float* rvar, const float* mean, const float* var, int C, float momentum,
                             int M, hipStream_t s);
hipError_t launch_act_fwd(const float* x, float* y, size_t n, int act, hipStream_t s);
hipError_t launch_act_bwd(const float* dy, const float* aux, float* dx, size_t n, int act, hipStream_t s);
// backward of y = dropout(relu(z)) from y alone: dx = y > 0 ? dy / (1 - p) : 0  (y > 0 <=> kept and z > 0)
hipError_t launch_relu_dropout_bwd(const float* dy, const float* y, float* dx, size_t n, float p, hipStream_t s);
hipError_t launch_mul_mixed(const float* a, const float* xt, float* out, size_t M, int S, int F, int ldx, hipStream_t s);
hipError_t launch_add_rows(const float* x, const float* r, float* y, size_t M, int C, int period, hipStream_t s);
hipError_t launch_avgpool_bwd(const float* dy, float* dx, int M, int P, int C, hipStream_t s);
hipError_t launch_interp_bwd(const float* dy, float* dx, int B, int N, int T, int d, hipStream_t s);
// dres (may be null): added to dx -- the gradient that reaches x along the residual path of a pre-norm block
hipError_t launch_layernorm_bwd_affine(const float* dy, const float* x, const float* gamma, const float* dres, float* dx,
                                       float* dgamma, float* dbeta, float* part, int M, int d, float eps, hipStream_t s);
hipError_t launch_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* dres, float* dx,
                                float* xhat, int M, int d, float eps, hipStream_t s);
// (B,F,T) -> (B,T,Fp) zero padded
hipError_t launch_transpose_pad(const float* x, float* y, int B, int F, int T, int Fp, hipStream_t s);
// first visual conv (Cin=1) + folded BN + ReLU: frames (M,H,W) -> act (M,Ho,Wo,32) channels-last
hipError_t launch_conv1_c1(const float* frames, const float* w9x32, const float* bias32, float* out, int M,
                           int H, int W, int Ho, int Wo, hipStream_t s);
// whole visual conv front-end (3 x conv+BN+ReLU + average pool) in one LDS-resident kernel; returns
// hipErrorNotSupported when the frame size does not fit (callers fall back to the three launches above/below)
// h2 (may be null: the fp32 MFMA kernel): conv2 / conv3 on two fp16 terms (conv_stack_h2_kernel) -- their weights as H2 planes
// [Co][9][2][Ci] with row exponents, sc = 2^-ew, and the constants of the per-pass activation bounds (avsep_finalize_weights)
struct ConvH2 {
  const unsigned short *w2h, *w3h;
  const float *sc2, *sc3;
  float s1max, b1max, l2max2, b2max;
};
hipError_t launch_conv_stack(const float* frames, const float* w1, const float* b1, const float* w2,
                             const float* b2, const float* w3, const float* b3, float* pooled, int Mv, int H,
                             int W, hipStream_t s, const ConvH2* h2 = nullptr);
hipError_t launch_pack_conv_h2(const float* w, const int* ew, unsigned short* wh, float* sc, int Co, int Ci, hipStream_t s);
// mean over P positions: x (M,P,C) -> y (M,C)
hipError_t launch_avgpool(const float* x, float* y, int M, int P, int C, hipStream_t s);
hipError_t launch_interp_linear(const float* x, float* y, int B, int N, int T, int d, hipStream_t s);
// the same values as the bf16 planes of the consuming GEMM's A operand (GemmParams::Ap; rows = the buffer's row count); d % 32 == 0
hipError_t launch_interp_linear_planes(const float* x, unsigned short* yp, long long rows, int B, int N, int T, int d, hipStream_t s);
// ... as two fp16 terms with one power of two per ROW (from the row's largest magnitude); rscale[row] = 2^-e for the GEMM's epilogue
hipError_t launch_interp_linear_h2(const float* x, unsigned short* yp, float* rscale, long long rows, int B, int N, int T, int d,
                                   hipStream_t s);
hipError_t launch_layernorm_planes(const float* x, const float* g, const float* b, unsigned short* yp, long long rows, int M, int d,
                                   float eps, hipStream_t s);
// ... as the two fp16 terms of gemm_h2.hip, scaled by 2^e
hipError_t launch_layernorm_h2(const float* x, const float* g, const float* b, unsigned short* yp, long long rows, int M, int d,
                               float eps, int e, hipStream_t s);
// Hann-windowed real-DFT basis [2*(n_fft/2+1)][n_fft]: row 2f = w[k] cos(2 pi f k / n_fft), row 2f+1 = -w[k] sin(..),
// w = np.hanning(n_fft) (symmetric); evaluated in double precision, rounded once
hipError_t launch_stft_basis(float* basis, int n_fft, hipStream_t s);

// instance names as rocprofv3 prints them (namespace prefix and argument list stripped): the live profiler's table joins
// profiles/*_kernel_stats.csv and profiles/pmc_hbm_traffic.json by string equality
const char* layernorm_instance_name(int d, bool stats_only);
const char* conv_stack_instance_name(int Mv, int H, int W, bool h2 = false);   // h2: conv_stack_h2_kernel

hipError_t launch_delay(unsigned us, hipStream_t s);
hipError_t launch_stamp(unsigned long long* buf, int idx, hipStream_t s);   // profiling aid, see rowops.hip

// weight packing (device -> device)
hipError_t launch_pack_rows(const float* src, float* dst, int rows, int K, int Kp, float scale, int scale_rows,
                            hipStream_t s);  // dst[r][k] = src[r][k] * (r < scale_rows ? scale : 1), zero pad to Kp
hipError_t launch_pack_conv1d(const float* w, float* dst, int Co, int Ci, int Cip, hipStream_t s);  // (Co,Ci,3)->[Co][3][Cip]
// (Co,Ci,3,3) + BN(gamma,beta,mean,var) -> wp [Co][9][Ci] (or [9][Co] when Ci==1), bp [Co]
hipError_t launch_pack_conv2d_bn(const float* w, const float* b, const float* gamma, const float* beta,
                                 const float* mean, const float* var, float* wp, float* bp, int Co, int Ci,
                                 float eps, hipStream_t s);
hipError_t launch_scale_copy(const float* src, float* dst, int n, float scale, int scale_n, hipStream_t s);
// operands of the LayerNorm-in-the-epilogue GEMM (GemmParams::lnx_c1) from a packed weight [N][K] and bias [N] (or null):
// wp = w o gamma, c1[n] = sum_k wp[n][k], c2[n] = sum_k w[n][k] beta[k] + bias[n]  (sums in double, rounded once)
hipError_t launch_pack_lnx(const float* w, const float* bias, const float* gamma, const float* beta, float* wp, float* c1,
                           float* c2, int N, int K, hipStream_t s);
