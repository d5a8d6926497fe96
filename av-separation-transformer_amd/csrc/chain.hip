// chain.hip -- a dependency-driven, persistent launch for chains of small dependent ops (round 4).
//
// The 32-clip step of BASELINE configs[1] is ~54 launches of 7-13 us on two streams; every launch boundary is a device-wide
// barrier although NO dependency of the eval forward crosses a clip (/root/reference/src/av_separation/model.py:54-60, 103-117,
// 166-173, 201-208: attention per clip, everything else per row).  Here a run of ops (the pre-norm encoder layers of one
// branch: LayerNorm-epilogue QKV GEMM -> short-sequence attention -> out-projection + residual -> LayerNorm-epilogue FFN-1 ->
// FFN-2 + residual, model.py:48-52 / 97-101 via nn.TransformerEncoderLayer) is ONE launch of resident workgroups that pull
// tiles from a topologically ordered work list:
//
//   * item = one tile of one op: a 32-row x 64- (LayerNorm-epilogue) or 32-column (plain) GEMM tile, or the 4 query tiles of one
//     (clip, head) of the attention.  The tile code is the code of the stand-alone kernels (gemm_tile.h, attn_tile.h): same
//     k order, same epilogues, so every output bit equals the launch-per-op path's (tests/test_gpu_parity.py).
//   * every op owns one arrival counter per 32-row block (GEMMs) or per clip (attention).  A tile starts when the counters of the
//     producer units its rows need have reached the producer's tile count -- a clip's attention waits for the 2-3 row blocks of
//     QKV that cover the clip, a row block of the out-projection for the 1-2 clips it touches, a GEMM behind a GEMM for its own
//     row block -- and adds 1 to its own unit's counter when its stores have drained.
//   * hand-off (cdna_hip_programming.md Guideline 16, form R1): every activation written inside the launch is stored
//     write-through (buffer_store ... sc1), every storing wave drains (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE
//     lane adds to the counter (agent scope); the consumer's first wave polls the counters with sc1 loads, the workgroup meets at
//     a barrier, and EVERY load of handed-off bytes is an L1-bypassing sc1 load.  No release / acquire fence, no dependence on
//     which XCD a workgroup landed on.  Weights, biases and the LayerNorm side vectors are read-only for the launch: plain loads.
//   * tickets come from one agent-scope atomic counter; the list order is a topological order, so a workgroup only ever waits for
//     tiles with LOWER tickets, all of which have been claimed by running workgroups: no co-residency requirement, no deadlock.
//     Every spin is bounded; a timeout sets an error word and the grid still drains.
//   * counters, the ticket words and the error word are zeroed by a memset node in front of the launch (replayed with it).
//
// Two hand-off forms (avsep_set_schedule 1 / 2; both bit-identical to the launch-per-op path):
//   1  ONE work queue, placement-independent: write-through (sc1) stores as above.  Measured (profiles/r04_chain_phase_stamps_sc1.txt):
//      an sc1 store DROPS its line from the XCD's L2, so every A-operand load of every consumer tile -- 12 column tiles re-read
//      each QKV row block -- goes to memory: tiles take 15-30 us where the stand-alone kernels' live 8.
//   2  XCD-LOCAL queues: the batch is cut into (up to) 8 groups of clips, group g's tiles (group-aligned row tiling, so no tile
//      straddles two groups) sit in queue g, and a workgroup pulls ONLY from the queue of the XCD it runs on, which it reads from
//      the hardware (s_getreg HW_REG_XCC_ID) -- not from its block index.  Producer and consumer of every hand-off therefore share
//      one L2: activations are stored PLAIN (the line stays in that L2), drained, signalled; consumers read them with L1-bypassing
//      sc1 loads, which that L2 serves.  Correct for any dispatch order; if the runtime gave an XCD no workgroup its queue would
//      not drain, which the launch's status word reports (chain_plan_error) instead of hanging.
#include "kernels.h"
#ifdef AVSEP_DEV   // measured slower than the launch-per-op schedule: developer build only, like every rejected instance
#include "gemm_tile.h"
#include "attn_tile.h"

#include <algorithm>
#include <cstdio>
#include <vector>

namespace {

enum { CH_GEMM_PLAIN = 0, CH_GEMM_LNX = 1, CH_ATTN = 2 };

struct ChainOp {              // device table, one per op; every field is wave-uniform
  int kind, act;
  const float *A, *W, *bias, *R, *c1, *c2;     // GEMM operands (R: residual or null; c1 / c2: LayerNorm-epilogue vectors)
  float* C;
  int M, N, K, lda, ldc, ldr;
  float eps, inv_k;
  const float *q, *k, *v;                      // attention operands (head h at column offset 64 h)
  float* o;
  int ldq, ldk, ldv, ldo, Lq, Lk, nhead, pad;
};

struct ChainItem {            // device table, one per ticket
  int op, m0, n0;             // GEMM: tile origin; attention: m0 = clip, n0 = head
  int dep_lo, dep_n, dep_target;   // counters [dep_lo, dep_lo + dep_n) must each have reached dep_target (dep_n = 0: no wait)
  int sig;                    // counter this tile adds 1 to when its stores have drained
  int mend;                   // GEMM: row bound of the tile (end of its clip group; rows at or beyond it are not computed)
};

struct ChainArgs {
  const ChainOp* ops;
  const ChainItem* items;     // queue q = items [qstart[q], qstart[q + 1])
  const int* qstart;          // 9 entries (device)
  unsigned* state;            // ticket head of queue q at word 16 q, error word at 128, counters from CHAIN_STATE_HDR
  unsigned long long* dbg;    // developer diagnostics (AVSEP_CHAIN_DBG): 4 wall-clock stamps per ticket, else null
  int backoff;                // s_sleep argument of the dependency poll's back-off
};

enum { CHAIN_STATE_HDR = 144, CHAIN_ERR_WORD = 128, CHAIN_SPIN_LIMIT = 40000 };

template <typename T>
__device__ __forceinline__ T uni(T v) {       // the value is wave-uniform: move it to SGPRs
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- or 8-byte scalars");
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
  } else {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
  }
}

// WPS = workgroups per CU the register budget is cut for: 4 (128 VGPRs: the persistent loop keeps lane constants of three tile
// kinds alive and spills ~25 of them, a few reloads inside the K loops) or 3 (168 VGPRs, no spill)
template <int WPS, int XL>
__global__ __launch_bounds__(256, WPS) void chain_kernel(const ChainArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[8192];     // 32 KB: the 32x32x64 plain tile; the 32x64x32 tile takes 24 KB
  __shared__ int s_ticket;
  constexpr int COH = XL ? 2 : 1;                              // XL: plain stores (same-XCD consumers), else write-through stores
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // provably wave-uniform: branches on it are scalar branches
  // XL: the queue of the XCD this workgroup RUNS on (hardware register, bits 3:0), never a guess from the block index
  const int q = XL ? (int)(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 7u) : 0;
  const int q_lo = uni(a.qstart[q]), q_hi = uni(a.qstart[q + 1]);
  unsigned* const head = a.state + 16 * q;
  unsigned* const err = a.state + CHAIN_ERR_WORD;
  unsigned* const cnt = a.state + CHAIN_STATE_HDR;
  // Control flow note: NO divergent branch may straddle the loop's back edge.  The first version fetched the ticket in an
  // `if (tid == 0)` at the loop top and signalled in another at the bottom; hipcc merged the two into one divergent region around
  // the back edge (lane 0 parked in an outer loop while lanes 1-63 of its wave went round the inner one): the wave re-entered the
  // barrier without lane 0, re-read the OLD ticket and ran the same tile forever.  Now lane 0's work (signal + next ticket) sits
  // in straight-line code between two barriers, under a scalar `wave == 0` branch, and the loop condition is an SGPR compare.
  if (wave == 0) {
    if (lane == 0) s_ticket = q_lo + (int)__hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  int t = uni(s_ticket);
  while (t < q_hi) {
    const ChainItem* ip = a.items + t;
    const int it_op = uni(ip->op), it_m0 = uni(ip->m0), it_n0 = uni(ip->n0);
    const int dep_lo = uni(ip->dep_lo), dep_n = uni(ip->dep_n), dep_target = uni(ip->dep_target), sig = uni(ip->sig);
    const int mend = uni(ip->mend);
    const ChainOp* op = a.ops + it_op;
    const int kind = uni(op->kind);
    if (a.dbg && tid == 0) a.dbg[4 * (size_t)t] = __builtin_amdgcn_s_memrealtime();
    if (dep_n > 0) {                                                    // block-uniform
      if (wave == 0) {
        const unsigned* w = cnt + dep_lo + min(lane, dep_n - 1);
        for (int spins = 0;; ++spins) {
          const unsigned v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_load_dword sc1
          if (__all((int)v >= dep_target)) break;
          if (spins > CHAIN_SPIN_LIMIT) {                               // never hang the device: flag it and go on (wrong data)
            if (lane == 0) __hip_atomic_store(err, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          // back off: every poll is a load that goes past L1 / L2 to the counter's home, and hundreds of waiting workgroups
          // polling a handful of words slow down the very adds they wait for (MI355X_MICROARCH.md, polling-cost)
          if (a.backoff <= 2) __builtin_amdgcn_s_sleep(2);
          else if (a.backoff <= 16) __builtin_amdgcn_s_sleep(16);
          else if (a.backoff <= 48) __builtin_amdgcn_s_sleep(48);
          else __builtin_amdgcn_s_sleep(127);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");             // no instruction: keeps the tile's loads below the poll
      __syncthreads();
    }
    if (a.dbg && tid == 0) a.dbg[4 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
    if (kind == CH_ATTN) {
      const int Lq = uni(op->Lq), Lk = uni(op->Lk), ldq = uni(op->ldq), ldk = uni(op->ldk), ldv = uni(op->ldv), ldo = uni(op->ldo);
      const int nqt = (Lq + 15) >> 4;
      if (wave < nqt) {                                                 // wave = 16-query tile of (clip it_m0, head it_n0)
        const int c = lane & 15, g = lane >> 4;
        const float* qb = uni(op->q) + (size_t)it_m0 * Lq * ldq + it_n0 * 64;
        const float* kb = uni(op->k) + (size_t)it_m0 * Lk * ldk + it_n0 * 64;
        const float* vb = uni(op->v) + (size_t)it_m0 * Lk * ldv + it_n0 * 64;
        float* ob = uni(op->o) + (size_t)it_m0 * Lq * ldo + it_n0 * 64;
        f32x4 acc[4];
        float lrun;
        attn_short_tile<4, 4, true>(qb, kb, vb, ldq, ldk, ldv, Lq, Lk, wave, c, g, acc, lrun);
        lrun = rows_sum(lrun);
        const float inv = 1.0f / lrun;
        const int qo = wave * 16 + c;
        if (qo < Lq) {
          const __amdgpu_buffer_rsrc_t ro = coh_rsrc(ob);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4 ov = f32x4{acc[0][r] * inv, acc[1][r] * inv, acc[2][r] * inv, acc[3][r] * inv};
            if (COH == 1) coh_store16(ro, (qo * ldo + 4 * (4 * g + r)) * 4, ov);
            else *reinterpret_cast<f32x4*>(ob + (size_t)qo * ldo + 4 * (4 * g + r)) = ov;
          }
        }
      }
    } else {
      GemmParams p{};
      p.A = uni(op->A); p.W = uni(op->W); p.bias = uni(op->bias); p.C = uni(op->C);
      p.M = mend; p.N = uni(op->N); p.K = uni(op->K);
      p.lda = uni(op->lda); p.ldw = p.K; p.ldc = uni(op->ldc);
      p.act = uni(op->act);
      p.R = uni(op->R); p.ldr = uni(op->ldr);
      p.ln_eps = uni(op->eps); p.ln_inv_k = uni(op->inv_k);
      if (kind == CH_GEMM_LNX) {
        p.amode = AMODE_LNX;
        p.lnx_c1 = uni(op->c1); p.lnx_c2 = uni(op->c2);
        gemm_tile<32, 64, 32, AMODE_LNX, false, 0, COH>(p, it_m0, it_n0, lds);
      } else {
        p.amode = AMODE_PLAIN;
        gemm_tile<32, 32, 64, AMODE_PLAIN, false, 0, COH>(p, it_m0, it_n0, lds);
      }
    }
    // publish: every storing wave drains its write-through stores, the workgroup meets, ONE lane signals -- and draws the
    // workgroup's next ticket
    if (a.dbg && tid == 0) a.dbg[4 * (size_t)t + 2] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (a.dbg && tid == 0) a.dbg[4 * (size_t)t + 3] = __builtin_amdgcn_s_memrealtime();
    if (wave == 0) {
      if (lane == 0) {
        __hip_atomic_fetch_add(cnt + sig, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ticket = q_lo + (int)__hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    t = uni(s_ticket);
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------ host side
struct ChainPlanImpl {
  ChainOp* ops_dev = nullptr;
  ChainItem* items_dev = nullptr;
  unsigned* state_dev = nullptr;
  int* qstart_dev = nullptr;
  int n_items = 0, n_ops = 0, grid = 0, wps = 4, xcd_local = 0;
  std::vector<int> qstart_host;
  unsigned long long* dbg_dev = nullptr;  // AVSEP_CHAIN_DBG (developer build): stamps per ticket
  int backoff = 48;
  std::vector<ChainItem> items_host;      // diagnostics (chain_plan_error)
  std::vector<int> op_kind_host;
  size_t state_bytes = 0;
  double flops = 0.0, bytes = 0.0;
};

namespace {

struct HostOp {
  ChainOp op;
  int units;                  // counters this op owns: row blocks (GEMM) or clips (attention)
  int tiles_per_unit;         // arrivals that complete a unit
  int dep;                    // producer op index or -1
  int rows_per_clip;          // geometry of the sequence tensor the op works on (for GEMM <-> attention unit mapping)
  int first_counter;
};

}  // namespace

void chain_plan_free(ChainPlanImpl* p);

template <typename F>
static hipError_t chain_dispatch(const ChainPlanImpl* p, F&& f) {      // the kernel instance of a plan
  if (p->xcd_local) return p->wps == 3 ? f(chain_kernel<3, 1>) : f(chain_kernel<4, 1>);
  return p->wps == 3 ? f(chain_kernel<3, 0>) : f(chain_kernel<4, 0>);
}
static hipError_t chain_occupancy(const ChainPlanImpl* p, int* per_cu) {
  return chain_dispatch(p, [&](auto kern) { return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, reinterpret_cast<const void*>(kern), 256, 0); });
}

struct ChainBuilder {
  std::vector<HostOp> ops;
  double flops = 0.0, bytes = 0.0;
};

ChainBuilder* chain_builder_new() { return new (std::nothrow) ChainBuilder(); }
void chain_builder_free(ChainBuilder* b) { delete b; }

// GEMM op from the launch-per-op path's own parameters (PLAIN A operand, or LayerNorm-in-the-epilogue when p.lnx_c1).
// dep: index of the producer op whose output this op's A rows (and residual rows) come from, or -1.  L: rows per clip.
int chain_add_gemm(ChainBuilder* b, const GemmParams& p, int dep, int L) {
  const bool lnx = p.lnx_c1 != nullptr;
  if (p.amode != AMODE_PLAIN || p.C2 || p.mag_F > 0 || p.ksplit > 1 || p.drop_p > 0.0f || (p.N & 3) || (p.ldc & 3) ||
      (p.R && ((p.ldr & 3) || p.rperiod > 0)) || p.ln_gamma || p.ln_stats)
    return -1;
  if (lnx ? (p.K % 32 != 0 || p.bias || p.R) : (p.K % 64 != 0)) return -1;
  if ((long long)p.M * std::max(p.lda, std::max(p.ldc, p.ldr)) * 4 >= (1LL << 31)) return -1;       // 32-bit buffer offsets
  HostOp h{};
  h.op.kind = lnx ? CH_GEMM_LNX : CH_GEMM_PLAIN;
  h.op.act = p.act;
  h.op.A = p.A; h.op.W = p.W; h.op.bias = p.bias; h.op.R = p.R; h.op.c1 = p.lnx_c1; h.op.c2 = p.lnx_c2; h.op.C = p.C;
  h.op.M = p.M; h.op.N = p.N; h.op.K = p.K; h.op.lda = p.lda; h.op.ldc = p.ldc; h.op.ldr = p.ldr;
  h.op.eps = p.ln_eps; h.op.inv_k = 1.0f / (float)p.K;
  const int bn = lnx ? 64 : 32;
  h.units = (p.M + 31) / 32;
  h.tiles_per_unit = (p.N + bn - 1) / bn;
  h.dep = dep;
  h.rows_per_clip = L;
  b->ops.push_back(h);
  b->flops += 2.0 * p.M * p.N * p.K;
  b->bytes += 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N * (p.R ? 2 : 1));
  return (int)b->ops.size() - 1;
}

// Short-sequence attention (dh = 64, 49..64 keys) of B clips x nhead heads; dep: the op that wrote q / k / v (or -1).
int chain_add_attention(ChainBuilder* b, const AttnProblem& a, int nhead, int dep) {
  if (a.Lq <= 0 || a.Lq > 64 || a.Lk <= 48 || a.Lk > 64 || ((a.ldq | a.ldk | a.ldv | a.ldo) & 3)) return -1;
  if ((long long)a.B * a.Lk * std::max(std::max(a.ldq, a.ldk), std::max(a.ldv, a.ldo)) * 4 >= (1LL << 31)) return -1;
  HostOp h{};
  h.op.kind = CH_ATTN;
  h.op.q = a.q; h.op.k = a.k; h.op.v = a.v; h.op.o = a.o;
  h.op.ldq = a.ldq; h.op.ldk = a.ldk; h.op.ldv = a.ldv; h.op.ldo = a.ldo; h.op.Lq = a.Lq; h.op.Lk = a.Lk; h.op.nhead = nhead;
  h.op.M = a.B * a.Lq;
  h.units = a.B;
  h.tiles_per_unit = nhead;
  h.dep = dep;
  h.rows_per_clip = a.Lq;
  b->ops.push_back(h);
  b->flops += 4.0 * a.B * nhead * (double)a.Lq * a.Lk * 64;
  b->bytes += 4.0 * a.B * nhead * 64 * (2.0 * a.Lq + 2.0 * a.Lk);
  return (int)b->ops.size() - 1;
}

// Work list: one item per tile.  The batch is cut into `NG` groups of G clips (xcd_local: up to 8 groups, one queue each; else
// one group = one queue) and every GEMM is tiled in 32-row blocks that START at its group's first row, so no tile straddles two
// groups; a tile's row bound is the end of its group.  Inside a queue the items follow a topological order steered by `skew`:
// ranked by (op index + skew x sub-group of `order_group` clips) and emitted in rank order as soon as every producer tile they
// wait for has been emitted.  skew = 0: op-major (the launch-per-op order without the launch boundaries); skew > 0: sub-groups
// run `skew` ops apart, so that tiles of different ops (latency-bound attention, MFMA-bound FFN) share the CUs.
hipError_t chain_build(ChainBuilder* b, int xcd_local, int order_group, float order_skew, ChainPlanImpl** out) {
  *out = nullptr;
  const int n_ops = (int)b->ops.size();
  if (n_ops == 0) return hipErrorInvalidValue;
  const int L = b->ops[0].rows_per_clip, M = b->ops[0].op.M;
  if (L <= 0 || M % L) return hipErrorInvalidValue;
  const int B = M / L;
  for (auto& h : b->ops)
    if (h.rows_per_clip != L || h.op.M != M) return hipErrorInvalidValue;      // one sequence geometry per chain
  if (xcd_local) {   // one queue per XCD, chosen by HW_REG_XCC_ID & 7: only on a device that exposes eight of them (CPX / DPX
    int dev = 0, xccs = 0;   // partitions or other parts would leave queues whose tickets nobody draws)
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || xccs != 8)
      return hipErrorNotSupported;
  }
  const int G = xcd_local ? (B + 7) / 8 : B;                                    // clips per group
  const int NG = (B + G - 1) / G;
  const int tpg = (G * L + 31) / 32;                                            // row tiles per group
  int n_counters = 0;
  for (auto& h : b->ops) {
    h.units = h.op.kind == CH_ATTN ? B : NG * tpg;
    h.first_counter = n_counters;
    n_counters += h.units;
  }
  struct Raw { ChainItem it; float rank; int queue; };
  std::vector<Raw> raw;
  for (int o = 0; o < n_ops; ++o) {
    const HostOp& h = b->ops[o];
    for (int u = 0; u < h.units; ++u) {
      int g, row_lo, row_hi, gbase, gend;
      if (h.op.kind == CH_ATTN) { g = u / G; row_lo = u * L; row_hi = u * L + L - 1; }
      else { g = u / tpg; row_lo = g * G * L + 32 * (u - g * tpg); row_hi = 0; }
      gbase = g * G * L;
      gend = std::min(gbase + G * L, M);
      if (h.op.kind != CH_ATTN) {
        if (row_lo >= gend) continue;                                           // a short last group: no such tile
        row_hi = std::min(row_lo + 31, gend - 1);
      }
      for (int t = 0; t < h.tiles_per_unit; ++t) {
        ChainItem it{};
        it.op = o;
        if (h.op.kind == CH_ATTN) { it.m0 = u; it.n0 = t; }
        else { it.m0 = row_lo; it.n0 = (h.op.kind == CH_GEMM_LNX ? 64 : 32) * t; it.mend = gend; }
        it.sig = h.first_counter + u;
        if (h.dep >= 0) {
          const HostOp& pr = b->ops[h.dep];
          int lo, hi;
          if (pr.op.kind == CH_ATTN) { lo = row_lo / L; hi = row_hi / L; }
          else { lo = g * tpg + (row_lo - gbase) / 32; hi = g * tpg + (row_hi - gbase) / 32; }
          it.dep_lo = pr.first_counter + lo; it.dep_n = hi - lo + 1; it.dep_target = pr.tiles_per_unit;
          if (it.dep_n > 64) return hipErrorInvalidValue;
        }
        const int sub = order_group > 0 ? (row_lo / L - g * G) / order_group : 0;
        raw.push_back({it, (float)o + order_skew * (float)sub, xcd_local ? g : 0});
      }
    }
  }
  std::stable_sort(raw.begin(), raw.end(), [](const Raw& x, const Raw& y) { return x.queue != y.queue ? x.queue < y.queue : x.rank < y.rank; });
  // emit in (queue, rank) order, an item only after all producer tiles of the counters it polls: a topological order inside
  // every queue by construction (all producers of an item are in its own queue: tiles never straddle groups)
  std::vector<int> emitted(n_counters, 0);
  std::vector<char> done(raw.size(), 0);
  std::vector<ChainItem> items;
  std::vector<int> qstart(9, 0);
  items.reserve(raw.size());
  size_t range_lo = 0;
  for (int qq = 0; qq < 8; ++qq) {
    size_t range_hi = range_lo;
    while (range_hi < raw.size() && raw[range_hi].queue == qq) ++range_hi;
    qstart[qq] = (int)items.size();
    size_t first_open = range_lo, left = range_hi - range_lo;
    while (left > 0) {
      bool progressed = false;
      for (size_t i = first_open; i < range_hi; ++i) {
        if (done[i]) { if (i == first_open) ++first_open; continue; }
        const ChainItem& it = raw[i].it;
        bool ready = true;
        for (int k = 0; k < it.dep_n && ready; ++k) ready = emitted[it.dep_lo + k] >= it.dep_target;
        if (!ready) continue;
        items.push_back(it);
        ++emitted[it.sig];
        done[i] = 1;
        --left;
        progressed = true;
        break;                                                             // restart from the lowest rank still open
      }
      if (!progressed) return hipErrorInvalidValue;                         // a dependency cycle: cannot happen for a chain
    }
    range_lo = range_hi;
  }
  qstart[8] = (int)items.size();
  if (items.size() != raw.size()) return hipErrorInvalidValue;
  ChainPlanImpl* p = new (std::nothrow) ChainPlanImpl();
  if (!p) return hipErrorOutOfMemory;
  p->n_items = (int)items.size();
  p->n_ops = n_ops;
  p->xcd_local = xcd_local ? 1 : 0;
  p->flops = b->flops; p->bytes = b->bytes;
  p->state_bytes = ((size_t)(CHAIN_STATE_HDR + n_counters) * sizeof(unsigned) + 15) / 16 * 16;
  std::vector<ChainOp> ops(n_ops);
  for (int o = 0; o < n_ops; ++o) ops[o] = b->ops[o].op;
  p->items_host = items;
  p->qstart_host = qstart;
  for (int o = 0; o < n_ops; ++o) p->op_kind_host.push_back(b->ops[o].op.kind);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->ops_dev), ops.size() * sizeof(ChainOp));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->items_dev), items.size() * sizeof(ChainItem));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->qstart_dev), 16 * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->state_dev), p->state_bytes);
  if (e == hipSuccess) e = hipMemcpy(p->ops_dev, ops.data(), ops.size() * sizeof(ChainOp), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->items_dev, items.data(), items.size() * sizeof(ChainItem), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->qstart_dev, qstart.data(), 9 * sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(p->state_dev, 0, p->state_bytes);
  if (e == hipSuccess && dev_env("AVSEP_CHAIN_DBG")) {
    e = hipMalloc(reinterpret_cast<void**>(&p->dbg_dev), items.size() * 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(p->dbg_dev, 0, items.size() * 4 * sizeof(unsigned long long));
  }
  if (const char* g = dev_env("AVSEP_CHAIN_BACKOFF")) p->backoff = atoi(g);                // developer sweep
  int per_cu = 0, dev = 0, cus = 256;
  p->wps = 4;
  if (const char* g = dev_env("AVSEP_CHAIN_WPS")) p->wps = atoi(g) == 3 ? 3 : 4;       // developer sweep
  if (e == hipSuccess) e = chain_occupancy(p, &per_cu);
  if (e == hipSuccess) e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) { chain_plan_free(p); return e; }
  if (per_cu < 1) per_cu = 1;
  if (per_cu > p->wps) per_cu = p->wps;
  if (const char* g = dev_env("AVSEP_CHAIN_WGPC")) per_cu = std::max(1, atoi(g));      // developer sweep
  // (XCD-local: every XCD needs its share of workgroups even when the list is short, so the grid is not cut to the item count)
  p->grid = xcd_local ? cus * per_cu : std::min(p->n_items, cus * per_cu);
  *out = p;
  return hipSuccess;
}

void chain_plan_free(ChainPlanImpl* p) {
  if (!p) return;
  if (p->ops_dev) (void)hipFree(p->ops_dev);
  if (p->items_dev) (void)hipFree(p->items_dev);
  if (p->state_dev) (void)hipFree(p->state_dev);
  if (p->qstart_dev) (void)hipFree(p->qstart_dev);
  if (p->dbg_dev) (void)hipFree(p->dbg_dev);
  delete p;
}

double chain_plan_flops(const ChainPlanImpl* p) { return p->flops; }
double chain_plan_bytes(const ChainPlanImpl* p) { return p->bytes; }
int chain_plan_items(const ChainPlanImpl* p) { return p->n_items; }

hipError_t launch_chain(const ChainPlanImpl* p, hipStream_t s) {
  hipError_t e = hipMemsetAsync(p->state_dev, 0, p->state_bytes, s);
  if (e != hipSuccess) return e;
  ChainArgs a{p->ops_dev, p->items_dev, p->qstart_dev, p->state_dev, p->dbg_dev, p->backoff};
  return chain_dispatch(p, [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)p->grid), dim3(256), 0, s, a);
    return hipGetLastError();
  });
}

// the error word of the last launch (0 = every wait was satisfied); synchronises the stream
hipError_t chain_plan_error(const ChainPlanImpl* p, hipStream_t s, unsigned* word) {
  hipError_t e = hipStreamSynchronize(s);
  if (e != hipSuccess) return e;
  std::vector<unsigned> st(p->state_bytes / sizeof(unsigned));
  e = hipMemcpy(st.data(), p->state_dev, p->state_bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return e;
  *word = st[CHAIN_ERR_WORD];
  for (int q = 0; q < 8 && !*word; ++q)                                   // a queue no workgroup drained (XCD-local: an XCD without one)
    if ((int)st[16 * q] < p->qstart_host[q + 1] - p->qstart_host[q]) {
      *word = 0x80000000u | (unsigned)q;
      fprintf(stderr, "[chain] queue %d was not drained: %u of %d tickets drawn\n", q, st[16 * q], p->qstart_host[q + 1] - p->qstart_host[q]);
    }
  if (p->dbg_dev) {      // developer diagnostics: where a tile's life goes, per op (100 MHz wall clock)
    std::vector<unsigned long long> d((size_t)p->n_items * 4);
    if (hipMemcpy(d.data(), p->dbg_dev, d.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
      unsigned long long t0 = ~0ULL, t1 = 0;
      for (int i = 0; i < p->n_items; ++i) { t0 = std::min(t0, d[4 * (size_t)i]); t1 = std::max(t1, d[4 * (size_t)i + 3]); }
      fprintf(stderr, "[chain dbg] %d items, %d ops, grid %d (%d per CU cap), back-off %d: launch span %.1f us\n", p->n_items, p->n_ops,
              p->grid, p->wps, p->backoff, (double)(t1 - t0) / 100.0);
      fprintf(stderr, "[chain dbg]  op kind tiles | first start .. last done (us) | mean wait  tile  drain+barrier (us) | max wait\n");
      for (int o = 0; o < p->n_ops; ++o) {
        double w = 0, tl = 0, pb = 0, wmax = 0; int n = 0; unsigned long long a0 = ~0ULL, a1 = 0;
        for (int i = 0; i < p->n_items; ++i) {
          if (p->items_host[i].op != o) continue;
          const unsigned long long* q = &d[4 * (size_t)i];
          w += (double)(q[1] - q[0]); tl += (double)(q[2] - q[1]); pb += (double)(q[3] - q[2]); ++n;
          wmax = std::max(wmax, (double)(q[1] - q[0]));
          a0 = std::min(a0, q[0]); a1 = std::max(a1, q[3]);
        }
        if (n) fprintf(stderr, "[chain dbg]  %2d  %d  %5d | %7.1f .. %7.1f | %6.2f %6.2f %6.2f | %6.1f\n", o, p->op_kind_host[o], n,
                       (double)(a0 - t0) / 100.0, (double)(a1 - t0) / 100.0, w / n / 100.0, tl / n / 100.0, pb / n / 100.0, wmax / 100.0);
      }
    }
  }
  if (st[CHAIN_ERR_WORD]) {                 // a wait gave up: say which tile and what its counters read now
    const int t = (int)st[CHAIN_ERR_WORD] - 1;
    if (t >= 0 && t < p->n_items) {
      const ChainItem& it = p->items_host[t];
      fprintf(stderr, "[chain] ticket %d of %d (head now %u, grid %d): op %d kind %d tile (%d, %d) waits for counters [%d, %d) >= %d; they read",
              t, p->n_items, st[0], p->grid, it.op, p->op_kind_host[it.op], it.m0, it.n0, it.dep_lo, it.dep_lo + it.dep_n, it.dep_target);
      for (int k = 0; k < it.dep_n; ++k) fprintf(stderr, " %u", st[CHAIN_STATE_HDR + it.dep_lo + k]);
      fprintf(stderr, "\n");
    }
  }
  return hipSuccess;
}

hipError_t chain_plan_peek(const ChainPlanImpl* p, hipStream_t s, unsigned* out, int n) {
  const size_t bytes = std::min((size_t)n * sizeof(unsigned), p->state_bytes);
  hipError_t e = hipMemcpyAsync(out, p->state_dev, bytes, hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);
}
#endif  // AVSEP_DEV
