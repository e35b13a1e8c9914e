// dtw.hip -- batched dynamic time warping + align post-processing on MI355X (gfx950).
//
// Replaces   fit!(d::DTW, template, sequence) + backward(d)    reference src/dtw.jl:93-145
//            lazy_init!(d, S, T)                               reference src/dtw.jl:44-51
//            transition / observation                          reference src/dtw.jl:23-35
//            align(src, tgt)                                   reference src/align.jl:8-35
//
// This file is compiled with -ffp-contract=off: the parity contract for DTW is BIT-EXACT cost tables and
// back-pointers against the CPU oracle, whose arithmetic is  o = sum_d (v_d - tmpl_d)^2  accumulated
// sequentially in d with separately rounded multiply and add, and  cost = (C[j,t] + o) + transition.
// A GEMM-form distance or a fused multiply-add would change roundings and is therefore not used.
//
// Parallelisation.  Column t+1 of the cost table depends only on column t (src/dtw.jl:110,117), so the S rows
// of a column are independent: one workgroup owns a pair, thread r owns template frame r (its D values live in
// registers), the previous cost column is double-buffered in LDS and one s_barrier separates columns.
// Back-pointers are kept as small step codes (row - predecessor + fstep), 2 bits per cell packed 16 columns
// per LDS word when bstep+fstep <= 3, so the backward pass never touches HBM; the full Float64 / Int64 tables
// of the reference (d.costtable, d.backpointer) are streamed to HBM only when the caller asks for them.
#include "vcmi_common.hpp"
#include "devgroup.hpp"
#include "hostpipe.hpp"

#include <algorithm>
#include <numeric>

namespace vcmi {

struct DtwPair {
  int64_t tmpl_off;     // template (D,S) at feats + tmpl_off
  int64_t seq_off;      // sequence (D,T) at feats + seq_off
  int64_t *path;        // (T) 1-based, may be null when only newtgt is wanted
  double *cost;         // (S,T+1) or null
  int64_t *bp;          // (S,T+1) or null
  double *newtgt;       // (D,S) or null: align() output
  unsigned char *codes; // (S,T) bytes in HBM, used only when the step codes do not fit in LDS
  int64_t obs_off;      // (S,T) observation costs at obs_ws + obs_off (fast path workspace)
  int64_t spad_off;     // offset of the zero-padded sequence copy (fast path, D != DMAX)
  int64_t tpad_off;     // offset of the zero-padded template copy (same workspace)
  int64_t fcodes_off;   // fused path: this pair's packed step codes, [ceil(T/16)][Se] dwords at fcodes_ws + fcodes_off
  int64_t clast_off;    // fused path: last cost column (S doubles) at clast_ws + clast_off
  int64_t path32_off;   // fused path: the pair's 0-based path (T ints) at path32_ws + path32_off
  int32_t S, T;
};

// One workgroup of the fused kernel: rows [row0, row0 + nrows) of a pair ("row strip").  Row r of column t+1 depends
// only on rows r, r-1, r-2 of column t (src/dtw.jl:107-121 with fstep = 0), so a strip needs nothing from the rows above
// it: a template longer than one workgroup's 512 rows is processed bottom strip first, each strip leaving the costs of
// its top two rows (per column) for the next one.
struct DtwStrip {
  int32_t pair, row0, nrows;
  int32_t flag_in, flag_out;     // indices into the flag array (-1: none): wait for / signal the neighbouring strip
  int64_t bnd_in, bnd_out;       // offsets (doubles) into bnd_ws of the (T,2) boundary arrays, -1: none
  // column segment [col0, col0 + ncol) of the pair (a multiple of 16 columns unless it is the last): a strip's columns are
  // cut into segments that run as separate jobs, so that the ~equal jobs of a batch do not come in a whole number per
  // slot (see dtw_run_fused); flag_prev: the job of the same strip's previous segment (-1: first segment)
  int32_t col0, ncol, flag_prev, packed;     // packed: member of a packed group of four one-wave strips
};

__device__ __forceinline__ double dtw_transition(int j, int i) {   // transition(d, j, i), src/dtw.jl:23-31
  return (i == j + 1) ? 0.0 : ((i == j) ? 1.0 : 2.0);
}

// align() post-processing, src/align.jl:20-32, from the 0-based path in LDS (one workgroup per pair)
__device__ void dtw_align_post(const DtwPair &P, const double *__restrict__ seq, int D, const int32_t *path32, int32_t *owner, int32_t *holes) {
  const int S = P.S, T = P.T, tid = threadIdx.x, nthr = blockDim.x;
  if (!P.newtgt) return;
  for (int i = tid; i < S; i += nthr) owner[i] = -1;
  __syncthreads();
  for (int t = tid; t < T; t += nthr) atomicMax(&owner[path32[t]], t);   // newtgt[:,path] = tgt: later frames win
  __syncthreads();
  for (int64_t e = tid; e < (int64_t)S * D; e += nthr) {
    const int i = (int)(e / D), d = (int)(e % D);
    const int k = owner[i];
    P.newtgt[e] = (k >= 0) ? seq[(size_t)D * k + d] : 0.0;
  }
  __shared__ int nholes;
  if (tid == 0) {                                   // hole = setdiff(path[1]:path[end], path), increasing order
    int n = 0;
    for (int i = path32[0]; i <= path32[T - 1]; ++i)
      if (owner[i] < 0 && i > 0 && i < S - 1) holes[n++] = i;   // 1 < i < S in 1-based terms
    nholes = n;
  }
  __syncthreads();
  // each thread owns feature rows d, d+nthr, ...; a row's holes are filled in increasing i, as the reference does
  for (int d = tid; d < D; d += nthr)
    for (int h = 0; h < nholes; ++h) {
      const int i = holes[h];
      P.newtgt[(size_t)D * i + d] = (P.newtgt[(size_t)D * (i - 1) + d] + P.newtgt[(size_t)D * (i + 1) + d]) / 2.0;
    }
}

// shared epilogue: argmin of the last column, backward pass, path output, align post-processing.
// PACKED: the codes are BITS-bit fields of 32-bit words [t / CPW][row] wherever they live (LDS, or global memory when
// LDSCODES is false); otherwise global codes are one byte per cell at P.codes.
template <bool LDSCODES, int BITS, bool PACKED = false>
__device__ void dtw_finish(const DtwPair &P, const double *__restrict__ seq, int D, int fstep, const double *clast, int32_t *path32, int32_t *owner,
                           int32_t *holes, const uint32_t *codes_lds, int Smax) {
  constexpr int CPW = 32 / BITS;
  const int S = P.S, T = P.T, tid = threadIdx.x, nthr = blockDim.x;
  // indmin of the last column (first minimum wins, src/dtw.jl:137): per-thread scan of a strided slice, wave
  // reduction on (value, index) with shuffles, then thread 0 combines the per-wave winners through LDS
  {
    double bv = INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < S; i += nthr) {
      const double c = clast[i];
      if (c < bv) { bv = c; bi = i; }                 // strided, increasing i: keeps the first minimum of the slice
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
      const double ov = __shfl_xor(bv, sh);
      const int oi = __shfl_xor(bi, sh);
      if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    __shared__ double wv_[16];
    __shared__ int wi_[16];
    if ((tid & 63) == 0) { wv_[tid >> 6] = bv; wi_[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      const int nw = (nthr + 63) >> 6;
      for (int w = 1; w < nw; ++w)
        if (wv_[w] < bv || (wv_[w] == bv && wi_[w] < bi)) { bv = wv_[w]; bi = wi_[w]; }
      int row = bi;
      path32[T - 1] = row;
      for (int t = T - 1; t >= 1; --t) {              // src/dtw.jl:140-142; `row` stays in a register
        int code;
        if (LDSCODES || PACKED) code = (codes_lds[(size_t)(t / CPW) * Smax + row] >> (BITS * (t % CPW))) & ((1u << BITS) - 1u);
        else code = P.codes[(size_t)S * t + row];
        row -= code - fstep;
        path32[t - 1] = row;
      }
    }
  }
  __syncthreads();
  if (P.path)
    for (int t = tid; t < T; t += nthr) P.path[t] = (int64_t)path32[t] + 1;
  dtw_align_post(P, seq, D, path32, owner, holes);
}

// ------------------------------------------------------------------------------------------------
// Fast path = two kernels.
//
// (1) dtw_obs_kernel<DMAX>: the observation costs O[i,t] = sum_d (seq[d,t] - tmpl[d,i])^2 of every cell, which do
//     not depend on the recurrence, are computed with NO barrier: lane = template frame (its D values in VGPRs),
//     the sequence column is wave-uniform and arrives through scalar loads (`feats` is a read-only, non-aliased
//     kernel argument), i.e. as SGPR operands of the FP64 ALU -- no LDS traffic, no per-lane loads.  (An LDS-staged
//     broadcast variant measured 2x slower than per-lane loads: the LDS pipe, not the FP64 ALU, became the limiter.)
//     Arithmetic order: sequential in d, unfused sub / mul / add (bit-exact contract).  O goes to an HBM workspace
//     in the cost table's (S,T) column-major order, so writes and the later reads are coalesced.
// (2) dtw_rec_kernel<STEPS,CODES>: one workgroup per pair, thread r = template frame r: per column a handful of
//     instructions (candidate costs, strict '<' selection in the reference's scan order), previous column
//     double-buffered in LDS, one barrier per column, O prefetched PF columns ahead in registers.
//     STEPS: 1 = (bstep 1, fstep 0), 2 = (bstep 2, fstep 0) -- candidate window unrolled; 0 = run-time window.
//     CODES: 0 = 2-bit step codes in LDS, 1 = byte codes in LDS, 2 = byte codes in HBM.
// ------------------------------------------------------------------------------------------------
static constexpr int kObsRows = 256;    // template frames per obs workgroup (4 waves)
static constexpr int kObsCols = 128;    // sequence frames per obs workgroup

// zero-pads the sequence and the template of every pair to DMAX doubles per frame (only launched when D != DMAX), so
// that the observation loop is branch-free: (0 - 0)^2 = +0.0 added to a non-negative sum leaves it bit-identical.
__global__ void __launch_bounds__(256)
dtw_pad_kernel(const double *__restrict__ feats, const DtwPair *__restrict__ pairs, int D, int DMAX, double *__restrict__ spad) {
  const DtwPair P = pairs[blockIdx.x];
  const int64_t ns = (int64_t)P.T * DMAX, nt = (int64_t)P.S * DMAX;
  for (int64_t e = (int64_t)blockIdx.y * 256 + threadIdx.x; e < ns + nt; e += (int64_t)gridDim.y * 256) {
    const bool isseq = e < ns;
    const int64_t k = isseq ? e : e - ns;
    const int64_t f = k / DMAX;
    const int d = (int)(k % DMAX);
    const double v = (d < D) ? feats[(isseq ? P.seq_off : P.tmpl_off) + f * D + d] : 0.0;
    spad[(isseq ? P.spad_off : P.tpad_off) + k] = v;
  }
}

// Observation kernels.  `base` + P.<off>: feature matrices with exactly DMAX doubles per frame (the caller's buffer
// when D == DMAX, else the zero-padded copies made by dtw_pad_kernel).
//
// Compiler-scheduled variant (DMAX > 40): RPL template frames per lane (a wave owns 64*RPL consecutive frames), so
// that every scalar-loaded sequence value feeds RPL subtractions.
template <int DMAX, int RPL>
__global__ void __launch_bounds__(kObsRows)
dtw_obs_kernel(const double *__restrict__ base, int padded, const DtwPair *__restrict__ pairs, double *__restrict__ obs_ws) {
  const DtwPair P = pairs[blockIdx.x];
  const int S = P.S, T = P.T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rbase = (blockIdx.y * (kObsRows / 64) + wave) * (64 * RPL);
  const int t0 = blockIdx.z * kObsCols;
  if (rbase >= S || t0 >= T) return;                          // wave-uniform (no barriers in this kernel)
  const double *__restrict__ seq = base + (padded ? P.spad_off : P.seq_off);
  const double *__restrict__ tmpl = base + (padded ? P.tpad_off : P.tmpl_off);
  double tm[RPL][DMAX];
  bool active[RPL];
#pragma unroll
  for (int q = 0; q < RPL; ++q) {
    const int r = rbase + 64 * q + lane;
    active[q] = r < S;
    const double *row = tmpl + (int64_t)DMAX * (active[q] ? r : S - 1);
#pragma unroll
    for (int d = 0; d < DMAX; ++d) tm[q][d] = row[d];
  }
  const int t1 = (t0 + kObsCols < T) ? t0 + kObsCols : T;
  double *__restrict__ O = obs_ws + P.obs_off + rbase + lane;   // kernel-argument base: global (not flat) stores
  for (int t = t0; t < t1; ++t) {
    const double *__restrict__ v = seq + (size_t)DMAX * t;     // wave-uniform -> scalar loads
    double o[RPL];                                              // observation(d, v, i), src/dtw.jl:33-35
#pragma unroll
    for (int q = 0; q < RPL; ++q) o[q] = 0.0;
#pragma unroll
    for (int d = 0; d < DMAX; ++d)
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const double df = v[d] - tm[q][d];
        const double sq = df * df;
        o[q] = o[q] + sq;
      }
#pragma unroll
    for (int q = 0; q < RPL; ++q)
      if (active[q]) O[(size_t)S * t + 64 * q] = o[q];
  }
}

// Hand-scheduled column loop (tools/gen_dtw_obs_asm.py -> dtw_obs_asm.inc), DMAX <= 40: lane = template frame (its
// DMAX values in v[0:2*DMAX-1]), the sequence column arrives through scalar loads in half-column chunks with two SGPR
// buffers -- wait(0) -> issue the next chunk's loads -> FP64 work of the current chunk -- see the generator for why the
// compiler cannot produce this schedule.  91 VGPRs -> 5 waves per SIMD.
#include "dtw_obs_asm.inc"
template <int DMAX>
__global__ void __launch_bounds__(kObsRows)
dtw_obs_asm_kernel(const double *__restrict__ base, int padded, const DtwPair *__restrict__ pairs,
                   double *__restrict__ obs_ws) {
  const DtwPair P = pairs[blockIdx.x];
  const int S = P.S, T = P.T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rbase = (blockIdx.y * (kObsRows / 64) + wave) * 64;
  const int t0 = blockIdx.z * kObsCols;
  if (rbase >= S || t0 >= T) return;                        // wave-uniform (no barriers in this kernel)
  const int t1 = (t0 + kObsCols < T) ? t0 + kObsCols : T;
  const int r0 = rbase + lane;
  // rows past the template are clamped for the loads and masked for the stores
  const double *row0 = base + (padded ? P.tpad_off : P.tmpl_off) + (int64_t)DMAX * (r0 < S ? r0 : S - 1);
  const double *seq = base + (padded ? P.spad_off : P.seq_off) + (int64_t)DMAX * t0;   // wave-uniform
  double *optr = obs_ws + P.obs_off + (int64_t)S * t0 + r0;
  const uint64_t m0 = __ballot(r0 < S);
  const uint64_t stride = (uint64_t)S * 8;
  uint32_t ncols = (uint32_t)(t1 - t0), off = 0;
  // Warm this XCD's L2 with the workgroup's block of sequence columns (one vector load per 64-byte line, all in flight
  // at once): a scalar load has one chunk of lead time, which covers an L2 hit but not an HBM / Infinity-Cache miss.
  double pf = 0.0;
  const int nlines = (int)ncols * DMAX / 8;
  for (int l = threadIdx.x; l < nlines; l += kObsRows) pf += seq[8 * l];
#define VCMI_OBS_ASM(BODY)                                                                                          \
  asm volatile(BODY                                                                                                 \
               : [seq] "+s"(seq), [n] "+s"(ncols), [optr] "+v"(optr), [off] "+s"(off)                               \
               : [row0] "v"(row0), [stride] "s"(stride), [m0] "s"(m0)                                               \
               : "memory", "vcc", "scc", VCMI_OBS_ASM_CLOBBERS)
  if constexpr (DMAX == 8) VCMI_OBS_ASM(VCMI_DTW_OBS_ASM_D8);
  if constexpr (DMAX == 16) VCMI_OBS_ASM(VCMI_DTW_OBS_ASM_D16);
  if constexpr (DMAX == 24) VCMI_OBS_ASM(VCMI_DTW_OBS_ASM_D24);
  if constexpr (DMAX == 32) VCMI_OBS_ASM(VCMI_DTW_OBS_ASM_D32);
  if constexpr (DMAX == 40) VCMI_OBS_ASM(VCMI_DTW_OBS_ASM_D40);
#undef VCMI_OBS_ASM
  if (pf == 0x1.123456789abcdp+1000) obs_ws[0] = pf;   // keeps the warm-up loads alive; never true for finite features
}

template <int STEPS, int CODES>
__global__ void __launch_bounds__(1024)
dtw_rec_kernel(const double *__restrict__ feats, const DtwPair *__restrict__ pairs, int D, int fstep, int bstep, int Smax,
               int Tmax, const double *__restrict__ obs_ws) {
  constexpr bool LDSCODES = (CODES != 2);
  constexpr int BITS = (CODES == 0) ? 2 : 8;
  constexpr int CPW = 32 / BITS;
  constexpr int PF = 8;                                   // columns of O per register set
  // This kernel is a chain of ~T short dependent steps; observation kernels of the next chunk run beside it on the same
  // CUs (dtw_run) and must not delay its instruction issue.
  __builtin_amdgcn_s_setprio(3);
  const DtwPair P = pairs[blockIdx.x];
  const int S = P.S, T = P.T;
  const int r = threadIdx.x;
  const bool active = r < S;
  extern __shared__ unsigned char smem_raw[];
  double *cbuf = reinterpret_cast<double *>(smem_raw);            // [2][Smax]
  int32_t *path32 = reinterpret_cast<int32_t *>(cbuf + 2 * Smax); // [Tmax]
  int32_t *owner = path32 + Tmax;                                 // [Smax]
  int32_t *holes = owner + Smax;                                  // [Smax]
  uint32_t *codes = reinterpret_cast<uint32_t *>(holes + Smax);   // [ceil(Tmax/CPW)][Smax] when LDSCODES
  if (T == 0) return;
  // kernel-argument base -> global_load with counted vmcnt (a pointer loaded from the descriptor would be a FLAT
  // load, which must be waited for with vmcnt(0) and would serialise the prefetch)
  const double *__restrict__ O = obs_ws + P.obs_off;

  // lazy_init!: costtable[:,1] = 1:S, backpointer[:,1] = 1:S  (src/dtw.jl:49-50)
  double cprev = (double)(r + 1);
  if (active) {
    cbuf[r] = cprev;
    if (P.cost) P.cost[r] = cprev;
    if (P.bp) P.bp[r] = r + 1;
  }
  // O is read PF columns at a time into one of two register sets (ping-pong, unconditional clamped loads issued as
  // a batch) while the other set is consumed, so HBM latency is covered by PF columns of recurrence work.
  const int rl = active ? r : 0;
  auto load_set = [&](double (&dst)[PF], int tbase) {
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      int tc = tbase + k;
      tc = tc < T ? tc : T - 1;
      dst[k] = O[(size_t)S * tc + rl];
    }
  };
  uint32_t word = 0;
  auto column = [&](int t, double o) {
    const double *cp = cbuf + (t & 1) * Smax;
    double *cn = cbuf + ((t + 1) & 1) * Smax;
    if (active) {
      int arg = r;                                    // minindex = i, src/dtw.jl:106-110
      double best = (cprev + o) + 1.0;
      if (STEPS == 0) {
        for (int j = r - bstep; j <= r + fstep; ++j) {  // src/dtw.jl:113-121
          if (j < 0 || j >= S) continue;
          const double c = (cp[j] + o) + dtw_transition(j, r);
          if (c < best) { best = c; arg = j; }
        }
      } else {
        // window j = r-STEPS .. r, increasing j; j == r repeats the start value and never wins the strict '<'
        if (STEPS == 2) {
          const double c2 = (cp[r >= 2 ? r - 2 : r] + o) + 2.0;
          if (r >= 2 && c2 < best) { best = c2; arg = r - 2; }
        }
        const double c1 = (cp[r >= 1 ? r - 1 : r] + o) + 0.0;
        if (r >= 1 && c1 < best) { best = c1; arg = r - 1; }
      }
      cn[r] = best;
      cprev = best;
      if (P.cost) P.cost[(size_t)S * (t + 1) + r] = best;
      if (P.bp) P.bp[(size_t)S * (t + 1) + r] = arg + 1;
      const uint32_t code = (uint32_t)(r - arg + fstep);
      if (LDSCODES) {
        word |= code << (BITS * (t % CPW));
        if ((t % CPW) == CPW - 1 || t == T - 1) {
          codes[(t / CPW) * Smax + r] = word;
          word = 0;
        }
      } else {
        P.codes[(size_t)S * t + r] = (unsigned char)code;
      }
    }
    // column barrier that waits for the LDS traffic only: __syncthreads() would also drain vmcnt, i.e. stall every
    // column on the O prefetches that are meant to stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  double oa[PF], ob[PF];
  load_set(oa, 0);
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += 2 * PF) {
    load_set(ob, t0 + PF);
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (t0 + k < T) column(t0 + k, oa[k]);              // wave-uniform guard
    load_set(oa, t0 + 2 * PF);
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (t0 + PF + k < T) column(t0 + PF + k, ob[k]);
  }
  if (!LDSCODES) {      // the codes in HBM were written by this workgroup: workgroup scope (an agent-scope fence is L2-wide)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  dtw_finish<LDSCODES, BITS>(P, feats + P.seq_off, D, fstep, cbuf + (T & 1) * Smax, path32, owner, holes, codes, Smax);
}

// ------------------------------------------------------------------------------------------------
// Fused path (fstep = 0, bstep in {1,2}, D <= 40, path-only): ONE forward kernel computes observation costs and the
// recurrence together -- O never exists in memory -- and streams the 2-bit step codes to HBM (62.5 KB per 500x500 pair,
// against 2 MB for O); a short second kernel per pair loads the pair's codes into LDS and runs backward + align.
//
// dtw_fused_kernel<DMAX,STEPS>: workgroup = one row strip (<= 512 rows) of a pair = up to 4 waves; a wave owns 128
// consecutive rows, lane l the rows 2l and 2l+1 (their features stay in 160 VGPRs); the sequence column is the wave-
// uniform SGPR operand.  The column loop is hand-scheduled (tools/gen_dtw_fused_asm.py -> dtw_fused_asm.inc, where the
// schedule, the register map and the arithmetic order are documented).  No workgroup barrier inside the loop: the only
// coupling between waves is "lane 0 needs the previous wave's top two rows of the previous column", handed over through
// a sequence-tagged ring in LDS.  LDS use is ~9 KB + 16 B per column, so occupancy is set by the registers (2 waves per
// SIMD = two workgroups per CU).
// ------------------------------------------------------------------------------------------------
#include "dtw_fused_asm.inc"
static constexpr int kFusedRows = 512;        // rows per strip (4 waves x 64 lanes x 2 rows)
static constexpr int kFusedThreads = 256;

// wave-uniform value held in VGPRs -> SGPRs (inline-asm "s" operands)
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}

#ifdef VCMI_DTW_PROF
__device__ unsigned long long dtw_prof[8];      // cycles summed over jobs: [0] prologue, [1] column loop (incl. the template loads), [2] epilogue, [3] jobs, [4] flag waits
#define DTW_PROF_MARK(var) const long long var = (long long)__builtin_readcyclecounter()
#else
#define DTW_PROF_MARK(var)
#endif
// Synchronisation between jobs of the fused kernel (strip below -> strip above, column segment -> next segment).
// The L2 caches of the eight XCDs are not coherent with each other, so an acquire / release at agent scope is an L2-WIDE
// invalidate / write-back (buffer_inv sc1 / buffer_wbl2 sc1) -- which every other workgroup of the XCD pays for: with one
// such pair per job a 100-column job took 325k cycles instead of 240k (cycle counter, VCMI_DTW_PROF).  Instead the few
// values that cross jobs -- boundary pairs, last cost columns, flags -- are themselves accessed at agent scope (sc1: stores
// write through to the device-wide coherence point, loads do not use a possibly stale cached copy), relaxed, and ordered by
// hand: the producer waits until its stores are acknowledged (vmcnt(0)), then stores the flag; the consumer sees the flag,
// then loads.  Everything else a job writes (step codes) is read by a later kernel.
__device__ __forceinline__ void dtw_store_shared(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double dtw_load_shared(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dtw_wait_flag(const int *flag, int epoch) {
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(32);
}
// all of this wave's stores have been acknowledged
__device__ __forceinline__ void dtw_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int DMAX, int STEPS>
__device__ __forceinline__ void dtw_fused_job(const int sfirst, const double *__restrict__ base, int padded,
                                              const DtwPair *__restrict__ pairs, const DtwStrip *__restrict__ strips,
                                              uint32_t *__restrict__ fcodes_ws, double *__restrict__ bnd_ws,
                                              double *__restrict__ clast_ws, int *__restrict__ flags, int epoch) {
  DTW_PROF_MARK(pt0);
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  unsigned char *outbox = fsm;                                                        // [4][VCMI_FUSED_OUTBOX]
  double *clast = reinterpret_cast<double *>(fsm + 4 * VCMI_FUSED_OUTBOX);            // [kFusedRows]
  double *pinit = clast + kFusedRows;                                                 // [kFusedRows] (DMAX > 40: initial neighbour costs)
  double *bnd = pinit + kFusedRows;                                                   // [1 + T][2]
  // A job is one strip segment (first descriptor `sfirst`), or a PACKED group: each of the four waves runs its own
  // one-wave strip (the short bottom strips of long templates, <= 128 rows), four unrelated strips per workgroup, so that
  // no wave slot idles beside them.
  const int tid = threadIdx.x, lane = tid & 63;
  const bool packed = strips[sfirst].packed != 0;
  const int slotw = tid >> 6;                                  // wave slot in the workgroup (outbox, clast area)
  const DtwStrip st = packed ? strips[sfirst + slotw] : strips[sfirst];
  const DtwPair P = pairs[st.pair];
  const int S = P.S, T = st.ncol, col0 = st.col0;              // T: the columns of THIS job
  if (!packed && T == 0) return;
  const int wave = packed ? 0 : slotw;                         // wave index within the strip
  const int nw = (st.nrows + 127) >> 7;
  const int lr0 = 128 * wave + 2 * lane;            // first row of the lane, local to the strip
  const int gr0 = st.row0 + lr0;                    // ... and in the pair
  // lazy_init!: costtable[:,1] = 1:S (src/dtw.jl:49); the neighbours of column 1 are rows gr0-2, gr0-1
  double c0 = (double)(gr0 + 1), c1 = (double)(gr0 + 2);
  double p0 = gr0 >= 2 ? (double)(gr0 - 1) : INFINITY, p1 = gr0 >= 1 ? (double)gr0 : INFINITY;
  if (__syncthreads_or(st.flag_prev >= 0)) {   // (a packed group: any of its four strips)
    // a later column segment: the previous one (an earlier job: running or finished) left the costs of its last column,
    // for every row of the strip, in clast_ws; rows below the strip belong to the strip below and reach lane 0 of wave 0
    // through the boundary array, like every column's
    if (lane == 0 && (packed || tid == 0) && st.flag_prev >= 0)
      dtw_wait_flag(&flags[st.flag_prev], epoch);
    __syncthreads();
  }
  DTW_PROF_MARK(ptw);
  if (st.flag_prev >= 0) {
    const double *cl = clast_ws + P.clast_off;
    const int last = S - 1;
    c0 = dtw_load_shared(cl + (gr0 < S ? gr0 : last));
    c1 = dtw_load_shared(cl + (gr0 + 1 < S ? gr0 + 1 : last));
    p0 = (gr0 - 2 >= st.row0) ? dtw_load_shared(cl + (gr0 - 2 < S ? gr0 - 2 : last)) : INFINITY;
    p1 = (gr0 - 1 >= st.row0) ? dtw_load_shared(cl + (gr0 - 1 < S ? gr0 - 1 : last)) : INFINITY;
  }
  if (tid < 4) *reinterpret_cast<uint32_t *>(outbox + tid * VCMI_FUSED_OUTBOX + VCMI_FUSED_RING * 16) = 0u;   // tags
  if (lane == 63) {   // what the next wave's lane 0 reads for column 0: the ring's last slot holds the initial costs
    double *slot = reinterpret_cast<double *>(outbox + slotw * VCMI_FUSED_OUTBOX + (VCMI_FUSED_RING - 1) * 16);
    slot[0] = c0;
    slot[1] = c1;
  }
  if (!packed && st.flag_in >= 0) {
    // the strip below must be complete (it precedes this workgroup in the grid, so it is resident or finished)
    if (tid == 0)
      dtw_wait_flag(&flags[st.flag_in], epoch);
    __syncthreads();
    // entry 0: initial costs of the two rows below the strip; entry 1+t: their costs after column t
    // (of this job's columns: LDS entry e <-> the pair's entry col0 + e)
    const double *src = bnd_ws + st.bnd_in + 2 * (int64_t)col0;
    if (tid == 0) {
      bnd[0] = col0 == 0 ? (double)(st.row0 - 1) : dtw_load_shared(src - 2);
      bnd[1] = col0 == 0 ? (double)st.row0 : dtw_load_shared(src - 1);
    }
    for (int i = tid; i < 2 * T; i += (int)blockDim.x) bnd[2 + i] = dtw_load_shared(src + i);
  }
  __syncthreads();
  DTW_PROF_MARK(pt1);
  if (wave < nw && T > 0) {
    // (the 40 loads of a lane's two rows touch 64 cache lines each; measured, they cost nothing: a packed, fully coalesced
    // copy of the templates left the job time unchanged and its packing kernel took 85 us)
    const double *tmpl = base + (padded ? P.tpad_off : P.tmpl_off);
    const double *rowA = tmpl + (int64_t)DMAX * (gr0 < S ? gr0 : S - 1);
    const double *rowB = tmpl + (int64_t)DMAX * (gr0 + 1 < S ? gr0 + 1 : S - 1);
    const double *seq = base + (padded ? P.spad_off : P.seq_off) + (int64_t)DMAX * col0;
    const int Se = (S + 1) & ~1;
    uint32_t *codes = fcodes_ws + P.fcodes_off + gr0 + (int64_t)(col0 >> 4) * Se;
    const uint64_t cstride = (uint64_t)Se * 4;
    int cnt = (st.nrows - 128 * wave + 1) >> 1;
    cnt = cnt > 64 ? 64 : cnt;
    uint32_t out = lds_addr(outbox + slotw * VCMI_FUSED_OUTBOX), bndl = lds_addr(bnd);
    uint32_t mode = (wave > 0 ? 1u : 0u) | ((wave == 0 && !packed && st.flag_in >= 0) ? 2u : 0u) | (wave + 1 < nw ? 4u : 0u);
    int explane = -1;
    double *gout = bnd_ws;
    if (st.bnd_out >= 0 && wave == nw - 1) {          // non-final strips have an even number of rows
      explane = (st.nrows >> 1) - 1 - 64 * wave;
      gout = bnd_ws + st.bnd_out + 2 * (int64_t)col0;
    }
    uint32_t clast_a = lds_addr(clast + 128 * slotw + 2 * lane);
    if constexpr (DMAX > 40) {     // the wide kernels read their initial costs from LDS (operand VGPRs are scarce there)
      clast[128 * slotw + 2 * lane] = c0;
      clast[128 * slotw + 2 * lane + 1] = c1;
      pinit[128 * slotw + 2 * lane] = p0;
      pinit[128 * slotw + 2 * lane + 1] = p1;
    }
    uint32_t ncols = (uint32_t)T, off = 0, tcol = 0;
    // every operand the loop reads is wave-uniform or per-lane as declared; force the uniform ones into SGPRs
    cnt = __builtin_amdgcn_readfirstlane(cnt);
    out = __builtin_amdgcn_readfirstlane(out);
    bndl = __builtin_amdgcn_readfirstlane(bndl);
    mode = __builtin_amdgcn_readfirstlane(mode);
    explane = __builtin_amdgcn_readfirstlane(explane);
    ncols = __builtin_amdgcn_readfirstlane(ncols);
    seq = reinterpret_cast<const double *>(uniform64(reinterpret_cast<uint64_t>(seq)));
    const uint64_t cstride_u = uniform64(cstride);
#define VCMI_FUSED_ASM(BODY) VCMI_FUSED_ASM_C(BODY, VCMI_FUSED_ASM_CLOBBERS)
#define VCMI_FUSED_ASM_C(BODY, CLOBBERS)                                                                              \
  asm volatile(BODY                                                                                                   \
               : [seq] "+s"(seq), [off] "+s"(off), [n] "+s"(ncols), [t] "+s"(tcol), [codes] "+v"(codes), [gout] "+v"(gout) \
               : [cstride] "s"(cstride_u), [cnt] "s"(cnt), [out] "s"(out), [bnd] "s"(bndl), [mode] "s"(mode),          \
                 [explane] "s"(explane), [rowA] "v"(rowA), [rowB] "v"(rowB), [c0] "v"(c0), [c1] "v"(c1), [p0] "v"(p0), \
                 [p1] "v"(p1), [clast] "v"(clast_a)                                                                   \
               : "memory", "vcc", "scc", CLOBBERS)
    // the wide kernels (40 < D <= 48) read the initial costs from LDS: (C0, C1) at %[clast], (P0, P1) kFusedRows doubles behind
#define VCMI_FUSED3_ASM(BODY, CLOBBERS)                                                                               \
  asm volatile(BODY                                                                                                   \
               : [seq] "+s"(seq), [off] "+s"(off), [n] "+s"(ncols), [t] "+s"(tcol), [codes] "+v"(codes), [gout] "+v"(gout) \
               : [cstride] "s"(cstride_u), [cnt] "s"(cnt), [out] "s"(out), [bnd] "s"(bndl), [mode] "s"(mode),          \
                 [explane] "s"(explane), [rowA] "v"(rowA), [rowB] "v"(rowB), [clast] "v"(clast_a)                      \
               : "memory", "vcc", "scc", CLOBBERS)
    if constexpr (STEPS == 1) {
      if constexpr (DMAX == 8) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D8_S1);
      if constexpr (DMAX == 16) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D16_S1);
      if constexpr (DMAX == 24) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D24_S1);
      if constexpr (DMAX == 32) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D32_S1);
      if constexpr (DMAX == 40) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D40_S1);
      if constexpr (DMAX == 41) VCMI_FUSED3_ASM(VCMI_DTW_FUSED_ASM_D41_S1, VCMI_FUSED_ASM_CLOBBERS_D41);
      if constexpr (DMAX == 48) VCMI_FUSED3_ASM(VCMI_DTW_FUSED_ASM_D48_S1, VCMI_FUSED_ASM_CLOBBERS_D48);
    } else {
      if constexpr (DMAX == 41) VCMI_FUSED3_ASM(VCMI_DTW_FUSED_ASM_D41_S2, VCMI_FUSED_ASM_CLOBBERS_D41);
      if constexpr (DMAX == 48) VCMI_FUSED3_ASM(VCMI_DTW_FUSED_ASM_D48_S2, VCMI_FUSED_ASM_CLOBBERS_D48);
      if constexpr (DMAX == 8) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D8_S2);
      if constexpr (DMAX == 16) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D16_S2);
      if constexpr (DMAX == 24) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D24_S2);
      if constexpr (DMAX == 32) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D32_S2);
#if !defined(VCMI_FUSED_VARIANT)
      if constexpr (DMAX == 40) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D40_S2);
#elif VCMI_FUSED_VARIANT == 1   // timing experiments only (wrong results): see tools/gen_dtw_fused_asm.py
      if constexpr (DMAX == 40) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D40_S2_V1);
#elif VCMI_FUSED_VARIANT == 2
      if constexpr (DMAX == 40) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D40_S2_V2);
#else
      if constexpr (DMAX == 40) VCMI_FUSED_ASM(VCMI_DTW_FUSED_ASM_D40_S2_V3);
#endif
    }
#undef VCMI_FUSED_ASM
#undef VCMI_FUSED_ASM_C
#undef VCMI_FUSED3_ASM
  }
  __syncthreads();
  DTW_PROF_MARK(pt2);
#ifdef VCMI_DTW_PROF
  auto prof_end = [&]() {
    const long long pt3 = (long long)__builtin_readcyclecounter();
    if (tid == 0) {
      atomicAdd(&dtw_prof[0], (unsigned long long)(pt1 - pt0));
      atomicAdd(&dtw_prof[1], (unsigned long long)(pt2 - pt1));
      atomicAdd(&dtw_prof[2], (unsigned long long)(pt3 - pt2));
      atomicAdd(&dtw_prof[3], 1ull);
      atomicAdd(&dtw_prof[4], (unsigned long long)(ptw - pt0));
    }
  };
#endif
  if (packed) {      // every wave finishes its own strip
    if (T > 0) {
      for (int i = lane; i < st.nrows; i += 64) dtw_store_shared(&clast_ws[P.clast_off + st.row0 + i], clast[128 * slotw + i]);
      if (st.flag_out >= 0) {
        dtw_stores_done();      // (also the boundary pairs the column loop exported)
        if (lane == 0) __hip_atomic_store(&flags[st.flag_out], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#ifdef VCMI_DTW_PROF
    prof_end();
#endif
    return;
  }
  for (int i = tid; i < st.nrows; i += (int)blockDim.x) dtw_store_shared(&clast_ws[P.clast_off + st.row0 + i], clast[i]);
  if (st.flag_out >= 0) {
    dtw_stores_done();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&flags[st.flag_out], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifdef VCMI_DTW_PROF
  prof_end();
#endif
}

// One job per workgroup, in grid order.
template <int DMAX, int STEPS>
__global__ void __launch_bounds__(kFusedThreads)
dtw_fused_kernel(const double *__restrict__ base, int padded, const DtwPair *__restrict__ pairs,
                 const DtwStrip *__restrict__ strips, uint32_t *__restrict__ fcodes_ws, double *__restrict__ bnd_ws,
                 double *__restrict__ clast_ws, int *__restrict__ flags, int epoch, const int *__restrict__ job_first) {
  dtw_fused_job<DMAX, STEPS>(job_first[blockIdx.x], base, padded, pairs, strips, fcodes_ws, bnd_ws, clast_ws, flags, epoch);
}

// Persistent form: as many workgroups as the device holds at once, each drawing the next job from a ticket counter until
// the list is exhausted.  A workgroup that waits on a flag waits for a job with a LOWER ticket (the list is in dependency
// order), i.e. for a job some running workgroup already holds: no deadlock, whatever the residency.  With the columns cut
// into segments the jobs are short; drawing them inside the kernel removes the gap the hardware dispatcher leaves
// between a finished workgroup and its successor.
template <int DMAX, int STEPS>
__global__ void __launch_bounds__(kFusedThreads, 2)      // two workgroups per CU: 256 registers
dtw_fused_persistent_kernel(const double *__restrict__ base, int padded, const DtwPair *__restrict__ pairs,
                            const DtwStrip *__restrict__ strips, uint32_t *__restrict__ fcodes_ws, double *__restrict__ bnd_ws,
                            double *__restrict__ clast_ws, int *__restrict__ flags, int epoch, const int *__restrict__ job_first,
                            int njobs, int *__restrict__ ticket) {
  __shared__ int job_s;
  for (;;) {
    if (threadIdx.x == 0) job_s = atomicAdd(ticket, 1);
    __syncthreads();
    const int job = __builtin_amdgcn_readfirstlane(job_s);     // uniform: the job's descriptors stay in SGPRs
    __syncthreads();                // (job_s is rewritten by the next draw)
    if (job >= njobs) break;
    dtw_fused_job<DMAX, STEPS>(job_first[job], base, padded, pairs, strips, fcodes_ws, bnd_ws, clast_ws, flags, epoch);
    __syncthreads();                // the next job reuses the outboxes, the boundary array and the last-column area
  }
}

// Backward pass of the fused path: ONE WAVE per pair, no LDS.  (Round 2's epilogue copied the pair's 64 KB of packed step
// codes into LDS and let one thread chase T dependent LDS reads: 110 us for 1000 pairs, two workgroups per CU.)  The path
// moves down by at most two rows per column (bstep <= 2, fstep = 0), so over the 16 columns of one code-word row it stays
// within 33 rows: a wave keeps a WINDOW of 192 rows of that word row in three registers per lane (coalesced loads), the
// walk itself is scalar -- v_readlane with the uniform row, shift, mask, subtract -- and the windows of the next three word
// rows are in flight meanwhile (a window anchored at the current row r covers [r - 191, r]; the word row three blocks
// ahead is entered at r - 64 or above and left at r - 96 or above).  Path values are collected one per lane, parked in LDS
// per 64 columns (no global store inside the walk: loads and stores share vmcnt and complete out of order with respect
// to each other, so with a store pending the compiler waits for vmcnt(0) before every block and the look-ahead is lost)
// and leave at the end as coalesced stores: 0-based to path32_ws (for the align kernel), 1-based Int64 to the caller's path.
__device__ __forceinline__ void dtw_window_load(const uint32_t *__restrict__ gc, int Se, int tb, int anchor, int lane, uint32_t (&w)[3]) {
  // (unconditional -- beyond the first word row it re-reads row 0: the compiler can then count the loads in flight, with
  // a conditional load it waits for vmcnt(0) before every block and the look-ahead is lost)
  const uint32_t *rowp = gc + (size_t)(tb < 0 ? 0 : tb) * Se;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int r = anchor - 191 + 64 * k + lane;
    r = r < 0 ? 0 : (r >= Se ? Se - 1 : r);            // (rows the walk cannot reach)
    w[k] = rowp[r];
  }
}

__global__ void __launch_bounds__(64)
dtw_fused_backward_kernel(const DtwPair *__restrict__ pairs, const uint32_t *__restrict__ fcodes_ws,
                          const double *__restrict__ clast_ws, int32_t *__restrict__ path32_ws) {
  extern __shared__ int32_t path_lds[];                 // [Tmax]
  const int lane = threadIdx.x;
  const DtwPair P = pairs[blockIdx.x];
  const int S = P.S, T = P.T;
  if (T == 0) return;
  // indmin of the last column, first minimum wins (src/dtw.jl:137)
  int row;
  {
    const double *cl = clast_ws + P.clast_off;
    double bv = INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < S; i += 64) {
      const double c = cl[i];
      if (c < bv) { bv = c; bi = i; }
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
      const double ov = __shfl_xor(bv, sh);
      const int oi = __shfl_xor(bi, sh);
      if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    row = __builtin_amdgcn_readfirstlane(bi);
  }
  const int Se = (S + 1) & ~1;
  const uint32_t *gc = fcodes_ws + P.fcodes_off;
  int32_t *p32 = path32_ws + P.path32_off;
  int64_t *p64 = P.path;
  int acc = 0;                                          // lane l: path value of column 64 q + l of the current group q
  auto flush = [&](int q) {
    const int col = 64 * q + lane;
    if (col < T) path_lds[col] = acc;
  };
  // path[c] = row, then the code of column c (for c >= 1) leads to path[c - 1] (src/dtw.jl:140-142)
  auto walk = [&](int tb, const uint32_t (&w)[3], int anchor) {
    const int thi = (16 * tb + 15 < T - 1) ? 16 * tb + 15 : T - 1, tlo = (16 * tb > 1) ? 16 * tb : 1;
    for (int t = thi; t >= tlo; --t) {
      const int idx = row - (anchor - 191);
      const int l = idx & 63;
      const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)w[0], l), w1 = (uint32_t)__builtin_amdgcn_readlane((int)w[1], l),
                     w2 = (uint32_t)__builtin_amdgcn_readlane((int)w[2], l);
      const uint32_t word = (idx < 64) ? w0 : ((idx < 128) ? w1 : w2);
      row -= (int)((word >> (2 * (t & 15))) & 3u);
      const int c = t - 1;
      if ((c & 63) == 63) flush((c + 1) >> 6);          // column c opens a new group: the finished one leaves
      acc = (lane == (c & 63)) ? row : acc;
    }
  };
  acc = (lane == ((T - 1) & 63)) ? row : acc;
  uint32_t wa[3] = {0, 0, 0}, wb[3] = {0, 0, 0}, wc[3] = {0, 0, 0};
  int aa = row, ab = row, ac = row;
  int tb = (T - 1) >> 4;
  dtw_window_load(gc, Se, tb, row, lane, wa);
  dtw_window_load(gc, Se, tb - 1, row, lane, wb);
  dtw_window_load(gc, Se, tb - 2, row, lane, wc);
  for (; tb >= 0; tb -= 3) {
    walk(tb, wa, aa);
    aa = row;
    dtw_window_load(gc, Se, tb - 3, row, lane, wa);
    if (tb - 1 < 0) break;
    walk(tb - 1, wb, ab);
    ab = row;
    dtw_window_load(gc, Se, tb - 4, row, lane, wb);
    if (tb - 2 < 0) break;
    walk(tb - 2, wc, ac);
    ac = row;
    dtw_window_load(gc, Se, tb - 5, row, lane, wc);
  }
  flush(0);
  __syncthreads();
  for (int col = lane; col < T; col += 64) {
    const int v = path_lds[col];
    p32[col] = v;
    if (p64) p64[col] = (int64_t)v + 1;
  }
}

// align() of the fused path: one workgroup per pair that wants `newtgt`, the path from path32_ws.
__global__ void __launch_bounds__(256)
dtw_fused_align_kernel(const double *__restrict__ feats, const DtwPair *__restrict__ pairs, int D, int Smax, int Tmax,
                       const int32_t *__restrict__ path32_ws) {
  const DtwPair P = pairs[blockIdx.x];
  if (!P.newtgt || P.T == 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int SmaxE = (Smax + 1) & ~1;
  int32_t *path32 = reinterpret_cast<int32_t *>(smem_raw);            // [Tmax]
  int32_t *owner = path32 + Tmax;                                     // [SmaxE]
  int32_t *holes = owner + SmaxE;                                     // [SmaxE]
  for (int t = threadIdx.x; t < P.T; t += 256) path32[t] = path32_ws[P.path32_off + t];
  __syncthreads();
  dtw_align_post(P, feats + P.seq_off, D, path32, owner, holes);
}

// ------------------------------------------------------------------------------------------------
// Generic kernel: any D, S up to the LDS capacity of two cost columns; rows strided over threads,
// template read from HBM/L2, step codes as bytes in HBM.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
dtw_generic_kernel(const double *__restrict__ feats, const DtwPair *__restrict__ pairs, int D, int fstep, int bstep, int Smax, int Tmax,
                   unsigned char *__restrict__ gscr, size_t gscr_stride) {
  // gscr (templates beyond ~5800 frames: two cost columns, the path and the align() lists no longer fit LDS): the same
  // arrays in a per-pair HBM scratch -- the reference has no length limit (src/dtw.jl:11-59); __syncthreads orders the
  // workgroup's global accesses as it orders its LDS accesses
  const DtwPair P = pairs[blockIdx.x];
  const int S = P.S, T = P.T, tid = threadIdx.x, nthr = blockDim.x;
  extern __shared__ unsigned char smem_lds[];
  unsigned char *smem_raw = gscr ? gscr + (size_t)blockIdx.x * gscr_stride : smem_lds;
  double *cbuf = reinterpret_cast<double *>(smem_raw);
  int32_t *path32 = reinterpret_cast<int32_t *>(cbuf + 2 * Smax);
  int32_t *owner = path32 + Tmax;
  int32_t *holes = owner + Smax;
  if (T == 0) return;
  for (int r = tid; r < S; r += nthr) {
    cbuf[r] = (double)(r + 1);
    if (P.cost) P.cost[r] = (double)(r + 1);
    if (P.bp) P.bp[r] = r + 1;
  }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const double *cp = cbuf + (t & 1) * Smax;
    double *cn = cbuf + ((t + 1) & 1) * Smax;
    const double *__restrict__ v = feats + P.seq_off + (size_t)D * t;
    for (int r = tid; r < S; r += nthr) {
      const double *__restrict__ tc = feats + P.tmpl_off + (size_t)D * r;
      double o = 0.0;
      for (int d = 0; d < D; ++d) {
        const double df = v[d] - tc[d];
        const double sq = df * df;
        o = o + sq;
      }
      int arg = r;
      double best = (cp[r] + o) + 1.0;
      for (int j = r - bstep; j <= r + fstep; ++j) {
        if (j < 0 || j >= S) continue;
        const double c = (cp[j] + o) + dtw_transition(j, r);
        if (c < best) { best = c; arg = j; }
      }
      cn[r] = best;
      if (P.cost) P.cost[(size_t)S * (t + 1) + r] = best;
      if (P.bp) P.bp[(size_t)S * (t + 1) + r] = arg + 1;
      P.codes[(size_t)S * t + r] = (unsigned char)(r - arg + fstep);
    }
    __syncthreads();
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  dtw_finish<false, 8>(P, feats + P.seq_off, D, fstep, cbuf + (T & 1) * Smax, path32, owner, holes, nullptr, Smax);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static const size_t kLdsLimit = 160 * 1024 - 256;   // dynamic LDS budget; leaves room for the kernels' static __shared__

struct DtwLaunch {
  std::vector<DtwPair> pairs;
  int D, fstep, bstep;
};

static int launch_rec(const double *feats, const DtwPair *dpairs, int n, int D, int fstep, int bstep, int Smax, int Tmax,
                      int steps, int codes, size_t shmem, int threads, const double *obs_ws, hipStream_t st) {
#define VCMI_DTW_LAUNCH(ST, CO)                                                                                \
  do {                                                                                                         \
    auto kern = dtw_rec_kernel<ST, CO>;                                                                        \
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 (int)shmem));                                                                 \
    hipLaunchKernelGGL(kern, dim3(n), dim3(threads), shmem, st, feats, dpairs, D, fstep, bstep, Smax, Tmax, obs_ws); \
  } while (0)
  if (steps == 1 && codes == 0) VCMI_DTW_LAUNCH(1, 0);
  else if (steps == 1) VCMI_DTW_LAUNCH(1, 2);
  else if (steps == 2 && codes == 0) VCMI_DTW_LAUNCH(2, 0);
  else if (steps == 2) VCMI_DTW_LAUNCH(2, 2);
  else if (codes == 0) VCMI_DTW_LAUNCH(0, 0);
  else if (codes == 1) VCMI_DTW_LAUNCH(0, 1);
  else VCMI_DTW_LAUNCH(0, 2);
#undef VCMI_DTW_LAUNCH
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

// Side stream on which the observation kernels run: the recurrence of chunk c (latency-bound: one barrier per column,
// few FP64 instructions) overlaps the observation costs of chunk c+1 (FP64-ALU-bound) on the same CUs.
struct DtwOverlap {
  static constexpr int kMaxChunks = 8;
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, done[kMaxChunks] = {};
  int device = -1;
  int init() {
    int dev = 0;
    VCMI_HIP(hipGetDevice(&dev));
    if (side && dev == device) return VCMI_OK;
    release();
    VCMI_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    VCMI_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (auto &e : done) VCMI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    device = dev;
    return VCMI_OK;
  }
  void release() {
    if (side) (void)hipStreamDestroy(side);
    if (fork) (void)hipEventDestroy(fork);
    for (auto &e : done)
      if (e) (void)hipEventDestroy(e);
    side = nullptr;
    fork = nullptr;
    for (auto &e : done) e = nullptr;
  }
  ~DtwOverlap() { release(); }
};
static DtwOverlap &dtw_overlap() {
  static thread_local DtwOverlap o;
  return o;
}

static int dtw_dmax(int D) {
  return D <= 8 ? 8 : D <= 16 ? 16 : D <= 24 ? 24 : D <= 32 ? 32 : D <= 40 ? 40 : D <= 48 ? 48 : D <= 64 ? 64 : 96;
}

static int launch_obs(const double *feats, const double *spad, const DtwPair *dpairs, int n, int D, int Smax, int Tmax,
                      double *obs_ws, hipStream_t st) {
  const int dmax = dtw_dmax(D);
  const int padded = (D != dmax);
  const double *sbase = padded ? spad : feats;
  if (padded) hipLaunchKernelGGL(dtw_pad_kernel, dim3(n, 8), dim3(256), 0, st, feats, dpairs, D, dmax, const_cast<double *>(spad));
  const unsigned colblocks = (Tmax + kObsCols - 1) / kObsCols;
  if (dmax <= 40) {   // hand-scheduled column loop, one template frame per lane (5 waves per SIMD)
    const dim3 grid(n, (Smax + kObsRows - 1) / kObsRows, colblocks);
#define VCMI_OBS_CASE(DM) \
  case DM: hipLaunchKernelGGL((dtw_obs_asm_kernel<DM>), grid, dim3(kObsRows), 0, st, sbase, padded, dpairs, obs_ws); break;
    switch (dmax) { VCMI_OBS_CASE(8) VCMI_OBS_CASE(16) VCMI_OBS_CASE(24) VCMI_OBS_CASE(32) VCMI_OBS_CASE(40) }
#undef VCMI_OBS_CASE
  } else {            // compiler-scheduled loop (two frames per lane at DMAX = 48 spills SGPRs: one frame per lane)
    const dim3 grid(n, (Smax + kObsRows - 1) / kObsRows, colblocks);
    switch (dmax) {
      case 48: hipLaunchKernelGGL((dtw_obs_kernel<48, 1>), grid, dim3(kObsRows), 0, st, sbase, padded, dpairs, obs_ws); break;
      case 64: hipLaunchKernelGGL((dtw_obs_kernel<64, 1>), grid, dim3(kObsRows), 0, st, sbase, padded, dpairs, obs_ws); break;
      default: hipLaunchKernelGGL((dtw_obs_kernel<96, 1>), grid, dim3(kObsRows), 0, st, sbase, padded, dpairs, obs_ws); break;
    }
  }
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

// per-thread scratch shared by the DTW entry points
struct DtwScratch {
  DevBuf<unsigned char> codes, longscr;      // longscr: cost columns / path / lists of the generic kernel beyond the LDS lengths
  DevBuf<DtwPair> dpairs;
  DevBuf<double> feats, cost, newtgt, obs, spad;
  DevBuf<int64_t> paths, bp;
  // Stream ordering of the shared workspaces (descriptors, observation costs, step codes): a call on ANY stream first
  // waits for the previous call's last kernel (`last_use`), and the descriptors travel to the device by an asynchronous
  // copy on the caller's stream from a small ring of pinned slots (a slot is reused only after its copy completed).
  // The ring is DEEP on purpose: a caller that issues calls back to back blocks in stage() once every slot is in flight, and
  // a blocked host thread wakes up late when the machine is busy (measured on a box with load average 30: 3-4.5 ms per
  // call instead of the GPU's 1.6 ms with 6 slots = two calls in flight) -- with 24 slots the GPU has many calls queued
  // and a late wake-up costs nothing.  (256 KB of pinned memory per slot at the benchmark batch.)
  static constexpr int kSlots = 24;
  hipEvent_t last_use = nullptr, slot_done[kSlots] = {};
  unsigned char *slot[kSlots] = {};
  size_t slot_cap[kSlots] = {};
  int next_slot = 0, device = -1;
  // fused path: strip descriptors, packed step codes, strip boundaries, last cost columns, strip-completion flags
  DevBuf<unsigned char> ddesc;   // fused path: pair, strip and job descriptors of the call (one upload)
  DevBuf<uint32_t> fcodes;
  DevBuf<double> bnd, clast;
  DevBuf<int32_t> path32;        // fused path: 0-based paths (dtw_fused_backward_kernel -> dtw_fused_align_kernel)
  DevBuf<int> flags;
  int epoch = 0;
  int init() {
    int dev = 0;
    VCMI_HIP(hipGetDevice(&dev));
    if (last_use && dev == device) return VCMI_OK;
    release_sync();
    VCMI_HIP(hipEventCreateWithFlags(&last_use, hipEventDisableTiming));
    for (auto &e : slot_done) VCMI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    device = dev;
    return VCMI_OK;
  }
  // pinned copy of `n` descriptors, valid until the returned slot's event (recorded by the caller) completes
  template <typename Desc>
  int stage(const Desc *src, size_t n, Desc **out, hipEvent_t *ev) {
    unsigned char *raw = nullptr;
    VCMI_TRY(stage_raw(n * sizeof(Desc), &raw, ev));
    memcpy(raw, src, n * sizeof(Desc));
    *out = reinterpret_cast<Desc *>(raw);
    return VCMI_OK;
  }
  // `bytes` of the next pinned slot (to be filled by the caller), valid until the slot's event -- recorded by the caller
  // after the copy it issues -- completes
  int stage_raw(size_t bytes, unsigned char **out, hipEvent_t *ev) {
    const int s = next_slot;
    next_slot = (next_slot + 1) % kSlots;
    VCMI_HIP(hipEventSynchronize(slot_done[s]));
    if (bytes > slot_cap[s]) {
      if (slot[s]) (void)hipHostFree(slot[s]);
      slot[s] = nullptr;
      slot_cap[s] = 0;
      const size_t cap = std::max<size_t>(bytes, (size_t)128 << 10);
      VCMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&slot[s]), cap, hipHostMallocDefault));
      slot_cap[s] = cap;
    }
    *out = slot[s];
    *ev = slot_done[s];
    return VCMI_OK;
  }
  // asynchronous upload of descriptors on `st` through a pinned slot
  template <typename Desc>
  int upload(Desc *dst, const Desc *src, size_t n, hipStream_t st) {
    Desc *pinned = nullptr;
    hipEvent_t copied = nullptr;
    VCMI_TRY(stage(src, n, &pinned, &copied));
    VCMI_HIP(hipMemcpyAsync(dst, pinned, sizeof(Desc) * n, hipMemcpyHostToDevice, st));
    VCMI_HIP(hipEventRecord(copied, st));
    return VCMI_OK;
  }
  void release_sync() {
    if (last_use) (void)hipEventDestroy(last_use);
    last_use = nullptr;
    for (int i = 0; i < kSlots; ++i) {
      if (slot_done[i]) (void)hipEventDestroy(slot_done[i]);
      if (slot[i]) (void)hipHostFree(slot[i]);
      slot_done[i] = nullptr;
      slot[i] = nullptr;
      slot_cap[i] = 0;
    }
  }
  ~DtwScratch() { release_sync(); }
};
static DtwScratch &scratch() {
  static thread_local DtwScratch s;
  return s;
}

// Fused path of dtw_run (see dtw_fused_kernel): strips, workspaces, the forward launch and the backward / align launch.
// Dynamic LDS of the two fused kernels; the caller takes the observation + recurrence path when either exceeds the budget
// (sequences beyond ~9.7k frames: the forward kernel keeps the boundary costs of a strip, 16 bytes per column, in LDS).
static inline size_t dtw_fused_lds_forward(int Tmax) {
  return (size_t)4 * VCMI_FUSED_OUTBOX + (size_t)2 * kFusedRows * 8 + ((size_t)Tmax + 1) * 16;
}
static inline size_t dtw_fused_lds_finish(int Smax, int Tmax) {      // dtw_fused_align_kernel: path, owner, holes
  const size_t SmaxE = (size_t)((Smax + 1) & ~1);
  return (size_t)Tmax * 4 + SmaxE * 8;
}
static inline bool dtw_fused_fits(int Smax, int Tmax) {
  return dtw_fused_lds_forward(Tmax) <= kLdsLimit && dtw_fused_lds_finish(Smax, Tmax) <= kLdsLimit;
}

// Row length the fused forward kernel works on.  D = 41 -- order-40 mel-cepstra WITH c0, what the reference's own pipeline
// aligns (bin/mcep.jl:12 --order=40, bin/align.jl:42-45 -> src/align.jl:45 -> src/dtw.jl:33-35) -- has its own loop that
// reads the 41-double rows as they lie in memory (no padded copy); every other D is padded up to the next instantiation.
// descriptors of a call: pinned host slot -> device (16 bytes per thread)
__global__ void __launch_bounds__(256) dtw_desc_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

static constexpr int kFusedMaxD = 48;
static int dtw_fused_row(int D) { return D == 41 ? 41 : dtw_dmax(D); }

static int dtw_run_fused(const double *feats, std::vector<DtwPair> &pairs, int D, int bstep, DtwScratch &sc, hipStream_t st,
                         int Smax, int Tmax) {
  const int n = (int)pairs.size();
  const int dmax = dtw_fused_row(D);
  // single-strip pairs largest first (shortens the tail); the strips of long templates are ordered by level: all
  // bottom strips lead the grid -- they are what the upper strips wait for -- and the upper strips follow the
  // single-strip pairs, by which time their predecessors are done.
  std::stable_sort(pairs.begin(), pairs.end(), [](const DtwPair &a, const DtwPair &b) {
    return (int64_t)a.S * a.T > (int64_t)b.S * b.T;
  });
  size_t ncodes = 0, nclast = 0, nbnd = 0, padn = 0, npath = 0;
  bool any_align = false;
  int nflags = 0;
  // pass 1: the row strips of every pair, whole-length (col0 = 0, ncol = T): one-wave or wider bottom strips of long
  // templates (levels[0]), single-strip pairs, the strips above by level.
  std::vector<std::vector<DtwStrip>> levels;
  std::vector<DtwStrip> singles;
  for (int k = 0; k < n; ++k) {
    DtwPair &p = pairs[k];
    const int Se = (p.S + 1) & ~1;
    p.fcodes_off = (int64_t)ncodes;
    ncodes += (size_t)((p.T + 15) >> 4) * Se;
    p.clast_off = (int64_t)nclast;
    nclast += (size_t)p.S;
    p.path32_off = (int64_t)npath;
    npath += (size_t)p.T;
    any_align = any_align || p.newtgt != nullptr;
    p.spad_off = (int64_t)padn;
    padn += (size_t)p.T * dmax;
    p.tpad_off = (int64_t)padn;
    padn += (size_t)p.S * dmax;
    if (p.T == 0) continue;
    const int nstr = (p.S + kFusedRows - 1) / kFusedRows;
    if (nstr == 1) {
      singles.push_back(DtwStrip{k, 0, p.S, -1, -1, -1, -1, 0, p.T, -1, 0});
      continue;
    }
    int first = p.S - kFusedRows * (nstr - 1);      // remainder strip at the bottom, an even number of rows
    first += first & 1;
    int row = 0;
    for (int s = 0; s < nstr; ++s) {
      const int rows = (s == 0) ? first : std::min(kFusedRows, p.S - row);
      DtwStrip ds{k, row, rows, -1, -1, -1, -1, 0, p.T, -1, 0};
      if (s > 0) ds.bnd_in = (int64_t)(nbnd - (size_t)2 * p.T);
      if (s + 1 < nstr) {
        ds.bnd_out = (int64_t)nbnd;
        nbnd += (size_t)2 * p.T;
      }
      if ((int)levels.size() <= s) levels.resize(s + 1);
      levels[s].push_back(ds);
      row += rows;
    }
  }
  // Column segments.  The jobs of a batch take about the same time (T columns each), so with one whole-length job per
  // workgroup a slot runs a whole number of them and the batch takes ceil(jobs / slots) job times: 1093 jobs on 512 slots
  // = 3, for 2.13 job times of work.  Cutting every strip's columns into `nseg` segments (a segment starts from the last
  // cost column its predecessor left in clast_ws) makes the unit smaller: 5 x 1093 jobs = 10.7 fifth-rounds.  A 100-column
  // segment costs what a 100-column sequence costs (7k cycles of prologue + 2.3k per column, cycle counter) PROVIDED the
  // jobs synchronise without agent-scope fences (see dtw_wait_flag): with one acquire / release pair per job the same
  // segment took 350k cycles instead of 250k and segments did not pay (1.72 / 1.68 / 1.74 / 1.82 / 2.43 ms for nseg = 1 / 2
  // / 3 / 4 / 8) -- which an earlier revision of this comment blamed on workgroup dispatch.  Now 1.69 / 1.52 / 1.50 / 1.51 /
  // 1.49 / 1.50 / 1.52 / 1.57 ms for nseg = 1 / 2 / 3 / 4 / 5 / 6 / 8 / 10.  nseg minimises ceil(jobs / slots) / nseg with 4 % per
  // extra segment -- its prologue, and above all its TRAFFIC: every segment job loads its strip's template rows again, 0.1 GB
  // per 1000 pairs and segment (0.39 GB at nseg = 1, 0.78 GB at nseg = 5, against 0.32 GB algorithmic), while the measured
  // times are flat from nseg = 2 to 8, so fewer is better (3 for the benchmark batch) --, segments of at least 64 columns,
  // nothing cut when everything fits one round.  The
  // segments are drawn from a ticket counter by persistent workgroups, one per slot (dtw_fused_persistent_kernel: 1.49 ms
  // against 1.53 ms with one workgroup per job in grid order at D = 40; the D = 48 kernel, whose persistent form spills 16
  // registers, stays with grid order: 1.77 against 1.80 ms).
  std::vector<DtwStrip> packed, wide;
  if (!levels.empty())
    for (const DtwStrip &ds : levels[0]) (ds.nrows <= 128 ? packed : wide).push_back(ds);
  size_t nwhole = wide.size() + singles.size();
  for (size_t l = 1; l < levels.size(); ++l) nwhole += levels[l].size();
  int slots = 512;
  {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    slots = 2 * cus;                              // two workgroups per CU (set by the registers)
  }
  int nseg = 1;
  const bool segments_allowed = !debug_flag(kDbgDtwNoSegments);
  {
    int64_t tsum = 0, tcnt = 0;
    for (const DtwPair &p : pairs)
      if (p.T > 0) {
        tsum += p.T;
        ++tcnt;
      }
    const int tavg = (int)(tsum / std::max<int64_t>(tcnt, 1));
    const double base_jobs = (double)((packed.size() + 3) / 4) + (double)nwhole;
    double best = 1e30;
    for (int c : {1, 2, 3, 4, 5, 6, 8}) {
      if (c > 1 && (!segments_allowed || tavg / c < 64 || base_jobs <= slots)) break;
      if (c > 2 && debug_flag(kDbgDtwTwoSegments)) break;
      const double t = std::ceil(base_jobs * c / slots) / c * (1.0 + 0.04 * (c - 1));
      if (t < best - 1e-9) {
        best = t;
        nseg = c;
      }
    }
  }
  // pass 2: job order, segment by segment: packed groups of one-wave bottom strips (four per workgroup, padded with empty
  // strips), the other bottom strips, the single-strip pairs (largest first), the upper strips level by level.  A job
  // waits only for jobs earlier in the order: the strip below (same segment) and its own previous segment.
  while (packed.size() % 4) packed.push_back(DtwStrip{packed.empty() ? 0 : packed.back().pair, 0, 0, -1, -1, -1, -1, 0, 0, -1, 0});
  std::vector<DtwStrip> whole = wide;
  whole.insert(whole.end(), singles.begin(), singles.end());
  for (size_t l = 1; l < levels.size(); ++l) whole.insert(whole.end(), levels[l].begin(), levels[l].end());
  // Segments of DECREASING length: the jobs drawn last decide how long slots idle at the end of the kernel (on average half
  // a job), so segment j gets the weight 1 + beta (nseg - 1 - 2 j) / (nseg - 1) -- 1.35 : 1 : 0.65 for three.  Measured at
  // beta = 0 / 0.2 / 0.3 / 0.4 / 0.5 / 0.6: 1.46 / 1.45 / 1.43 / 1.43 / 1.45 / 1.44 ms (D = 41: 1.59 -> 1.54).
  constexpr double seg_beta = 0.35;
  auto seg_bounds = [&](int T, int k) {            // first column of segment k (multiples of 16), k = nseg -> T
    if (k >= nseg) return T;
    if (k <= 0) return 0;
    double acc = 0.0;
    for (int j = 0; j < k; ++j) acc += 1.0 + (nseg > 1 ? seg_beta * (nseg - 1 - 2 * j) / (nseg - 1) : 0.0);
    return (int)(T * acc / nseg) & ~15;
  };
  // the strip below every upper strip: a packed bottom or a whole-length strip of the same pair that ends where it starts
  std::vector<int> below(whole.size(), -1);
  std::vector<char> below_packed(whole.size(), 0);
  {
    std::vector<std::vector<int>> by_pair_packed((size_t)n), by_pair_whole((size_t)n);
    for (size_t i = 0; i < packed.size(); ++i)
      if (packed[i].nrows > 0) by_pair_packed[(size_t)packed[i].pair].push_back((int)i);
    for (size_t i = 0; i < whole.size(); ++i) by_pair_whole[(size_t)whole[i].pair].push_back((int)i);
    for (size_t i = 0; i < whole.size(); ++i) {
      const DtwStrip &ds = whole[i];
      if (ds.bnd_in < 0) continue;
      for (int j : by_pair_packed[(size_t)ds.pair])
        if (packed[(size_t)j].row0 + packed[(size_t)j].nrows == ds.row0) {
          below[i] = j;
          below_packed[i] = 1;
        }
      for (int j : by_pair_whole[(size_t)ds.pair])
        if (whole[(size_t)j].row0 + whole[(size_t)j].nrows == ds.row0) below[i] = j;
    }
  }
  // Round 4 experiment (off by default: `vcmi_debug_force` kDbgDtwWholeFirst): whole-length jobs for the first rounds,
  // segments only where they balance.  Cutting EVERY strip makes every strip's template rows travel nseg times (2.7 x the
  // algorithmic bytes at nseg = 3), and the segments earn their keep in the last round only: with J jobs on S slots the first
  // floor(J / S) - 1 rounds are full whatever the unit is.  So that many slots' worth of single-strip pairs ran as ONE job
  // each, ahead of everything else, and only the rest was cut -- the same number of (1 / nseg)-rounds on paper, a third of
  // the re-reads (2.06 template loads per strip instead of 3: 1.55 x the algorithmic bytes).  Measured, 1000 pairs, one box,
  // alternating: 1.455 / 1.477 ms against 1.419 / 1.424 ms with every strip cut (two segments at most: 1.515): the long jobs
  // of the first round end ragged, and the slots that finish early find only segments whose predecessors are still running.
  // The re-reads are what the balance costs; every strip stays cut.
  std::vector<char> run_whole(whole.size(), 0);
  if (nseg > 1 && debug_flag(kDbgDtwWholeFirst)) {
    const double base_jobs = (double)(packed.size() / 4) + (double)whole.size();
    int64_t quota = (int64_t)slots * std::max<int64_t>(0, (int64_t)std::floor(base_jobs / slots) - 1);
    for (size_t i = wide.size(); i < wide.size() + singles.size() && quota > 0; ++i, --quota) run_whole[i] = 1;
  }
  std::vector<std::vector<int>> pflag(packed.size(), std::vector<int>((size_t)nseg, -1));
  std::vector<std::vector<int>> wflag(whole.size(), std::vector<int>((size_t)nseg, -1));
  std::vector<DtwStrip> strips;
  std::vector<int> job_first;
  auto segment_of = [&](DtwStrip ds, int k, std::vector<int> &flags_of, bool has_consumer_above) {
    const int T = ds.nrows > 0 ? pairs[(size_t)ds.pair].T : 0;
    ds.col0 = seg_bounds(T, k);
    ds.ncol = std::max(0, seg_bounds(T, k + 1) - ds.col0);
    ds.flag_prev = -1;
    for (int kp = k - 1; kp >= 0 && ds.flag_prev < 0; --kp) ds.flag_prev = flags_of[(size_t)kp];
    if (ds.ncol > 0 && (has_consumer_above || seg_bounds(T, k + 1) < T)) flags_of[(size_t)k] = ds.flag_out = nflags++;
    return ds;
  };
  for (int k = 0; k < nseg; ++k) {
    for (size_t i = 0; i + 3 < packed.size(); i += 4) {
      bool any = false;
      DtwStrip g4[4];
      for (int q = 0; q < 4; ++q) {
        g4[q] = segment_of(packed[i + q], k, pflag[i + q], packed[i + q].bnd_out >= 0);
        g4[q].packed = 1;
        any = any || g4[q].ncol > 0;
      }
      if (!any) continue;
      job_first.push_back((int)strips.size());
      strips.insert(strips.end(), g4, g4 + 4);
    }
    for (size_t i = 0; i < whole.size(); ++i) {
      DtwStrip ds = segment_of(whole[i], k, wflag[i], whole[i].bnd_out >= 0);
      if (run_whole[i]) {                          // one job over all columns, in the first pass
        if (k > 0) continue;
        ds = whole[i];
        ds.col0 = 0;
        ds.ncol = pairs[(size_t)ds.pair].T;
        ds.flag_prev = ds.flag_in = ds.flag_out = -1;      // a single strip: nobody waits for it, it waits for nobody
      }
      if (ds.ncol <= 0) continue;                  // (short sequences: fewer segments than nseg)
      // (the strip below has the same T, hence the same segment boundaries: its segment k exists whenever this one does)
      if (below[i] >= 0) ds.flag_in = below_packed[i] ? pflag[(size_t)below[i]][(size_t)k] : wflag[(size_t)below[i]][(size_t)k];
      job_first.push_back((int)strips.size());
      strips.push_back(ds);
    }
  }
  const unsigned ngroups = (unsigned)job_first.size();
  if (strips.empty()) return VCMI_OK;
  VCMI_TRY(sc.fcodes.reserve(ncodes));
  VCMI_TRY(sc.clast.reserve(nclast));
  VCMI_TRY(sc.path32.reserve(std::max<size_t>(npath, 1)));
  VCMI_TRY(sc.bnd.reserve(std::max<size_t>(nbnd, 1)));
  if ((size_t)nflags > sc.flags.n) {
    VCMI_TRY(sc.flags.reserve(std::max<size_t>((size_t)nflags, 1024)));
    VCMI_HIP(hipMemsetAsync(sc.flags.p, 0, sc.flags.n * sizeof(int), st));
    sc.epoch = 0;
  }
  const int epoch = ++sc.epoch;
  // one upload for everything the kernels read: [pairs | strips | first descriptor of every job, ticket counter = 0]
  const size_t o_strips = ((size_t)n * sizeof(DtwPair) + 255) & ~(size_t)255;
  const size_t o_jobs = (o_strips + strips.size() * sizeof(DtwStrip) + 255) & ~(size_t)255;
  const size_t desc_bytes = (o_jobs + (job_first.size() + 1) * sizeof(int) + 15) & ~(size_t)15;
  VCMI_TRY(sc.ddesc.reserve(desc_bytes));
  unsigned char *descp = sc.ddesc.p;
  {
    unsigned char *pinned = nullptr;
    hipEvent_t copied = nullptr;
    VCMI_TRY(sc.stage_raw(desc_bytes, &pinned, &copied));
    memcpy(pinned, pairs.data(), (size_t)n * sizeof(DtwPair));
    memcpy(pinned + o_strips, strips.data(), strips.size() * sizeof(DtwStrip));
    memcpy(pinned + o_jobs, job_first.data(), job_first.size() * sizeof(int));
    memset(pinned + o_jobs + job_first.size() * sizeof(int), 0, sizeof(int));
    // a copy KERNEL reading the pinned slot, not hipMemcpyAsync: the copy engine's hand-over to the compute queue left the
    // GPU idle for ~36 us between consecutive calls (kernel trace); a kernel on the same queue starts within a few us
    hipLaunchKernelGGL(dtw_desc_copy_kernel, dim3((unsigned)((desc_bytes / 16 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const uint4 *>(pinned), reinterpret_cast<uint4 *>(sc.ddesc.p), desc_bytes / 16);
    VCMI_HIP(hipEventRecord(copied, st));
  }
  const DtwPair *dpairs = reinterpret_cast<const DtwPair *>(descp);
  const DtwStrip *dstrips = reinterpret_cast<const DtwStrip *>(descp + o_strips);
  int *djobs = reinterpret_cast<int *>(descp + o_jobs);
  int *ticket = djobs + job_first.size();
  const bool padded = (D != dmax);
  if (padded) {
    VCMI_TRY(sc.spad.reserve(padn));
    hipLaunchKernelGGL(dtw_pad_kernel, dim3(n, 8), dim3(256), 0, st, feats, dpairs, D, dmax, sc.spad.p);
  }
  const double *base = padded ? sc.spad.p : feats;
  const size_t shf = dtw_fused_lds_forward(Tmax);
  if (shf > kLdsLimit) return fail(VCMI_ERR_ARG, "DTW: sequence of %d frames exceeds the supported length", Tmax);
#define VCMI_FUSED_LAUNCH(DM, STP)                                                                                    \
  do {                                                                                                                \
    if ((int)ngroups > slots && DM <= 41 && !debug_flag(kDbgDtwGridOrder)) { /* more jobs than slots: ticket-drawn */ \
      auto kernp = dtw_fused_persistent_kernel<DM, STP>;                                                              \
      VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernp), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)shf));                                                                        \
      hipLaunchKernelGGL(kernp, dim3((unsigned)slots), dim3(kFusedThreads), shf, st, base, (int)padded, dpairs,  \
                         dstrips, sc.fcodes.p, sc.bnd.p, sc.clast.p, sc.flags.p, epoch, djobs, (int)ngroups, \
                         ticket);                                                                                     \
    } else {                                                                                                          \
      auto kern = dtw_fused_kernel<DM, STP>;                                                                          \
      VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                   (int)shf));                                                                        \
      hipLaunchKernelGGL(kern, dim3(ngroups), dim3(kFusedThreads), shf, st, base, (int)padded, dpairs,           \
                         dstrips, sc.fcodes.p, sc.bnd.p, sc.clast.p, sc.flags.p, epoch, djobs);             \
    }                                                                                                                 \
  } while (0)
#define VCMI_FUSED_CASE(DM) \
  case DM: if (bstep == 1) VCMI_FUSED_LAUNCH(DM, 1); else VCMI_FUSED_LAUNCH(DM, 2); break;
  switch (dmax) {
    VCMI_FUSED_CASE(8) VCMI_FUSED_CASE(16) VCMI_FUSED_CASE(24) VCMI_FUSED_CASE(32) VCMI_FUSED_CASE(40) VCMI_FUSED_CASE(41)
    VCMI_FUSED_CASE(48)
  }
#undef VCMI_FUSED_CASE
#undef VCMI_FUSED_LAUNCH
  VCMI_HIP(hipGetLastError());
  // backward (a wave per pair), then align() for the pairs that want it
  hipLaunchKernelGGL(dtw_fused_backward_kernel, dim3((unsigned)n), dim3(64), (size_t)Tmax * 4, st, dpairs, sc.fcodes.p, sc.clast.p,
                     sc.path32.p);
  if (any_align) {
    const size_t shb = dtw_fused_lds_finish(Smax, Tmax);
    if (shb > kLdsLimit) return fail(VCMI_ERR_ARG, "DTW: template of %d frames exceeds the supported length", Smax);
    auto kern = dtw_fused_align_kernel;
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
    hipLaunchKernelGGL(kern, dim3(n), dim3(256), shb, st, feats, dpairs, D, Smax, Tmax, sc.path32.p);
  }
  VCMI_HIP(hipGetLastError());
  VCMI_HIP(hipEventRecord(sc.last_use, st));
#ifdef VCMI_DTW_PROF
  {
    unsigned long long h[8], z[8] = {0};
    (void)hipStreamSynchronize(st);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(dtw_prof), sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(dtw_prof), z, sizeof(z));
    if (h[3])
      fprintf(stderr, "dtw_prof: %llu jobs (nseg %d); cycles per job: prologue %.0f (of which flag wait %.0f), column loop %.0f, epilogue %.0f\n",
              h[3], nseg, (double)h[0] / h[3], (double)h[4] / h[3], (double)h[1] / h[3], (double)h[2] / h[3]);
  }
#endif
  return VCMI_OK;
}

// Launch one workgroup per pair.  `pairs` holds DEVICE pointers (codes filled in here when needed).
// codes_ws: grow-only device scratch for HBM step codes.
static int dtw_run(const double *feats, std::vector<DtwPair> &pairs, int D, int fstep, int bstep, DtwScratch &sc, hipStream_t st) {
  DevBuf<unsigned char> &codes_ws = sc.codes;
  DevBuf<double> &obs_ws = sc.obs, &spad_ws = sc.spad;
  DevBuf<DtwPair> &dpairs = sc.dpairs;
  const int n = (int)pairs.size();
  if (n == 0) return VCMI_OK;
  VCMI_TRY(sc.init());
  VCMI_HIP(hipStreamWaitEvent(st, sc.last_use, 0));   // the workspaces may still be read by the previous call's kernels
  if (D < 1) return fail(VCMI_ERR_DIM, "DTW: feature dimension %d invalid", D);
  if (fstep < 0 || bstep < 0 || fstep + bstep > 255) return fail(VCMI_ERR_ARG, "DTW: fstep/bstep out of range");
  int Smax = 0, Tmax = 0;
  for (auto &p : pairs) {
    if (p.S < 1) return fail(VCMI_ERR_DIM, "DTW: template must have at least one frame");
    Smax = std::max(Smax, p.S);
    Tmax = std::max(Tmax, p.T);
  }
  if (Tmax == 0) return VCMI_OK;
  {
    bool tables = false;
    for (auto &p : pairs) tables = tables || p.cost || p.bp;
    if (fstep == 0 && (bstep == 1 || bstep == 2) && D <= kFusedMaxD && !tables && !debug_flag(kDbgDtwTwoKernels) &&
        dtw_fused_fits(Smax, Tmax))
      return dtw_run_fused(feats, pairs, D, bstep, sc, st, Smax, Tmax);
  }
  // largest pairs first: shortens the tail when n is not a multiple of the resident workgroup count
  std::stable_sort(pairs.begin(), pairs.end(), [](const DtwPair &a, const DtwPair &b) {
    return (int64_t)a.S * a.T > (int64_t)b.S * b.T;
  });
  const int bits = (fstep + bstep + 1 <= 4) ? 2 : 8;
  const size_t base = (size_t)2 * Smax * 8 + (size_t)Tmax * 4 + (size_t)2 * Smax * 4;
  const size_t base_generic = base;
  // two cost columns + path + lists beyond LDS (a template of > ~5800 frames, or a SEQUENCE of > ~34k frames whatever the
  // template: the path alone is 4 T bytes): the generic kernel on an HBM scratch -- the two-kernel fast path keeps them in LDS
  const bool longseq = base > kLdsLimit;
  const bool fast = (Smax <= 1024 && D <= 96 && !longseq);
  const size_t codes_lds = (size_t)((Tmax + (32 / bits) - 1) / (32 / bits)) * Smax * 4;
  const bool ldscodes = fast && (base + codes_lds <= kLdsLimit);
  if (!ldscodes) {
    size_t total = 0;
    for (auto &p : pairs) total += (size_t)p.S * p.T;
    VCMI_TRY(codes_ws.reserve(total));
    size_t off = 0;
    for (auto &p : pairs) {
      p.codes = codes_ws.p + off;
      off += (size_t)p.S * p.T;
    }
  }
  VCMI_TRY(dpairs.reserve(n));
  if (fast) {
    // observation-cost workspace: S*T doubles per pair; large batches run in slices of at most ~4 GiB
    const int threads = (Smax + 63) / 64 * 64;
    const size_t shmem = base + (ldscodes ? codes_lds : 0);
    const int steps = (fstep == 0 && bstep == 1) ? 1 : (fstep == 0 && bstep == 2) ? 2 : 0;
    const int codes = !ldscodes ? 2 : (bits == 2 ? 0 : 1);
    const size_t kMaxCells = (size_t)1 << 29;
    int lo = 0;
    while (lo < n) {
      size_t cells = 0;
      int hi = lo;
      while (hi < n && (hi == lo || cells + (size_t)pairs[hi].S * pairs[hi].T <= kMaxCells)) {
        cells += (size_t)pairs[hi].S * pairs[hi].T;
        ++hi;
      }
      VCMI_TRY(obs_ws.reserve(cells));
      const int dmax = dtw_dmax(D);
      if (D != dmax) {
        size_t padn = 0;
        for (int k = lo; k < hi; ++k) padn += (size_t)(pairs[k].T + pairs[k].S) * dmax;
        VCMI_TRY(spad_ws.reserve(padn));
      }
      size_t off = 0, poff = 0;
      int smax = 0, tmax = 0;
      for (int k = lo; k < hi; ++k) {
        pairs[k].obs_off = (int64_t)off;
        off += (size_t)pairs[k].S * pairs[k].T;
        pairs[k].spad_off = (int64_t)poff;
        poff += (size_t)pairs[k].T * dmax;
        pairs[k].tpad_off = (int64_t)poff;
        poff += (size_t)pairs[k].S * dmax;
        smax = std::max(smax, pairs[k].S);
        tmax = std::max(tmax, pairs[k].T);
      }
      VCMI_TRY(sc.upload(dpairs.p + lo, pairs.data() + lo, (size_t)(hi - lo), st));
      if (tmax > 0) {
        // chunks of the slice: observation costs on the side stream, recurrences on the caller's stream behind them
        const int m = hi - lo;
        // (a recurrence launch occupies one CU per pair for ~T barriers whatever its size: chunks of about one pair
        // per CU keep every recurrence round full)
        const int nchunks = (m < 192) ? 1 : std::min(DtwOverlap::kMaxChunks, (m + 255) / 256);
        if (nchunks == 1) {
          VCMI_TRY(launch_obs(feats, spad_ws.p, dpairs.p + lo, m, D, smax, tmax, obs_ws.p, st));
          VCMI_TRY(launch_rec(feats, dpairs.p + lo, m, D, fstep, bstep, Smax, Tmax, steps, codes, shmem, threads, obs_ws.p, st));
        } else {
          DtwOverlap &ov = dtw_overlap();
          VCMI_TRY(ov.init());
          VCMI_HIP(hipEventRecord(ov.fork, st));                 // the side stream starts behind the caller's earlier work
          VCMI_HIP(hipStreamWaitEvent(ov.side, ov.fork, 0));     // (which includes the previous call's recurrences)
          for (int c = 0; c < nchunks; ++c) {
            const int c0 = lo + (int)((int64_t)m * c / nchunks), c1 = lo + (int)((int64_t)m * (c + 1) / nchunks);
            VCMI_TRY(launch_obs(feats, spad_ws.p, dpairs.p + c0, c1 - c0, D, smax, tmax, obs_ws.p, ov.side));
            VCMI_HIP(hipEventRecord(ov.done[c], ov.side));
            VCMI_HIP(hipStreamWaitEvent(st, ov.done[c], 0));
            VCMI_TRY(launch_rec(feats, dpairs.p + c0, c1 - c0, D, fstep, bstep, Smax, Tmax, steps, codes, shmem, threads, obs_ws.p, st));
          }
        }
      }
      if (hi < n) VCMI_HIP(hipStreamSynchronize(st));   // the next slice reuses the workspace
      lo = hi;
    }
    VCMI_HIP(hipEventRecord(sc.last_use, st));
    return VCMI_OK;
  }
  VCMI_TRY(sc.upload(dpairs.p, pairs.data(), (size_t)n, st));
  if (longseq) {
    const size_t stride = (base_generic + 255) / 256 * 256;
    VCMI_TRY(sc.longscr.reserve((size_t)n * stride));
    hipLaunchKernelGGL(dtw_generic_kernel, dim3(n), dim3(1024), 0, st, feats, dpairs.p, D, fstep, bstep, Smax, Tmax, sc.longscr.p, stride);
    VCMI_HIP(hipGetLastError());
    VCMI_HIP(hipEventRecord(sc.last_use, st));
    return VCMI_OK;
  }
  VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(dtw_generic_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)base_generic));
  hipLaunchKernelGGL(dtw_generic_kernel, dim3(n), dim3(1024), base_generic, st, feats, dpairs.p, D, fstep, bstep, Smax, Tmax,
                     (unsigned char *)nullptr, (size_t)0);
  VCMI_HIP(hipGetLastError());
  VCMI_HIP(hipEventRecord(sc.last_use, st));
  return VCMI_OK;
}

// Host-pointer batch on the current device: the feature matrices are gathered straight into the pinned staging slots
// (no intermediate host copy), run, and paths / tables / newtgt come back through the same ring.
static int dtw_host_batch_local(int64_t n, const double *const *tmpl, const int64_t *S, const double *const *seq,
                                const int64_t *T, int D, int fstep, int bstep, int64_t *const *path, double *cost,
                                int64_t *bp, double *const *newtgt) {
  VCMI_TRY(check_device());
  DtwScratch &sc = scratch();
  size_t nfeat = 0, npath = 0, ntgt = 0;
  for (int64_t p = 0; p < n; ++p) {
    nfeat += (size_t)D * (S[p] + T[p]);
    npath += (size_t)T[p];
    ntgt += (size_t)D * S[p];
  }
  VCMI_TRY(sc.feats.reserve(nfeat));
  VCMI_TRY(sc.paths.reserve(std::max<size_t>(npath, 1)));
  if (newtgt) VCMI_TRY(sc.newtgt.reserve(ntgt));
  const bool tables = (cost || bp);
  if (tables) {   // single-pair entry point only
    VCMI_TRY(sc.cost.reserve((size_t)S[0] * (T[0] + 1)));
    VCMI_TRY(sc.bp.reserve((size_t)S[0] * (T[0] + 1)));
  }
  std::vector<DtwPair> pairs(n);
  std::vector<HostPiece> up;
  up.reserve((size_t)2 * n);
  size_t fo = 0, po = 0, to = 0;
  for (int64_t p = 0; p < n; ++p) {
    DtwPair &q = pairs[p];
    up.push_back(HostPiece{const_cast<double *>(tmpl[p]), sizeof(double) * D * S[p]});
    q.tmpl_off = (int64_t)fo;
    fo += (size_t)D * S[p];
    if (T[p] > 0) up.push_back(HostPiece{const_cast<double *>(seq[p]), sizeof(double) * D * T[p]});
    q.seq_off = (int64_t)fo;
    fo += (size_t)D * T[p];
    q.path = sc.paths.p + po;
    po += (size_t)T[p];
    q.cost = tables ? sc.cost.p : nullptr;
    q.bp = tables ? sc.bp.p : nullptr;
    q.newtgt = newtgt ? sc.newtgt.p + to : nullptr;
    to += (size_t)D * S[p];
    q.codes = nullptr;
    q.obs_off = 0;
    q.spad_off = 0;
    q.tpad_off = 0;
    q.S = (int32_t)S[p];
    q.T = (int32_t)T[p];
  }
  VCMI_TRY(staged_upload_gather(sc.feats.p, up, nullptr));
  if (tables && T[0] == 0) {   // lazy_init! only: column 1 = 1:S
    for (int64_t i = 0; i < S[0]; ++i) {
      if (cost) cost[i] = (double)(i + 1);
      if (bp) bp[i] = i + 1;
    }
  }
  std::vector<DtwPair> order = pairs;   // dtw_run sorts its argument; keep `pairs` in caller order for the copy-back
  VCMI_TRY(dtw_run(sc.feats.p, order, D, fstep, bstep, sc, nullptr));
  if (path && npath > 0) {
    std::vector<HostPiece> down;
    std::vector<int64_t> dummy;
    bool all = true;
    for (int64_t p = 0; p < n; ++p) all = all && (path[p] != nullptr || T[p] == 0);
    if (all) {
      for (int64_t p = 0; p < n; ++p)
        if (T[p] > 0) down.push_back(HostPiece{path[p], sizeof(int64_t) * T[p]});
      VCMI_TRY(staged_download_scatter(down, sc.paths.p, nullptr));
    } else {   // some pairs want no path: take everything, hand out what was asked for
      dummy.resize(npath);
      VCMI_TRY(staged_download(dummy.data(), sc.paths.p, npath * 8, nullptr));
      po = 0;
      for (int64_t p = 0; p < n; ++p) {
        if (path[p] && T[p] > 0) memcpy(path[p], &dummy[po], sizeof(int64_t) * T[p]);
        po += (size_t)T[p];
      }
    }
  }
  if (newtgt) {
    // aligned targets: contiguous on the device in pair order; pairs without a sequence are all-zero (src/align.jl:19)
    std::vector<HostPiece> down;
    bool contiguous = true;
    for (int64_t p = 0; p < n; ++p) contiguous = contiguous && newtgt[p] && T[p] > 0;
    if (contiguous) {
      for (int64_t p = 0; p < n; ++p) down.push_back(HostPiece{newtgt[p], sizeof(double) * D * S[p]});
      VCMI_TRY(staged_download_scatter(down, sc.newtgt.p, nullptr));
    } else {
      VCMI_HIP(hipStreamSynchronize(nullptr));
      to = 0;
      for (int64_t p = 0; p < n; ++p) {
        if (newtgt[p]) {
          if (T[p] > 0) VCMI_HIP(hipMemcpy(newtgt[p], sc.newtgt.p + to, sizeof(double) * D * S[p], hipMemcpyDeviceToHost));
          else memset(newtgt[p], 0, sizeof(double) * D * S[p]);
        }
        to += (size_t)D * S[p];
      }
    }
  }
  if (tables && T[0] > 0) {
    if (cost) VCMI_TRY(staged_download(cost, sc.cost.p, sizeof(double) * S[0] * (T[0] + 1), nullptr));
    if (bp) VCMI_TRY(staged_download(bp, sc.bp.p, sizeof(int64_t) * S[0] * (T[0] + 1), nullptr));
  }
  VCMI_HIP(hipStreamSynchronize(nullptr));
  return VCMI_OK;
}

// pairs below which a batch stays on one device even when a device group is set
static constexpr int64_t kGroupMinPairs = 8;

// Pairs are independent: with a device group they are split by cost S*T (longest-processing-time) and every
// member aligns its share on its own device; no collective.
static int dtw_host_batch(int64_t n, const double *const *tmpl, const int64_t *S, const double *const *seq,
                          const int64_t *T, int D, int fstep, int bstep, int64_t *const *path, double *cost,
                          int64_t *bp, double *const *newtgt, bool allow_group = true) {
  if (n < 0) return fail(VCMI_ERR_ARG, "DTW: negative batch size");
  if (n == 0) return VCMI_OK;
  if (!tmpl || !S || !seq || !T) return fail(VCMI_ERR_ARG, "DTW: NULL argument");
  if (D < 1) return fail(VCMI_ERR_DIM, "DTW: feature dimension %d invalid", D);
  for (int64_t p = 0; p < n; ++p) {
    if (S[p] < 1 || T[p] < 0 || S[p] > INT32_MAX || T[p] > INT32_MAX) return fail(VCMI_ERR_DIM, "DTW: bad sequence length");
    if (!tmpl[p] || (T[p] > 0 && !seq[p])) return fail(VCMI_ERR_ARG, "DTW: NULL feature matrix");
  }
  const int m = group_size();
  if (!allow_group || m == 0 || n < kGroupMinPairs || cost || bp)
    return dtw_host_batch_local(n, tmpl, S, seq, T, D, fstep, bstep, path, cost, bp, newtgt);
  std::vector<int64_t> costs((size_t)n);
  for (int64_t p = 0; p < n; ++p) costs[(size_t)p] = S[p] * std::max<int64_t>(T[p], 1);
  const std::vector<int> part = shard_by_cost(costs, m);
  return group_run(m, [&](int i) -> int {
    std::vector<const double *> t2, s2;
    std::vector<int64_t> S2, T2;
    std::vector<int64_t *> p2;
    std::vector<double *> n2;
    for (int64_t p = 0; p < n; ++p) {
      if (part[(size_t)p] != i) continue;
      t2.push_back(tmpl[p]);
      s2.push_back(seq[p]);
      S2.push_back(S[p]);
      T2.push_back(T[p]);
      if (path) p2.push_back(path[p]);
      if (newtgt) n2.push_back(newtgt[p]);
    }
    if (t2.empty()) return VCMI_OK;
    return dtw_host_batch_local((int64_t)t2.size(), t2.data(), S2.data(), s2.data(), T2.data(), D, fstep, bstep,
                                path ? p2.data() : nullptr, nullptr, nullptr, newtgt ? n2.data() : nullptr);
  });
}

// align(src_p, tgt_p) for a batch, results LEFT ON THE DEVICE (dataset.hip builds the training matrix from them):
// *d_feats + src_off[p] is src_p (D,S_p), *d_newtgt + nt_off[p] the aligned target (D,S_p).  The buffers belong to the
// calling thread's DTW scratch and stay valid until its next DTW / align call.
int dtw_align_on_device(int64_t n, const double *const *src, const int64_t *S, const double *const *tgt, const int64_t *T,
                        int D, const double **d_feats, const double **d_newtgt, std::vector<int64_t> &src_off,
                        std::vector<int64_t> &nt_off) {
  std::vector<double *> nulls((size_t)std::max<int64_t>(n, 1), nullptr);
  // results stay in THIS thread's scratch on the current device: never spread over a device group
  VCMI_TRY(dtw_host_batch(n, src, S, tgt, T, D, /*fstep=*/0, /*bstep=*/2, nullptr, nullptr, nullptr, nulls.data(), false));
  DtwScratch &sc = scratch();
  *d_feats = sc.feats.p;
  *d_newtgt = sc.newtgt.p;
  src_off.resize((size_t)n);
  nt_off.resize((size_t)n);
  int64_t fo = 0, to = 0;
  for (int64_t p = 0; p < n; ++p) {
    src_off[p] = fo;
    nt_off[p] = to;
    fo += (int64_t)D * (S[p] + T[p]);
    to += (int64_t)D * S[p];
  }
  return VCMI_OK;
}

}  // namespace vcmi

using namespace vcmi;

extern "C" int vcmi_dtw_fit(const double *tmpl, int64_t S, const double *seq, int64_t T, int D, int fstep, int bstep,
                            int64_t *path, double *costtable, int64_t *backpointer) {
  int64_t *paths[1] = {path};
  const double *tp[1] = {tmpl}, *sp[1] = {seq};
  return dtw_host_batch(1, tp, &S, sp, &T, D, fstep, bstep, paths, costtable, backpointer, nullptr);
}

extern "C" int vcmi_dtw_fit_batch(int64_t n, const double *const *tmpl, const int64_t *S, const double *const *seq,
                                  const int64_t *T, int D, int fstep, int bstep, int64_t *const *path) {
  if (n > 0 && !path) return fail(VCMI_ERR_ARG, "vcmi_dtw_fit_batch: NULL path array");
  return dtw_host_batch(n, tmpl, S, seq, T, D, fstep, bstep, path, nullptr, nullptr, nullptr);
}

extern "C" int vcmi_dtw_fit_batch_dev(int64_t n, const double *feats, const int64_t *tmpl_off, const int64_t *S,
                                      const int64_t *seq_off, const int64_t *T, int D, int fstep, int bstep,
                                      int64_t *paths, const int64_t *path_off, void *stream) {
  if (n < 0) return fail(VCMI_ERR_ARG, "vcmi_dtw_fit_batch_dev: negative batch size");
  if (n == 0) return VCMI_OK;
  if (!feats || !tmpl_off || !S || !seq_off || !T || !paths || !path_off)
    return fail(VCMI_ERR_ARG, "vcmi_dtw_fit_batch_dev: NULL argument");
  std::vector<DtwPair> pairs(n);
  for (int64_t p = 0; p < n; ++p) {
    if (S[p] < 1 || T[p] < 0 || S[p] > INT32_MAX || T[p] > INT32_MAX) return fail(VCMI_ERR_DIM, "DTW: bad sequence length");
    DtwPair q{};
    q.tmpl_off = tmpl_off[p];
    q.seq_off = seq_off[p];
    q.path = paths + path_off[p];
    q.S = (int32_t)S[p];
    q.T = (int32_t)T[p];
    pairs[p] = q;
  }
  DtwScratch &sc = scratch();
  return dtw_run(feats, pairs, D, fstep, bstep, sc, as_stream(stream));
}

extern "C" int vcmi_align(const double *src, int64_t S, const double *tgt, int64_t T, int D, double *newtgt,
                          int64_t *path) {
  if (!newtgt) return fail(VCMI_ERR_ARG, "vcmi_align: NULL output");
  int64_t *paths[1] = {path};
  const double *tp[1] = {src}, *sp[1] = {tgt};
  double *nt[1] = {newtgt};
  return dtw_host_batch(1, tp, &S, sp, &T, D, /*fstep=*/0, /*bstep=*/2, paths, nullptr, nullptr, nt);   // src/align.jl:16
}

extern "C" int vcmi_align_batch(int64_t n, const double *const *src, const int64_t *S, const double *const *tgt,
                                const int64_t *T, int D, double *const *newtgt) {
  if (n > 0 && !newtgt) return fail(VCMI_ERR_ARG, "vcmi_align_batch: NULL output array");
  return dtw_host_batch(n, src, S, tgt, T, D, 0, 2, nullptr, nullptr, nullptr, newtgt);
}
