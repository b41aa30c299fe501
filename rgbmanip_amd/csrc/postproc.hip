// AdaPose post-processing on the GPU (gfx950): replaces the numpy code the reference runs on the
// host after every network call
//   /root/reference/models/pose_estimator/AdaPose/lib/utils.py:76-119  (compute_scale[_and_translation])
//   /root/reference/models/pose_estimator/AdaPose/lib/utils.py:40-74   (get_3d_bbox, transform_coordinates_3d)
//   /root/reference/models/pose_estimator/AdaPose/interface_v5.py:354-374 (bbox -> world, default_bbox)
// One workgroup (1024 threads) per pose.  The O(P^2) pair ratios are never stored: the exact median is
// found by radix selection, recomputing the ratios each pass from LDS-resident points (only i<j pairs: the
// reference's flattened P x P list holds every ratio twice and i==j is excluded by the nocs>0.01 filter, so
// both lists have the same median).  Arithmetic follows the reference's dtypes: NOCS distances in fp32,
// camera-space distances / ratios / median in fp64 (contraction disabled where numpy rounds twice).
//
// Two selection paths, both exact:
//  * generic: 12-bit-digit radix select over the order-preserving bit pattern of the positive fp64 ratios — every pass
//    evaluates all 523 776 ratios in fp64 (sqrt + divide: ~200 issue slots per pair; 1.5 ms per pose and workgroup);
//  * fast (the default; the generic one is its fall-back): the passes over all pairs work on an fp32 APPROXIMATION a of the
//    ratio x (|a - x| <= 4.5e-7 x by construction, PP_DELTA = 2^-18 assumed: 8x the bound) — two histogram passes locate a
//    bucket of approximations that holds the ranks n/2 - 1 and n/2, order statistics move by at most the perturbation, so
//    the exact median elements lie in that bucket widened by 2 PP_DELTA; the third pass counts the pairs that are certainly
//    below the window from the approximation alone and evaluates the exact fp64 ratio only for the few dozen pairs in or
//    next to it (exact count below + exact candidate keys -> the exact rank inside the candidates).  The result is the same
//    64-bit median; a pose whose selection does not fit the scheme (median outside [2^-10, 2^6), more than PP_CAND
//    candidates, inconsistent counts) takes the generic path.
#include "common.h"
#include "kernels.h"
#include "bbox_emit.h"

// numpy rounds every elementwise op separately: forbid a*b+c -> fma fusion in this file
#pragma clang fp contract(off)

namespace rgbm {

constexpr int PP_MAXP = 1024;
constexpr int PP_THREADS = 1024;
constexpr int PP_CAND = 3072;
constexpr unsigned PP_ABASE = (127u - 10u) << 8;   // fast path, first digit: float bits >> 15 (exponent + 8 mantissa bits) of [2^-10, 2^6)
constexpr double PP_DELTA = 0x1p-18;               // assumed bound of |a - x| / x (the arithmetic of pair_approx_core gives 8.5 * 2^-24)
constexpr int PP_F_GENERIC = 1, PP_F_GUARD = 2;    // kernel flags (debug): generic selection only; guard band in every even-count pose

struct PPArrays { const double *cx, *cy, *cz; const float *nx, *ny, *nz; };
struct PPPoint { double cx, cy, cz; float nx, ny, nz; };

__device__ __forceinline__ PPPoint pp_point(const PPArrays& A, int i) {
  return PPPoint{A.cx[i], A.cy[i], A.cz[i], A.nx[i], A.ny[i], A.nz[i]};
}

// the reference's ratio of pair (p, j), exactly: lib/utils.py:88-101
__device__ __forceinline__ bool pair_ratio(const PPPoint& p, const PPArrays& A, int j, double& ratio) {
  // plain IEEE operators (this file is built with -ffp-contract=off; sqrtf/sqrt/÷ are correctly rounded,
  // whereas HIP's __fsqrt_rn/__fdiv_rn wrappers lower to the approximate native instructions)
  const float dx = p.nx - A.nx[j], dy = p.ny - A.ny[j], dz = p.nz - A.nz[j];
  const float nd = sqrtf((dx * dx + dy * dy) + dz * dz);
  if (!(nd > 0.01f)) return false;
  const double ex = p.cx - A.cx[j], ey = p.cy - A.cy[j], ez = p.cz - A.cz[j];
  const double rd = sqrt((ex * ex + ey * ey) + ez * ez);
  if (!(rd < 0.3)) return false;
  ratio = rd / (double)nd;
  return true;
}

// The same pair with the fp64 square root and division replaced by fp32 ones: the VALIDITY is the exact one, the ratio a is an
// approximation.  Validity: the NOCS test nd > 0.01f is the reference's fp32 arithmetic, decided on the squared sum (sqrtf is
// correctly rounded and monotone: sqrtf(s) > 0.01f <=> s > PP_ND2, the largest float whose root rounds to 0.01f or less —
// tests/test_host_logic.py checks the constant); the 0.3 m test is decided in fp64 whenever the fp32 distance is within 3e-5 of it.
// Approximation: fp64 differences rounded to fp32 (2^-24 each), two rounded squares-and-sums (<= 5 * 2^-24 on the sum, half of it
// after the root), v_sqrt_f32 and v_rsq_f32 (1 ulp = 2^-23 each; the reference divides by the ROUNDED nd: 2^-24), one product:
// |a - x| <= 8.5 * 2^-24 x.  Squared distances below 2^-80 (ratios below 2^-33, far under the first digit's window) report a = 0.
// Straight-line but for the rare fp64 decision, so that the unrolled pair loops batch their LDS reads.
constexpr float PP_ND2 = 0x1.a36e30p-14f;
// straight-line part: a, the verdict `valid` where it is certain, `unc` where the 0.3 m test needs fp64
__device__ __forceinline__ void pair_approx_core(const PPPoint& p, const PPArrays& A, int j, float& a, bool& valid, bool& unc) {
  const float dx = p.nx - A.nx[j], dy = p.ny - A.ny[j], dz = p.nz - A.nz[j];
  const float s2 = (dx * dx + dy * dy) + dz * dz;
  const double ex = p.cx - A.cx[j], ey = p.cy - A.cy[j], ez = p.cz - A.cz[j];
  const float fx = (float)ex, fy = (float)ey, fz = (float)ez;
  const float r2 = (fx * fx + fy * fy) + fz * fz;
  valid = s2 > PP_ND2 && r2 < 0.090006f;             // r2 >= 0.30001^2: rd >= 0.3 for certain (or NaN / overflow)
  unc = valid && r2 > 0.089994f;                      // within 3e-5 of 0.3 m
  a = r2 < 0x1p-80f ? 0.f : __builtin_amdgcn_sqrtf(r2) * __builtin_amdgcn_rsqf(s2);
}
__device__ __forceinline__ bool pair_near_test(const PPPoint& p, const PPArrays& A, int j) {
  const double ex = p.cx - A.cx[j], ey = p.cy - A.cy[j], ez = p.cz - A.cz[j];
  const double rd = sqrt((ex * ex + ey * ey) + ez * ez);
  return rd < 0.3;
}

__device__ __forceinline__ unsigned pp_digit_a(float a) {
  const int d = (int)(__float_as_uint(a) >> 15) - (int)PP_ABASE;
  return d < 0 ? 0u : d > 4095 ? 4095u : (unsigned)d;
}

// all i<j pairs of row pair r (rows r and P-1-r together have P-1 partners), partners [q_lo, q_hi): f(point i, j)
template <class F>
__device__ __forceinline__ void for_pairs(const PPArrays& A, int P, int r, int q_lo, int q_hi, F&& f) {
  const int split = P - 1 - r;                   // q < split: i = r, j = r + 1 + q; else i = P - 1 - r, j = q + 1
  const int qb = q_hi < split ? q_hi : split;
  if (q_lo < qb) {
    const PPPoint p = pp_point(A, r);
    for (int q = q_lo; q < qb; ++q) f(p, r + 1 + q);
  }
  const int qa = q_lo > split ? q_lo : split;
  if (qa < q_hi) {
    const PPPoint p = pp_point(A, P - 1 - r);
    for (int q = qa; q < q_hi; ++q) f(p, q + 1);
  }
}

// the same enumeration through the approximation, four partners at a time (their LDS reads and arithmetic interleave; the fp64
// decisions of a group, rare, come after it): f(point i, j, valid, a)
template <class F>
__device__ __forceinline__ void for_pairs_approx(const PPArrays& A, int P, int r, int q_lo, int q_hi, F&& f) {
  const int split = P - 1 - r;
  auto run = [&](int i, int qa, int qb, int joff) {
    if (qa >= qb) return;
    const PPPoint p = pp_point(A, i);
    int q = qa;
    for (; q + 4 <= qb; q += 4) {
      float a[4];
      bool v[4], u[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) pair_approx_core(p, A, joff + q + k, a[k], v[k], u[k]);
      if (u[0] | u[1] | u[2] | u[3]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (u[k]) v[k] = pair_near_test(p, A, joff + q + k);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) f(p, joff + q + k, v[k], a[k]);
    }
    for (; q < qb; ++q) {
      float a;
      bool v, u;
      pair_approx_core(p, A, joff + q, a, v, u);
      if (u) v = pair_near_test(p, A, joff + q);
      f(p, joff + q, v, a);
    }
  };
  run(r, q_lo, q_hi < split ? q_hi : split, r + 1);
  run(P - 1 - r, q_lo > split ? q_lo : split, q_hi, 1);
}

__device__ double block_sum(double v, double* red) {
  // 1024 threads -> deterministic tree in LDS
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = PP_THREADS / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// Block-wide bucket selection (all 1024 threads, 4 bins each; `hist` complete and visible): the first bin whose cumulative count
// exceeds rank rk (rk = total / 2 when `half`), the count below it, its own count, the total.  An empty histogram (or a rank past
// the end) reports digit 4095 with eq = 0.  (A single thread walking 4096 bins per pass cost ~0.9 ms of the 3.3 ms the first
// version of this kernel took for 256 poses.)
struct PPSel { unsigned digit, below, eq, total; };
__device__ PPSel pp_select(const unsigned* hist, unsigned* wsum, unsigned* sres, bool half, unsigned rk_in) {
  const int t = threadIdx.x;
  unsigned c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = hist[4 * t + k];
  const unsigned tsum = c[0] + c[1] + c[2] + c[3];
  unsigned incl = tsum;
  const int ln = t & 63, wv = t >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(incl, off);
    if (ln >= off) incl += o;
  }
  if (ln == 63) wsum[wv] = incl;
  if (t == 0) { sres[0] = 4095u; sres[1] = 0u; sres[2] = 0u; }
  __syncthreads();
  unsigned wbase = 0, total = 0;
  for (int k = 0; k < PP_THREADS / 64; ++k) { const unsigned w = wsum[k]; if (k < wv) wbase += w; total += w; }
  const unsigned excl = wbase + incl - tsum;
  const unsigned rk = half ? total / 2 : rk_in;
  if (rk >= excl && rk < excl + tsum) {
    unsigned acc = excl;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (acc + c[k] > rk) { sres[0] = (unsigned)(4 * t + k); sres[1] = acc; sres[2] = c[k]; break; }
      acc += c[k];
    }
  }
  __syncthreads();
  const PPSel s{sres[0], sres[1], sres[2], total};
  __syncthreads();
  return s;
}

// block-wide: the last non-empty bin below d (-1: none)
__device__ int pp_prev_nonempty(const unsigned* hist, unsigned d, int* sprev) {
  const int t = threadIdx.x;
  if (t == 0) *sprev = -1;
  __syncthreads();
  int best = -1;
#pragma unroll
  for (int k = 0; k < 4; ++k) if ((unsigned)(4 * t + k) < d && hist[4 * t + k]) best = 4 * t + k;
  if (best >= 0) atomicMax(sprev, best);
  __syncthreads();
  const int r = *sprev;
  __syncthreads();
  return r;
}

// What the fast path knows after its two histogram passes (identical in every thread and in every kernel that derives it from the
// same two histograms).
struct PPPlan {
  bool ok;                 // the pose fits the scheme so far
  bool want_lo, guard;     // even count: the lower middle is needed; guard: it lies below the first digit's bucket (see below)
  unsigned n, keyA, rank;  // valid pairs; float bits >> 15 of the selected first bucket; rank n/2 inside that bucket
  float T0f, T1f, T4f;     // approximations below T0f: certainly below the window, uncounted candidates none; [T0f, T4f): evaluate exactly
  double Lx, Ux;           // the window of exact ratios that holds the median elements
  double Gx;               // guard: a largest-below-the-window ratio is only trusted above this (pairs under T0f cannot reach it)
};

__device__ __forceinline__ void pp_plan_a(PPPlan& pl, const PPSel& sa) {
  pl.n = sa.total;
  pl.ok = sa.total > 0 && sa.digit != 0u && sa.digit != 4095u;      // the end bins also hold everything outside the window
  pl.keyA = sa.digit + PP_ABASE;
  pl.rank = sa.total / 2 - sa.below;
  pl.want_lo = (sa.total & 1u) == 0u;
  pl.guard = false;
}

// second digit: bits 14..3 of the approximations inside the first bucket (buckets of 2^-20 relative width).  The window spans the
// buckets of ranks n/2 - 1 (when needed) and n/2.  If rank n/2 is the first of its first-digit bucket, the lower middle lies in
// an earlier first-digit bucket: the window then starts at the bucket of rank n/2 and a guard band of 2^-7 below it is evaluated
// exactly for the largest ratio under the window (rare: one pose in a few thousand).
__device__ void pp_plan_b(PPPlan& pl, const unsigned* hist, const PPSel& sb, int* sprev, int flags) {
  unsigned lowB = sb.digit;
  bool guard = (flags & PP_F_GUARD) && pl.want_lo;
  if (pl.want_lo && pl.rank == sb.below) {         // block-uniform
    if (sb.below == 0u) guard = true;
    else {
      const int pv = pp_prev_nonempty(hist, sb.digit, sprev);
      if (pv >= 0) lowB = (unsigned)pv; else pl.ok = false;
    }
  }
  if (sb.eq == 0u) pl.ok = false;
  pl.guard = guard;
  const float Lf = __uint_as_float((pl.keyA << 15) | (lowB << 3));
  const float Uf = __uint_as_float(((pl.keyA << 15) | (sb.digit << 3)) + 8u);
  pl.Lx = (double)Lf * (1.0 - 2.0 * PP_DELTA);
  pl.Ux = (double)Uf * (1.0 + 2.0 * PP_DELTA);
  pl.T1f = (float)((double)Lf * (1.0 - 4.0 * PP_DELTA));
  pl.T4f = (float)((double)Uf * (1.0 + 4.0 * PP_DELTA));
  pl.T0f = guard ? Lf * (1.f - 0x1p-7f) : pl.T1f;
  pl.Gx = (double)pl.T0f * (1.0 + 2.0 * PP_DELTA);
}

// third pass of the fast path over partners [q_lo, q_hi) of row pair r: count what lies below the window, collect the candidates
template <class AddCand, class LowKey>
__device__ __forceinline__ unsigned pp_collect(const PPPlan& pl, const PPArrays& A, int P, int r, int q_lo, int q_hi, AddCand&& add_cand,
                                               LowKey&& low_key) {
  unsigned nb = 0;
  for_pairs_approx(A, P, r, q_lo, q_hi, [&](const PPPoint& p, int j, bool v, float a) {
    nb += (v && a < pl.T0f) ? 1u : 0u;
    if (!(v && a >= pl.T0f && a < pl.T4f)) return;    // rare below here
    double x;
    if (!pair_ratio(p, A, j, x)) return;            // (cannot differ from the approximate pass's verdict)
    const unsigned long long key = (unsigned long long)__double_as_longlong(x);
    if (x < pl.Lx) { ++nb; if (pl.guard) low_key(key); }
    else if (x < pl.Ux) add_cand(key);
  });
  return nb;
}

// Split form for small batches (one workgroup per pose leaves 255 CUs idle at B = 1).  The three passes of the fast path are cut
// into G slices per pose, one workgroup each (pp_split_kernel, STAGE 0 / 1 / 2; the slices of a pose add their histograms /
// candidates / counts into global scratch), and the finishing kernel (postprocess_kernel<true>, one workgroup per pose) picks the
// state up from there: same integer histograms, same plan, same candidate set, hence the same median bit for bit.  A pose that
// does not fit the fast path is recomputed by the finishing kernel on the generic path.
struct PPScratch {                    // per pose, in caller-provided device memory (zeroed in front of every call)
  unsigned hist0[4096], hist1[4096];
  unsigned long long cand[PP_CAND];
  unsigned ncand, below;
  unsigned long long lowkey;
};

template <bool PRE>
__global__ __launch_bounds__(PP_THREADS) void postprocess_kernel(
    const float* __restrict__ nocs /*[B,P,3]*/, const float* __restrict__ depth /*[B,P]*/, const float* __restrict__ rot /*[B,9]*/,
    const int* __restrict__ choose /*[B,P]*/, const double* __restrict__ Kc /*[B,9]*/, const double* __restrict__ E1 /*[B,16]*/,
    double* __restrict__ bbox /*[B,8,3]*/, double* __restrict__ ts_out /*[B,4]: t(3), s*/, int* __restrict__ valid, int P, int img,
    const PPScratch* __restrict__ pre, int flags) {
  __shared__ double cx[PP_MAXP], cy[PP_MAXP], cz[PP_MAXP];
  __shared__ float nx[PP_MAXP], ny[PP_MAXP], nz[PP_MAXP];
  __shared__ double red[PP_THREADS];
  __shared__ unsigned hist[4096];
  __shared__ unsigned wsum[PP_THREADS / 64];
  __shared__ unsigned sres[4];
  __shared__ int sprev;
  __shared__ unsigned long long cand[PP_CAND];   // candidate keys (fast path: the window; generic: the selected 24-bit bucket)
  __shared__ unsigned ncand, s_below;
  __shared__ unsigned long long cand_med, cand_lo, s_lowkey;
  __shared__ float hmax[3];

  const int b = blockIdx.x, t = threadIdx.x;
  const double fx = Kc[b * 9 + 0], fy = Kc[b * 9 + 4], pcx = Kc[b * 9 + 2], pcy = Kc[b * 9 + 5];
  if (t < P) {
    const int ch = choose[(long long)b * P + t];
    const double u = (double)(ch % img), v = (double)(ch / img);
    const double z = (double)depth[(long long)b * P + t];
    cx[t] = ((u - pcx) * z) / fx;
    cy[t] = ((v - pcy) * z) / fy;
    cz[t] = z;
    nx[t] = nocs[((long long)b * P + t) * 3 + 0];
    ny[t] = nocs[((long long)b * P + t) * 3 + 1];
    nz[t] = nocs[((long long)b * P + t) * 3 + 2];
  }
  if (t < 3) hmax[t] = 0.f;
  if (t == 0) { ncand = 0u; s_below = 0u; s_lowkey = 0ull; cand_med = 0ull; cand_lo = 0ull; }
  __syncthreads();
  const PPArrays A{cx, cy, cz, nx, ny, nz};

  const int npair_rows = P / 2;                  // row pairs (P even)
  const int per_row = P - 1;
  const int nthr_per_row = PP_THREADS / npair_rows;   // >=2 for P<=1024 ... 1024/512 = 2
  const int r = t % npair_rows, part = t / npair_rows;
  const int q_lo = (int)(((long long)per_row * part) / nthr_per_row);
  const int q_hi = (int)(((long long)per_row * (part + 1)) / nthr_per_row);
  const bool worker = part < nthr_per_row;

  // rank the m candidates in LDS: the keys of ranks rc and (want_lo, rc >= 1) rc - 1 -> cand_med / cand_lo
  auto rank_candidates = [&](unsigned m, unsigned rc, bool want_lo) {
    for (unsigned c = t; c < m; c += PP_THREADS) {
      const unsigned long long kc = cand[c];
      unsigned lt = 0, eq_before = 0;
      for (unsigned o = 0; o < m; ++o) { const unsigned long long ko = cand[o]; lt += ko < kc; eq_before += (ko == kc) & (o < c); }
      if (lt + eq_before == rc) cand_med = kc;                      // exactly one candidate has each rank
      if (want_lo && lt + eq_before + 1u == rc) cand_lo = kc;
    }
    __syncthreads();
  };

  double med_hi = __builtin_nan(""), med_lo = __builtin_nan("");
  bool have = false, done = false;

  // ---- exact median of the valid pair ratios, fast path (see the head of the file) ----------------------
  if (!(flags & PP_F_GENERIC)) {
    if (PRE) {
      for (int i = t; i < 4096; i += PP_THREADS) hist[i] = pre[b].hist0[i];
    } else {
      for (int i = t; i < 4096; i += PP_THREADS) hist[i] = 0u;
      __syncthreads();
      if (worker)
        for_pairs_approx(A, P, r, q_lo, q_hi, [&](const PPPoint&, int, bool v, float a) {
          if (v) atomicAdd(&hist[pp_digit_a(a)], 1u);
        });
    }
    __syncthreads();
    const PPSel sa = pp_select(hist, wsum, sres, true, 0u);
    PPPlan pl;
    pp_plan_a(pl, sa);
    if (sa.total == 0u) done = true;             // no valid pair: scale = NaN (have stays false)
    if (pl.ok) {
      if (PRE) {
        for (int i = t; i < 4096; i += PP_THREADS) hist[i] = pre[b].hist1[i];
      } else {
        for (int i = t; i < 4096; i += PP_THREADS) hist[i] = 0u;
        __syncthreads();
        if (worker)
          for_pairs_approx(A, P, r, q_lo, q_hi, [&](const PPPoint&, int, bool v, float a) {
            const unsigned fb = __float_as_uint(a);
            if (v && (fb >> 15) == pl.keyA) atomicAdd(&hist[(fb >> 3) & 4095u], 1u);
          });
      }
      __syncthreads();
      const PPSel sb = pp_select(hist, wsum, sres, false, pl.rank);
      pp_plan_b(pl, hist, sb, &sprev, flags);
    }
    if (pl.ok) {
      unsigned m, below;
      if (PRE) {
        m = pre[b].ncand;
        below = pre[b].below;
        if (t == 0) s_lowkey = pre[b].lowkey;
        if (m <= (unsigned)PP_CAND)
          for (unsigned c = t; c < m; c += PP_THREADS) cand[c] = pre[b].cand[c];
        __syncthreads();
      } else {
        if (worker) {
          const unsigned nb = pp_collect(pl, A, P, r, q_lo, q_hi,
                                         [&](unsigned long long key) { const unsigned i = atomicAdd(&ncand, 1u); if (i < (unsigned)PP_CAND) cand[i] = key; },
                                         [&](unsigned long long key) { atomicMax(&s_lowkey, key); });
          if (nb) atomicAdd(&s_below, nb);
        }
        __syncthreads();
        m = ncand;
        below = s_below;
      }
      const unsigned k = pl.n / 2;               // upper middle (0-based) of the i<j list
      bool ok = m <= (unsigned)PP_CAND && k >= below && k - below < m;
      const unsigned rc = k - below;
      const unsigned long long lowkey = s_lowkey;
      if (ok && pl.want_lo && rc == 0u) ok = pl.guard && lowkey != 0ull && __longlong_as_double((long long)lowkey) > pl.Gx;
      if (ok) {                                   // block-uniform
        rank_candidates(m, rc, pl.want_lo);
        med_hi = __longlong_as_double((long long)cand_med);
        med_lo = !pl.want_lo ? med_hi : rc >= 1u ? __longlong_as_double((long long)cand_lo) : __longlong_as_double((long long)lowkey);
        have = true;
        done = true;
      }
    }
  }

  // ---- generic path: radix select over the fp64 keys, digits 12,12,12,12,12,4 bits ----------------------
  if (!done) {
    __syncthreads();
    if (t == 0) { ncand = 0u; cand_med = 0ull; cand_lo = 0ull; }
    unsigned long long prefix = 0ull;     // selected high bits so far (right-aligned)
    unsigned rank = 0;                    // rank to find within the current prefix bucket
    unsigned n_valid = 0;
    bool want_lo = false, lo_found = false;
    const int shifts[6] = {52, 40, 28, 16, 4, 0};
    const int widths[6] = {12, 12, 12, 12, 12, 4};
    for (int pass = 0; pass < 6; ++pass) {
      const int sh = shifts[pass], wd = widths[pass];
      for (int i = t; i < 4096; i += PP_THREADS) hist[i] = 0u;
      __syncthreads();
      if (worker)
        for_pairs(A, P, r, q_lo, q_hi, [&](const PPPoint& p, int j) {
          double ratio;
          if (!pair_ratio(p, A, j, ratio)) return;
          const unsigned long long key = (unsigned long long)__double_as_longlong(ratio);
          if (pass > 0 && (key >> (sh + wd)) != prefix) return;
          atomicAdd(&hist[(unsigned)((key >> sh) & ((1u << wd) - 1u))], 1u);
        });
      __syncthreads();
      const PPSel s = pp_select(hist, wsum, sres, pass == 0, rank);
      if (pass == 0) {
        n_valid = s.total;
        if (n_valid == 0u) break;
        rank = n_valid / 2;
        want_lo = (n_valid & 1u) == 0u;
      }
      prefix = (prefix << wd) | (unsigned long long)s.digit;
      rank -= s.below;
      // Short cut: after two passes the bucket is 2^-12 wide in relative terms and holds a few hundred of the 523 776 ratios.
      // Collect its keys once and finish the selection on that list instead of four more passes over all pairs.
      if (pass == 1 && s.eq <= (unsigned)PP_CAND) {
        if (worker)
          for_pairs(A, P, r, q_lo, q_hi, [&](const PPPoint& p, int j) {
            double ratio;
            if (!pair_ratio(p, A, j, ratio)) return;
            const unsigned long long key = (unsigned long long)__double_as_longlong(ratio);
            if ((key >> 40) == prefix) cand[atomicAdd(&ncand, 1u)] = key;
          });
        __syncthreads();
        rank_candidates(ncand /* == s.eq */, rank, want_lo);
        prefix = cand_med;                         // the full 64-bit key
        if (want_lo && rank >= 1u) { med_lo = __longlong_as_double((long long)cand_lo); lo_found = true; }
        break;
      }
      if (pass == 5 && want_lo && rank >= 1u) lo_found = true;      // the bucket of identical keys also holds the lower middle
    }
    if (n_valid > 0u) {
      med_hi = __longlong_as_double((long long)prefix);
      have = true;
      if (!want_lo || (lo_found && med_lo != med_lo)) med_lo = med_hi;
      if (want_lo && !lo_found) {
        // the lower middle is the largest valid ratio strictly below med_hi
        double best = -1.0;
        if (worker)
          for_pairs(A, P, r, q_lo, q_hi, [&](const PPPoint& p, int j) {
            double ratio;
            if (!pair_ratio(p, A, j, ratio)) return;
            if (ratio < med_hi && ratio > best) best = ratio;
          });
        red[t] = best;
        __syncthreads();
        for (int s = PP_THREADS / 2; s > 0; s >>= 1) {
          if (t < s) red[t] = red[t] > red[t + s] ? red[t] : red[t + s];
          __syncthreads();
        }
        med_lo = red[0];
        __syncthreads();
      }
    }
  }
  const double scale = have ? 0.5 * (med_lo + med_hi) : __builtin_nan("");

  // ---- translation = mean(cam) - mean(s*R*nocs)   (utils.py:113-118) --------------------------------
  double R[9];
  for (int i = 0; i < 9; ++i) R[i] = (double)rot[b * 9 + i];
  double sx = 0, sy = 0, sz = 0, tx = 0, ty = 0, tz = 0;
  if (t < P) {
    sx = cx[t]; sy = cy[t]; sz = cz[t];
    const double a = nx[t], bb = ny[t], c = nz[t];
    tx = (scale * R[0]) * a + (scale * R[1]) * bb + (scale * R[2]) * c;
    ty = (scale * R[3]) * a + (scale * R[4]) * bb + (scale * R[5]) * c;
    tz = (scale * R[6]) * a + (scale * R[7]) * bb + (scale * R[8]) * c;
    atomicMax((int*)&hmax[0], __float_as_int(fabsf(nx[t])));   // non-negative floats order like ints
    atomicMax((int*)&hmax[1], __float_as_int(fabsf(ny[t])));
    atomicMax((int*)&hmax[2], __float_as_int(fabsf(nz[t])));
  }
  const double mcx = block_sum(sx, red) / P, mcy = block_sum(sy, red) / P, mcz = block_sum(sz, red) / P;
  const double mtx = block_sum(tx, red) / P, mty = block_sum(ty, red) / P, mtz = block_sum(tz, red) / P;
  if (t == 0) {
    double tr[3] = {mcx - mtx, mcy - mty, mcz - mtz};
    // NaN in nocs makes np.max propagate NaN; atomicMax on bit patterns would hide it -> re-check via the sums
    bool nocs_nan = !(mtx == mtx) && (scale == scale);
    ts_out[b * 4 + 0] = tr[0]; ts_out[b * 4 + 1] = tr[1]; ts_out[b * 4 + 2] = tr[2]; ts_out[b * 4 + 3] = scale;
    // bbox corners (utils.py:49-56), sRT = [R | t] in float32 (interface_v5.py:358-361), then inv(E1)
    const double size[3] = {2.0 * (double)hmax[0] * scale, 2.0 * (double)hmax[1] * scale, 2.0 * (double)hmax[2] * scale};
    const float tf[3] = {(float)tr[0], (float)tr[1], (float)tr[2]};
    emit_bbox_world(b, R, tf, size, !nocs_nan, E1, bbox, valid);
  }
}

// One slice (blockIdx.x = pose * G + slice) of the fast path's three pair passes: STAGE 0 counts the first digit, STAGE 1 re-derives
// the selected first bucket from the pose's summed histogram and counts the second digit, STAGE 2 re-derives the whole plan and
// collects candidates / the count below the window.  The plan is the finishing kernel's arithmetic on the same counts.
template <int STAGE>
__global__ __launch_bounds__(PP_THREADS) void pp_split_kernel(const float* __restrict__ nocs, const float* __restrict__ depth,
                                                              const int* __restrict__ choose, const double* __restrict__ Kc,
                                                              PPScratch* __restrict__ scr, int P, int img, int G, int flags) {
  __shared__ double cx[PP_MAXP], cy[PP_MAXP], cz[PP_MAXP];
  __shared__ float nx[PP_MAXP], ny[PP_MAXP], nz[PP_MAXP];
  __shared__ unsigned hist[4096];
  __shared__ unsigned wsum[PP_THREADS / 64];
  __shared__ unsigned sres[4];
  __shared__ int sprev;
  __shared__ unsigned s_below;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G, t = threadIdx.x;
  const double fx = Kc[b * 9 + 0], fy = Kc[b * 9 + 4], pcx = Kc[b * 9 + 2], pcy = Kc[b * 9 + 5];
  if (t < P) {
    const int ch = choose[(long long)b * P + t];
    const double u = (double)(ch % img), v = (double)(ch / img);
    const double z = (double)depth[(long long)b * P + t];
    cx[t] = ((u - pcx) * z) / fx;
    cy[t] = ((v - pcy) * z) / fy;
    cz[t] = z;
    nx[t] = nocs[((long long)b * P + t) * 3 + 0];
    ny[t] = nocs[((long long)b * P + t) * 3 + 1];
    nz[t] = nocs[((long long)b * P + t) * 3 + 2];
  }
  if (t == 0) s_below = 0u;
  const PPArrays A{cx, cy, cz, nx, ny, nz};
  PPScratch& S = scr[b];
  PPPlan pl;
  pl.ok = true;
  if (STAGE >= 1) {
    for (int i = t; i < 4096; i += PP_THREADS) hist[i] = S.hist0[i];
    __syncthreads();
    const PPSel sa = pp_select(hist, wsum, sres, true, 0u);
    pp_plan_a(pl, sa);
    if (!pl.ok) return;                          // block-uniform: the finishing kernel takes the generic path (or reports no valid pair)
    if (STAGE >= 2) {
      for (int i = t; i < 4096; i += PP_THREADS) hist[i] = S.hist1[i];
      __syncthreads();
      const PPSel sb = pp_select(hist, wsum, sres, false, pl.rank);
      pp_plan_b(pl, hist, sb, &sprev, flags);
      if (!pl.ok) return;
    }
  }
  __syncthreads();
  for (int i = t; i < 4096; i += PP_THREADS) hist[i] = 0u;
  __syncthreads();
  // the pairs of this slice: row pair r, a 1 / (parts * G) share of its P - 1 partners (postprocess_kernel's map, cut G times finer)
  const int npair_rows = P / 2, per_row = P - 1;
  const int parts = (PP_THREADS / npair_rows) * G;
  const int r = t % npair_rows, part = (t / npair_rows) * G + g;
  if (t / npair_rows < PP_THREADS / npair_rows) {
    const int q_lo = (int)(((long long)per_row * part) / parts), q_hi = (int)(((long long)per_row * (part + 1)) / parts);
    if (STAGE == 0) {
      for_pairs_approx(A, P, r, q_lo, q_hi, [&](const PPPoint&, int, bool v, float a) {
          if (v) atomicAdd(&hist[pp_digit_a(a)], 1u);
      });
    } else if (STAGE == 1) {
      for_pairs_approx(A, P, r, q_lo, q_hi, [&](const PPPoint&, int, bool v, float a) {
        const unsigned fb = __float_as_uint(a);
        if (v && (fb >> 15) == pl.keyA) atomicAdd(&hist[(fb >> 3) & 4095u], 1u);
      });
    } else {
      const unsigned nb = pp_collect(pl, A, P, r, q_lo, q_hi,
                                     [&](unsigned long long key) { const unsigned i = atomicAdd(&S.ncand, 1u); if (i < (unsigned)PP_CAND) S.cand[i] = key; },
                                     [&](unsigned long long key) { atomicMax(&S.lowkey, key); });
      if (nb) atomicAdd(&s_below, nb);
    }
  }
  __syncthreads();
  if (STAGE < 2) {
    unsigned* gh = STAGE == 0 ? S.hist0 : S.hist1;
    for (int i = t; i < 4096; i += PP_THREADS) { const unsigned c = hist[i]; if (c) atomicAdd(&gh[i], c); }
  } else if (t == 0 && s_below) {
    atomicAdd(&S.below, s_below);
  }
}

size_t postprocess_scratch_bytes(int B) { return (size_t)B * sizeof(PPScratch); }

// slices per pose of the split form: fill the chip (256 CUs) without exceeding 32 slices; 1 = the one-kernel form
int postprocess_slices(int B) {
  int g = 1;
  while (g < 32 && (long long)B * g * 2 <= 256) g *= 2;
  return g;
}

int launch_postprocess(const float* nocs, const float* depth, const float* rot, const int* choose, const double* Kc,
                       const double* E1, double* bbox, double* ts_out, int* valid, int B, int P, int img, hipStream_t s,
                       void* scratch, size_t scratch_bytes) {
  RGBM_REQUIRE(P >= 2 && P <= PP_MAXP && (P % 2) == 0 && (PP_THREADS % (P / 2)) == 0, "postprocess needs even P<=1024 dividing 2048");
  // debug flags: 33554432 = generic (fp64 radix) selection only, 67108864 = guard band in every even-count pose (tests)
  const int flags = ((g_debug_flags & (1 << 25)) ? PP_F_GENERIC : 0) | ((g_debug_flags & (1 << 26)) ? PP_F_GUARD : 0);
  const int G = scratch && !(flags & PP_F_GENERIC) ? postprocess_slices(B) : 1;
  if (G > 1 && !(g_debug_flags & (1 << 23))) {        // debug flag 8388608: one-kernel form even when scratch is given (A/B)
    RGBM_REQUIRE(scratch_bytes >= postprocess_scratch_bytes(B) && ((uintptr_t)scratch & 7) == 0, "postprocess scratch too small or misaligned");
    PPScratch* scr = reinterpret_cast<PPScratch*>(scratch);
    RGBM_CHECK_HIP(hipMemsetAsync(scr, 0, postprocess_scratch_bytes(B), s));
    hipLaunchKernelGGL(pp_split_kernel<0>, dim3(B * G), dim3(PP_THREADS), 0, s, nocs, depth, choose, Kc, scr, P, img, G, flags);
    hipLaunchKernelGGL(pp_split_kernel<1>, dim3(B * G), dim3(PP_THREADS), 0, s, nocs, depth, choose, Kc, scr, P, img, G, flags);
    hipLaunchKernelGGL(pp_split_kernel<2>, dim3(B * G), dim3(PP_THREADS), 0, s, nocs, depth, choose, Kc, scr, P, img, G, flags);
    hipLaunchKernelGGL(postprocess_kernel<true>, dim3(B), dim3(PP_THREADS), 0, s, nocs, depth, rot, choose, Kc, E1, bbox, ts_out,
                       valid, P, img, (const PPScratch*)scr, flags);
  } else {
    hipLaunchKernelGGL(postprocess_kernel<false>, dim3(B), dim3(PP_THREADS), 0, s, nocs, depth, rot, choose, Kc, E1, bbox, ts_out,
                       valid, P, img, (const PPScratch*)nullptr, flags);
  }
  RGBM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace rgbm
