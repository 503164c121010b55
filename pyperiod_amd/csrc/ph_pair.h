// Window-pair screen of the all-p sweep (gfx950 / CDNA4 only).
//
// The norm sweeps of ph_device.h are VALU-issue bound: per ds_read_b64 the fold spends one v_add_f64 plus
// 2-3 instructions of addressing, masking and reduction.  Here TWO windows share a workgroup and every LDS
// element is the pair {fl32(a[n] * sa), fl32(b[n] * sb)} of their float-rounded (power-of-two scaled) samples:
// one ds_read_b64 brings a sample of both windows, v_pk_add_f32 / v_pk_fma_f32 fold both, and every address,
// mask, weight and cross-lane instruction of a pass is shared between them -- the instruction stream of one
// fp64 pass now serves two windows.  Always 8-byte aligned, no shifted copies, same register footprint as the
// fp64 fold (a float pair is as wide as a double).  The window is followed by kPad ZEROED pairs: the chunk that holds
// the cut of a period reads the missing last sample of its short residues from there (pair_rows_group, STR).
//
// The float values only SCREEN the candidates of an m_best iteration (Periods.py:501-515): a rigorous radius
// (pair_radius) bounds |screen - exact|, and the periods whose upper bound reaches the best lower bound are
// re-evaluated in fp64 by the kernel (k_mbest_step1_pair in ph_kernels.h), which decides on those values.
#pragma once

#include "ph_device.h"

namespace ph {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) f2* pair_ptr;

// float geometry of a period for the screen (host table next to PGeom)
struct PGeomF {
  int rows;       // R = ceil(N / p)
  int nfull;      // residues j < nfull own R samples, the others R-1
  float w_full;   // fl32(1 / R)
  float w_short;  // fl32(1 / (R-1))  (0 when R == 1)
};

__device__ __forceinline__ f2 f2_make(float a, float b) {
  f2 r;
  r.x = a;
  r.y = b;
  return r;
}
__device__ __forceinline__ f2 f2_zero() { return f2_make(0.0f, 0.0f); }
__device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 f2_max(f2 a, f2 b) { return __builtin_elementwise_max(a, b); }
// accumulate one residue sum t: sum of squares (norm screens) or largest square (MX: the max |S| screen of
// best_correlation, Periods.py:327-331)
template <bool MX>
__device__ __forceinline__ f2 f2_acc(f2 part, f2 t) {
  return MX ? f2_max(part, t * t) : f2_fma(t, t, part);
}

template <int CTRL>
__device__ __forceinline__ f2 dpp_f2(f2 v) {
  const int a = __builtin_amdgcn_update_dpp(0, __float_as_int(v.x), CTRL, 0xF, 0xF, false);
  const int b = __builtin_amdgcn_update_dpp(0, __float_as_int(v.y), CTRL, 0xF, 0xF, false);
  return f2_make(__int_as_float(a), __int_as_float(b));
}

// Lane index, recomputed where it is needed (two VALU) instead of kept live across the folds: as a long-lived
// value it was spilled, and every segment of every pass started with a scratch reload in front of its first LDS
// address.  The volatile asm keeps the compiler from hoisting it back into one register.
__device__ __forceinline__ int pair_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// ---------------------------------------------------------------- few-row single passes (R <= 6)
// STR (with MASK): the last chunk STRADDLES the cut of the period -- its lanes below `nvalid` are residues with NR
// samples, the lanes from there up to `nvalid2` residues with NR - 1 whose last-row load falls into the zeroed pad
// behind the window (index (NR-1) p + j >= N exactly for those j, and < N + 64): one chunk serves both count classes,
// and the part with the short residues starts on a chunk boundary.  Their squares go to `part2`.
template <int NR, int C, bool MASK, bool MX, bool STR = false>
__device__ __forceinline__ void pair_rows_group(pair_ptr ptr, int p, int nvalid, int nvalid2, int lane, f2& part, f2& part2) {
  f2 v[NR][C];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < C; ++c) v[r][c] = ptr[r * p + 64 * c];
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    f2 t = v[0][c];
#pragma unroll
    for (int r = 1; r < NR; ++r) t += v[r][c];
    if (MASK && c == C - 1) {  // a masked group is cut in its LAST chunk only
      const int col = 64 * c + lane;
      if (STR) {
        const f2 t2 = (col >= nvalid && col < nvalid2) ? t : f2_zero();
        part2 = f2_acc<MX>(part2, t2);
      }
      t = (col < nvalid) ? t : f2_zero();
    }
    part = f2_acc<MX>(part, t);
  }
}
template <int NR, int C, bool MASK, bool MX>
__device__ __forceinline__ void pair_rows_group(pair_ptr ptr, int p, int nvalid, int lane, f2& part) {
  pair_rows_group<NR, C, MASK, MX, false>(ptr, p, nvalid, nvalid, lane, part, part);
}

// ---------------------------------------------------------------- general segmented group (see seg_group)
// STR: as in pair_rows_group -- the lanes of the last chunk from `nvalid` up to `nvalid2` are residues with one sample
// fewer (their last row reads the zeroed pad); their squares take the weights `wgt2` and go to `part2`.
template <int M, int U, int C, bool MASK, bool MX, bool STR = false>
__device__ __forceinline__ void pair_seg_group(pair_ptr ptr, int p, int nrows, int nvalid, int nvalid2, int lane,
                                               const float (&wgt)[7], const float (&wgt2)[7], f2 (&part)[3], f2 (&part2)[3]) {
  static_assert(U % M == 0, "a row block must cover whole class cycles");
  f2 a[M][C];
  int r = 0;
  if (nrows >= U) {  // the first row block initialises the class sums (no zeroing, no adds)
    f2 v[U][C];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) v[u][c] = ptr[u * p + 64 * c];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < M; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) {
        a[u][c] = v[u][c];
#pragma unroll
        for (int w = u + M; w < U; w += M) a[u][c] += v[w][c];
      }
    ptr += U * p;
    r = U;
  } else {
#pragma unroll
    for (int u = 0; u < M; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) a[u][c] = f2_zero();
  }
  for (; r + U <= nrows; r += U) {
    f2 v[U][C];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) v[u][c] = ptr[u * p + 64 * c];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) a[u % M][c] += v[u][c];
    ptr += U * p;
  }
  if (U > 1) {
    const int rem = nrows - r;
#pragma unroll
    for (int u = 0; u < U - 1; ++u) {
      if (u < rem) {
        asm volatile("" ::: "memory");  // keep the wave-uniform branch (see seg_group)
        f2 v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = ptr[u * p + 64 * c];
        __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int c = 0; c < C; ++c) a[u % M][c] += v[c];
      }
    }
  }
  // squares of the class sums of chunk c into `pt` with the weights `w`
  auto fin = [&](int c, const float (&w)[7], f2 (&pt)[3]) {
    if (M == 1) {
      pt[0] = f2_acc<MX>(pt[0], a[0][c]);
    } else if (MX) {  // max |S| of p, 2p and (M == 4) 4p from the class sums; no count weights
      const f2 e = M == 4 ? a[0][c] + a[2 % M][c] : a[0][c], o = M == 4 ? a[1 % M][c] + a[3 % M][c] : a[1 % M][c];
      pt[0] = f2_acc<true>(pt[0], e + o);
      pt[1] = f2_acc<true>(f2_acc<true>(pt[1], e), o);
      if (M == 4) {
#pragma unroll
        for (int u = 0; u < M; ++u) pt[2] = f2_acc<true>(pt[2], a[u][c]);
      }
    } else if (M == 2) {
      const f2 e = a[0][c], o = a[1 % M][c], t = e + o;
      pt[0] = f2_fma(t, t, pt[0]);
      pt[1] = f2_fma(e, e, pt[1]);
      pt[2] = f2_fma(o, o, pt[2]);
    } else {
      const f2 e = a[0][c] + a[2 % M][c], o = a[1 % M][c] + a[3 % M][c], t = e + o;
      pt[0] = f2_fma(t, t * w[0], pt[0]);
      pt[1] = f2_fma(e, e * w[1], pt[1]);
      pt[1] = f2_fma(o, o * w[2], pt[1]);
#pragma unroll
      for (int u = 0; u < M; ++u) pt[2] = f2_fma(a[u][c], a[u][c] * w[3 + u], pt[2]);
    }
  };
#pragma unroll
  for (int c = 0; c < C; ++c) {
    if (MASK && c == C - 1) {
      // a masked group holds exactly the chunks that are left: only the last one is cut -- one class without a straddle
      // by zeroing its sum, otherwise by running the squares under the lane masks (two scalar instructions, no selects)
      const int col = 64 * c + lane;
      if (M == 1) {  // (selects, not branches: the compiler turns the choice between the two arrays into a scratch pointer)
        part[0] = f2_acc<MX>(part[0], (col < nvalid) ? a[0][c] : f2_zero());
        if (STR) part2[0] = f2_acc<MX>(part2[0], (col >= nvalid && col < nvalid2) ? a[0][c] : f2_zero());
      } else if (col < nvalid) {
        fin(c, wgt, part);
      } else if (STR && col < nvalid2) {
        fin(c, wgt2, part2);
      }
    } else {
      fin(c, wgt, part);
    }
  }
}
template <int M, int U, int C, bool MASK, bool MX>
__device__ __forceinline__ void pair_seg_group(pair_ptr ptr, int p, int nrows, int nvalid, int lane,
                                               const float (&wgt)[7], f2 (&part)[3]) {
  pair_seg_group<M, U, C, MASK, MX, false>(ptr, p, nrows, nvalid, nvalid, lane, wgt, wgt, part, part);
}

// Per-lane partials of sum_j S_q[j]^2 / cnt_q[j] of BOTH windows, base period p >= 64; segment logic of seg_group.
// The pass is dispatched once, on the row count of the period, to straight-line code for both segments (see "passes
// with one dispatch" in ph_device.h: one CU has one scalar unit, and the screens kept it ~90 % busy; measured with
// tools/micro/pair_pass_bench.hip: 129 -> 71 scalar instructions per few-row pass, 100 -> 85 ns per pass and CU).
// columns [0, len) from `base` (lane included), NR samples each; STR: `len` is the cut of the period, and the chunk
// that holds it also takes the residues behind it (up to column `cols`) into `part2`
template <int NR, bool MX, bool STR = false>
__device__ __forceinline__ void pair_single_rows(pair_ptr base, int p, int len, int cols, int lane, f2& part, f2& part2) {
  constexpr int CG = NR <= 4 ? 4 : 2;
  const int whole = len >> 6;
  int c = 0;
  for (; c + CG <= whole; c += CG) pair_rows_group<NR, CG, false, MX>(base + 64 * c, p, 0, lane, part);
  for (; c < whole; ++c) {
    asm volatile("" ::: "memory");
    pair_rows_group<NR, 1, false, MX>(base + 64 * c, p, 0, lane, part);
  }
  const int rem = len & 63;
  if (rem) {
    asm volatile("" ::: "memory");
    pair_rows_group<NR, 1, true, MX, STR>(base + 64 * whole, p, rem, cols - 64 * whole, lane, part, part2);
  }
}

#ifndef PH_PAIR_U1
#define PH_PAIR_U1 2  // rows per load block of the many-row single passes
#endif
// one part of a pass: columns [0, len) from `base` (lane included), nrows samples each, squares into `part` with the
// weights `wgt`.  STR: `len` is the cut of the period; the chunk that holds it also takes the residues behind it, up to
// column `cols`, with the weights `wgt2` into `part2`.
template <int M, int U, bool MX, bool STR = false>
__device__ __forceinline__ void pair_multi_segment(pair_ptr base, int p, int len, int cols, int nrows, int lane,
                                                   const float (&wgt)[7], const float (&wgt2)[7], f2 (&part)[3], f2 (&part2)[3]) {
  constexpr int CM = (M == 4) ? 2 : 4;
  const int whole = len >> 6;
  int c = 0;
  for (; c + CM <= whole; c += CM) pair_seg_group<M, U, CM, false, MX>(base + 64 * c, p, nrows, 64 * CM, lane, wgt, part);
  const int left = len - 64 * c;  // < 64 CM columns
  const int left2 = cols - 64 * c;
  const pair_ptr at = base + 64 * c;
  if (CM == 4 && left > 128) {
    if (left > 192) pair_seg_group<M, U, (CM == 4 ? 4 : 1), true, MX, STR>(at, p, nrows, left, left2, lane, wgt, wgt2, part, part2);
    else pair_seg_group<M, U, (CM == 4 ? 3 : 1), true, MX, STR>(at, p, nrows, left, left2, lane, wgt, wgt2, part, part2);
  } else if (left > 64) {
    pair_seg_group<M, U, 2, true, MX, STR>(at, p, nrows, left, left2, lane, wgt, wgt2, part, part2);
  } else if (left > 0) {
    pair_seg_group<M, U, 1, true, MX, STR>(at, p, nrows, left, left2, lane, wgt, wgt2, part, part2);
  }
}

// Per-lane partials of sum_j S_p[j]^2 / cnt_p[j] (MX: max_j S_p[j]^2) of both windows, p >= 64.  The residues below
// the cut c = nfull own `rows` samples, the others one fewer; part A is every chunk up to and including the one that
// holds c - 1 (its lanes behind the cut ride along: STR), part B starts on the next chunk boundary -- ceil(p / 64)
// chunk columns per pass instead of ceil(c / 64) + ceil((p - c) / 64).
template <bool MX = false>
__device__ __forceinline__ f2 pair_pass_single(const f2* __restrict__ xs, int p, const PGeomF g) {
  const int lane = pair_lane();
  const int cut = g.nfull;
  const int acols = min(p, (cut + 63) & ~63), rest = p - acols;
  const pair_ptr a = (pair_ptr)xs + lane, b = a + acols;
  f2 sa = f2_zero(), sb = f2_zero();
  switch (g.rows) {
#define PH_SINGLE_CASE(R)                                              \
  case R:                                                              \
    pair_single_rows<R, MX, true>(a, p, cut, acols, lane, sa, sb);     \
    pair_single_rows<R - 1, MX>(b, p, rest, rest, lane, sb, sb);       \
    break;
    PH_SINGLE_CASE(2)
    PH_SINGLE_CASE(3)
    PH_SINGLE_CASE(4)
    PH_SINGLE_CASE(5)
    PH_SINGLE_CASE(6)
#undef PH_SINGLE_CASE
    default: {
      const float wgt[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      f2 pa[3] = {f2_zero(), f2_zero(), f2_zero()}, pb[3] = {f2_zero(), f2_zero(), f2_zero()};
      pair_multi_segment<1, PH_PAIR_U1, MX, true>(a, p, cut, acols, g.rows, lane, wgt, wgt, pa, pb);
      if (rest > 0) pair_multi_segment<1, PH_PAIR_U1, MX>(b, p, rest, rest, g.rows - 1, lane, wgt, wgt, pb, pb);
      sa = pa[0];
      sb = pb[0];
      break;
    }
  }
  if (MX) return f2_max(sa, sb);
  return f2_fma(sa, f2_make(g.w_full, g.w_full), sb * g.w_short);
}

// ---------------------------------------------------------------- multi-class pass, straight-line segments
// Period p, 2p (and 4p) from the class sums of one fold: the weights of a part are compile-time indexed and chosen on
// the scalar unit, the tail of a part is picked by two compares, and no flag survives a branch.
// (x < y) ? a : b of wave-uniform values on the scalar unit (the compiler's version of a float select under a scalar
// condition is two v_mov and a v_cndmask)
__device__ __forceinline__ float scalar_select_lt(int x, int y, float a, float b) {
  float r;
  asm("s_cmp_lt_i32 %1, %2\n\ts_cselect_b32 %0, %3, %4" : "=s"(r) : "s"(x), "s"(y), "s"(a), "s"(b) : "scc");
  return r;
}

template <int M, bool MX = false>
__device__ __forceinline__ void pair_pass_multi(const f2* __restrict__ xs, int p, const PGeomF* __restrict__ geom,
                                                f2 (&total)[3]) {
  static_assert(M == 2 || M == 4, "two or four classes");
  const PGeomF g1 = geom[p], g2 = geom[2 * p], g4 = geom[(M == 4 ? 4 : 2) * p];
  const int lane = pair_lane();
  const int cut = g1.nfull;
  const int acols = min(p, (cut + 63) & ~63), rest = p - acols;  // part A: the chunks up to the one that holds the cut
  const pair_ptr a = (pair_ptr)xs + lane;
  // class u of period 2p / 4p at column j is the residue u p + j: the columns below the cut lie on one side of
  // nfull(2p), nfull(4p), the columns behind it on one side as well (weights wa / wb)
  float wa[7], wb[7];
  wa[0] = g1.w_full;
  wb[0] = g1.w_short;
  wa[1] = g2.w_full;  // residue 0 is never short
  wa[3] = g4.w_full;
  if (!MX) {
    wa[2] = scalar_select_lt(p, g2.nfull, g2.w_full, g2.w_short);
#pragma unroll
    for (int u = 0; u < 2; ++u) wb[1 + u] = scalar_select_lt(cut + u * p, g2.nfull, g2.w_full, g2.w_short);
#pragma unroll
    for (int u = 1; u < 4; ++u) wa[3 + u] = M == 4 ? scalar_select_lt(u * p, g4.nfull, g4.w_full, g4.w_short) : 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u) wb[3 + u] = M == 4 ? scalar_select_lt(cut + u * p, g4.nfull, g4.w_full, g4.w_short) : 0.0f;
  }
  if (M == 4) {  // the weights are applied per chunk, straight into the totals
    total[0] = total[1] = total[2] = f2_zero();
    pair_multi_segment<M, M, MX, true>(a, p, cut, acols, g1.rows, lane, wa, wb, total, total);
    if (rest > 0) pair_multi_segment<M, M, MX>(a + acols, p, rest, rest, g1.rows - 1, lane, wb, wb, total, total);
    return;
  }
  f2 sa[3] = {f2_zero(), f2_zero(), f2_zero()}, sb[3] = {f2_zero(), f2_zero(), f2_zero()};
  pair_multi_segment<M, M, MX, true>(a, p, cut, acols, g1.rows, lane, wa, wb, sa, sb);
  if (rest > 0) pair_multi_segment<M, M, MX>(a + acols, p, rest, rest, g1.rows - 1, lane, wb, wb, sb, sb);
  if (MX) {
    total[0] = f2_max(sa[0], sb[0]);
    total[1] = f2_max(sa[1], sb[1]);
  } else {
    total[0] = f2_fma(sa[0], f2_make(wa[0], wa[0]), sb[0] * wb[0]);
    total[1] = f2_fma(sa[1], f2_make(wa[1], wa[1]), f2_fma(sa[2], f2_make(wa[2], wa[2]), f2_fma(sb[1], f2_make(wb[1], wb[1]), sb[2] * wb[2])));
  }
  total[2] = f2_zero();
}

// p < 64: row-split path of wave_fold_small / wave_partial_small for pairs.
template <bool MX = false>
__device__ __forceinline__ f2 pair_partial_small(const f2* __restrict__ xs, int N, int p, const PGeomF& g) {
  const int lane = pair_lane();
  const int G = 64 / p;
  const int L = G * p;
  const int full = N / L;
  const bool on = lane < L;
  const f2* ptr = xs + (on ? lane : 0);
  f2 s0 = f2_zero(), s1 = f2_zero();
  int r = 0;
  for (; r + 4 <= full; r += 4) {
    const f2 a = ptr[0], b = ptr[L], c = ptr[2 * L], d = ptr[3 * L];
    s0 += a;
    s1 += b;
    s0 += c;
    s1 += d;
    ptr += 4 * L;
  }
  for (; r < full; ++r) {
    s0 += ptr[0];
    ptr += L;
  }
  const bool tail = on && (full * L + lane < N);
  const f2 tv = xs[tail ? full * L + lane : 0];
  f2 tot = s0 + s1 + (tail ? tv : f2_zero());
  tot = on ? tot : f2_zero();
  int s = 1;
  while (s < G) s <<= 1;
  for (s >>= 1; s >= 1; s >>= 1) {
    const int src = lane + s * p;
    const float ox = __shfl(tot.x, src & (kWave - 1), kWave);
    const float oy = __shfl(tot.y, src & (kWave - 1), kWave);
    tot += (src < L) ? f2_make(ox, oy) : f2_zero();
  }
  const float w = MX ? 1.0f : (lane < g.nfull) ? g.w_full : g.w_short;
  return (lane < p) ? tot * tot * w : f2_zero();
}

// Wavefront totals of the per-lane pairs, PACKED: the consumers of a screen store the totals or test them in one lane,
// nobody needs them in all 64.  v_permlane32_swap puts the two windows of a period side by side in one register (lanes
// 0-31 window a, 32-63 window b: half the data per level from there on), v_permlane16_swap does the same with two
// periods (rows of 16 lanes), and the rest of the tree runs inside the rows with the DPP operand folded into the add
// (ror 8, half mirror, xor 2, xor 1 -- nothing goes through the LDS pipe).  9 instructions for one period, 10 for two
// (the all-lanes versions of round 3: 20 and 22).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

template <bool MX>
__device__ __forceinline__ float pair_row_reduce(float z) {
  auto op = [](float u, float v) { return MX ? fmaxf(u, v) : u + v; };
  z = op(z, dpp_f32<kDppRor8>(z));
  z = op(z, dpp_f32<kDppHalfMirror>(z));
  z = op(z, dpp_f32<kDppXor2>(z));
  z = op(z, dpp_f32<kDppXor1>(z));
  return z;
}

// one period: afterwards lanes 0-31 hold the wavefront total (MX: maximum) of v.x, lanes 32-63 that of v.y
template <bool MX>
__device__ __forceinline__ float pair_reduce1(f2 v) {
  auto op = [](float u, float w) { return MX ? fmaxf(u, w) : u + w; };
  const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(v.x), (unsigned)__float_as_int(v.y), false, false);
  const float z = op(__int_as_float((int)r0[0]), __int_as_float((int)r0[1]));  // rows: x, x, y, y (two partials each)
  const unsigned zi = (unsigned)__float_as_int(z);
  const auto r1 = __builtin_amdgcn_permlane16_swap(zi, zi, false, false);
  return pair_row_reduce<MX>(op(__int_as_float((int)r1[0]), __int_as_float((int)r1[1])));
}

// two periods: afterwards the rows of 16 lanes hold the totals of a.x, a.y, b.x, b.y (lanes 0-15, 16-31, 32-47, 48-63)
template <bool MX>
__device__ __forceinline__ float pair_reduce2(f2 a, f2 b) {
  auto op = [](float u, float w) { return MX ? fmaxf(u, w) : u + w; };
  const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(a.x), (unsigned)__float_as_int(b.x), false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(a.y), (unsigned)__float_as_int(b.y), false, false);
  const float x = op(__int_as_float((int)r0[0]), __int_as_float((int)r0[1]));  // rows: a.x, a.x, b.x, b.x
  const float y = op(__int_as_float((int)r1[0]), __int_as_float((int)r1[1]));  // rows: a.y, a.y, b.y, b.y
  const auto r2 = __builtin_amdgcn_permlane16_swap((unsigned)__float_as_int(x), (unsigned)__float_as_int(y), false, false);
  return pair_row_reduce<MX>(op(__int_as_float((int)r2[0]), __int_as_float((int)r2[1])));  // rows: a.x, a.y, b.x, b.y
}

// where the packed totals go in an array of pairs `vals` (entry q - q0 = {window a, window b} of period q)
__device__ __forceinline__ void pair_store1(f2* vals, float z, int q, int q0) {
  const int l = pair_lane();
  if ((l & 31) == 0) reinterpret_cast<float*>(vals)[2 * (q - q0) + (l >> 5)] = z;
}
__device__ __forceinline__ void pair_store2(f2* vals, float z, int qa, int qb, int q0) {
  const int l = pair_lane();
  if ((l & 15) == 0) reinterpret_cast<float*>(vals)[2 * ((l < 32 ? qa : qb) - q0) + ((l >> 4) & 1)] = z;
}

// ---------------------------------------------------------------- chains of periods up to 64
// The row-split pass at L <= 64 (lane l < L adds x[l + n L]: the fold to period L, contiguous reads) is the fold of
// every divisor of L as well: S_{L/2}[l] = S_L[l] + S_L[l + L/2], and so on down the powers of two.  One pass of
// N / L loads yields L, L/2, ..., L / 2^(nlev-1) -- the host plans 32 chains for the 63 periods up to 64 (PassPlan
// m = 8 + nlev).  Two levels share one wavefront reduction (pair_reduce2).  Any summation order is covered by
// pair_radius.
__device__ __forceinline__ f2 pair_shift_down(f2 v, int lane, int by) {  // v of lane + by (lanes that matter: < 64 - by)
  const int addr = ((lane + by) & (kWave - 1)) << 2;
  return f2_make(__int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v.x))),
                 __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v.y))));
}

template <bool MX>
__device__ __forceinline__ f2 pair_chain_term(f2 tot, int lane, int q, const PGeomF& g) {
  const float w = MX ? 1.0f : (lane < g.nfull) ? g.w_full : g.w_short;
  return (lane < q) ? tot * tot * w : f2_zero();
}

template <bool MX, typename F, typename F2>
__device__ __forceinline__ void pair_chain_small(const f2* __restrict__ xs, int N, int L, int nlev,
                                                 const PGeomF* __restrict__ geom, F&& consume, F2&& consume2) {
  const int lane = pair_lane();
  PGeomF g = geom[L];
  const int full = g.nfull == L ? g.rows : g.rows - 1;  // floor(N / L)
  const bool on = lane < L;
  const f2* ptr = xs + (on ? lane : 0);
  f2 s0 = f2_zero(), s1 = f2_zero();
  int r = 0;
  for (; r + 4 <= full; r += 4) {
    const f2 a = ptr[0], b = ptr[L], c = ptr[2 * L], d = ptr[3 * L];
    s0 += a;
    s1 += b;
    s0 += c;
    s1 += d;
    ptr += 4 * L;
  }
  for (; r < full; ++r) {
    s0 += ptr[0];
    ptr += L;
  }
  const bool tail = on && (full * L + lane < N);
  const f2 tv = xs[tail ? full * L + lane : 0];
  f2 tot = s0 + s1 + (tail ? tv : f2_zero());
  tot = on ? tot : f2_zero();
  int q = L;
  for (int lev = 0; lev < nlev; lev += 2) {
    const f2 va = pair_chain_term<MX>(tot, lane, q, g);
    if (lev + 1 < nlev) {
      const int q2 = q >> 1;
      const PGeomF g2 = geom[q2];
      tot += pair_shift_down(tot, lane, q2);
      const f2 vb = pair_chain_term<MX>(tot, lane, q2, g2);
      consume2(pair_reduce2<MX>(va, vb), q, q2);
      if (lev + 2 < nlev) {
        q = q2 >> 1;
        g = geom[q];
        tot += pair_shift_down(tot, lane, q);
      }
    } else {
      consume(pair_reduce1<MX>(va), q);
    }
  }
}

// Screen sweep driven by the pass plan.  consume(z, q) gets the packed totals of one period (pair_reduce1: lanes 0-31
// window a, 32-63 window b), consume2(z, q_a, q_b) those of two periods of a multi-class pass or two levels of a chain
// (pair_reduce2: rows of 16 lanes = q_a window a, q_a window b, q_b window a, q_b window b); pair_store1 / pair_store2
// put them into an array.  Every period is reduced on its own: nothing is live across the folds (the online 8-period
// butterfly of the fp64 sweeps kept pending partials that were spilled in every pass: 3.03 -> 2.82 ms in round 3).
template <bool MX = false, typename F, typename F2>
__device__ __forceinline__ void pair_sweep_plan(const f2* __restrict__ xs, int N, const PGeomF* __restrict__ geom,
                                                const PassPlan* __restrict__ plan, int i_first, int i_end, int stride,
                                                F&& consume, F2&& consume2, int* __restrict__ queue = nullptr) {
  auto red = [](f2 v) { return pair_reduce1<MX>(v); };
  // `queue` (an LDS counter the caller has set to `stride`, the number of wavefronts): the passes are taken in plan
  // order by whichever wavefront is free -- the passes differ in cost and the sweep ends at a barrier.
  for (int i = i_first; i < i_end;) {
    int ticket = 0;
    if (queue && pair_lane() == 0) ticket = lds_ticket(queue);  // the next pass: claimed now, looked at after this one
    const int p = plan[i].p, m = plan[i].m;
    if (m >= 8) {
      pair_chain_small<MX>(xs, N, p, m - 8, geom, consume, consume2);
    } else if (m == 0) {
      consume(red(pair_partial_small<MX>(xs, N, p, geom[p])), p);
    } else if (m == 1) {
      consume(red(pair_pass_single<MX>(xs, p, geom[p])), p);
    } else if (m == 2) {
      f2 part[3];
      pair_pass_multi<2, MX>(xs, p, geom, part);
      consume2(pair_reduce2<MX>(part[0], part[1]), p, 2 * p);
    } else {
      f2 part[3];
      pair_pass_multi<4, MX>(xs, p, geom, part);
      consume2(pair_reduce2<MX>(part[0], part[1]), p, 2 * p);
      consume(red(part[2]), 4 * p);
    }
    i = queue ? lds_ticket_value(ticket) : i + stride;
  }
}

// Rigorous radius of the screen, in units of the (scaled) sum of squares ssq of the fp64 residual r:
//   |screen(q) - sum_j S_q[j]^2 / cnt_q[j]| <= pair_radius(q) * ssq.
// With u = 2^-24, rf = fl32(r s) (s a power of two, |rf - r s| <= u |r s|), n_j <= R terms per residue and
// A_j = sum |r s| over the coset: any summation order gives |S^_j - S_j| <= (u + gamma_{R-1}) A_j <= R u' A_j, so
// |S^_j^2 - S_j^2| / cnt_j <= (2 |S_j| + R u' A_j) R u' A_j / cnt_j <= 2 R u' (1 + R u') Q_j by Cauchy-Schwarz
// (A_j^2 <= cnt_j Q_j, Q_j = sum (r s)^2 over the coset, sum_j Q_j = ssq).  Squaring, weighting with fl32(1/cnt)
// and the class sums add <= 6 u per term; the positive sum over q/64 chunks, two segments and the butterfly adds
// gamma_{q/64 + 10}.  First order: (2 R + q/64 + 16) u; the factor 1.5 and the +32 cover the second-order terms
// (R u <= 2^-13 for R <= 2048), the error of the fp64 value itself (<= (2R + q/64 + 16) 2^-53) and the error of ssq.
// Underflow: the windows are scaled to RMS ~ 1, so denormal roundings (<= 2^-149 each) are far below the radius.
__device__ __forceinline__ double pair_radius(int rows, int q) {
  return 1.5 * (2.0 * (double)rows + (double)(q >> 6) + 32.0) * 5.9604644775390625e-08;
}

}  // namespace ph
