// step_boundary_x.hpp -- Planes mode: step boundary with the x passes of both transforms fused in.
// Part of the bchmc engine's kernel set; include through kernels.hpp (definition order matters).
#pragma once
#include "common.hpp"

#ifndef BCHMC_BX_SWIZZLE
#define BCHMC_BX_SWIZZLE 1
#endif
#ifndef BCHMC_BX_WAVES
#define BCHMC_BX_WAVES 4
#endif

namespace bchmc {

// ======================================================================================================
// Step boundary with the x passes of both transforms fused in ("planes" mode).
//
// A 3-D real transform is the batched 2-D transform of the (y, z) planes followed by complex FFTs along x, and the
// step boundary is element-wise in k-space.  So between the gather and the next scatter the pipeline can be
//   rocFFT 2-D R2C over the 3 n planes of V  ->  THIS kernel: [x-FFT of the three V^ columns, the boundary
//   arithmetic of k_step_boundary, inverse x-FFT of the three Psi^ columns]  ->  rocFFT 2-D C2R over the planes,
// which saves two of the six strided passes over the 3-component arrays (measured: 2-D batched transforms
// 0.36 + 0.33 ms against 0.58 + 0.55 ms for the 3-D ones at 256^3 fp64, scripts/fft2d_bench.hip).
//
// One workgroup owns the x-columns of KB = 128 B / sizeof(complex) adjacent k at one j: n x KB elements,
// staged in LDS, four transforms per workgroup (V_x and ky V_y + kz V_z forward; kx B and B inverse, see below); radix-4 decimation-in-time FFTs in place (bit-reversed fill, twiddles from a table).
// FFT arithmetic in T (like rocFFT's plan precision), boundary arithmetic in double (like every k-space kernel).
// Requires n a power of two with n == PER * NT / KB, and nhp a multiple of KB.
// ======================================================================================================
// Every array of this kernel is touched once per step and none fits a cache, so its loads and stores carry the
// streaming (`nt`) hint: 0 = off, 1 = stores, 2 = loads and stores.  Same box, 4 alternating runs each at 256^3: kernel
// 0.354-0.387 -> 0.335 ms and the whole step 310.9 -> 317.7 steps/s (the L2 keeps what the neighbours re-read).
#ifndef BCHMC_BX_NT
#define BCHMC_BX_NT 2
#endif
typedef double bx_dv2 __attribute__((ext_vector_type(2)));
typedef float bx_fv2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bx_store(double2 *p, const double2 v) {
  if (BCHMC_BX_NT) {
    bx_dv2 t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<bx_dv2 *>(p));
  } else {
    *p = v;
  }
}
__device__ __forceinline__ void bx_store(float2 *p, const float2 v) {
  if (BCHMC_BX_NT) {
    bx_fv2 t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<bx_fv2 *>(p));
  } else {
    *p = v;
  }
}

__device__ __forceinline__ double2 bx_load(const double2 *p) {
  if (BCHMC_BX_NT >= 2) {
    const bx_dv2 t = __builtin_nontemporal_load(reinterpret_cast<const bx_dv2 *>(p));
    return make_double2(t.x, t.y);
  }
  return *p;
}
__device__ __forceinline__ float2 bx_load(const float2 *p) {
  if (BCHMC_BX_NT >= 2) {
    const bx_fv2 t = __builtin_nontemporal_load(reinterpret_cast<const bx_fv2 *>(p));
    return make_float2(t.x, t.y);
  }
  return *p;
}
__device__ __forceinline__ double bx_load(const double *p) {
  return BCHMC_BX_NT >= 2 ? __builtin_nontemporal_load(p) : *p;
}

template <typename T>
__device__ __forceinline__ C2<T> cmul(const C2<T> a, const C2<T> b) {
  C2<T> r;
  r.x = a.x * b.x - a.y * b.y;
  r.y = a.x * b.y + a.y * b.x;
  return r;
}

// In-place decimation-in-time FFT of KB interleaved columns: s[i * KB + c], bit-reversed input order on entry,
// natural order on exit.  Two radix-2 stages are fused into one radix-4 pass (4 LDS reads + 4 writes per 4 points
// per two stages); an odd log2 n gets one plain radix-2 stage first.  tw[r] = exp(-2 pi i r / n), r < n / 2.
template <typename T>
__device__ __forceinline__ void xfft_inplace(C2<T> *__restrict__ s, const C2<T> *__restrict__ tw, int n, int log2n,
                                             int KB, bool inverse) {
  int st = 1;
  if (log2n & 1) {  // stage 1: half = 1, twiddle 1
    const int nb = (n >> 1) * KB;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
      const int c = b % KB, i0 = (b / KB) << 1;
      const C2<T> a = s[i0 * KB + c], x = s[(i0 + 1) * KB + c];
      C2<T> o0, o1;
      o0.x = a.x + x.x; o0.y = a.y + x.y;
      o1.x = a.x - x.x; o1.y = a.y - x.y;
      s[i0 * KB + c] = o0;
      s[(i0 + 1) * KB + c] = o1;
    }
    __syncthreads();
    st = 2;
  }
  const int nq = (n >> 2) * KB;
  for (; st < log2n; st += 2) {  // stages st and st + 1
    const int half = 1 << (st - 1);
    const int t1 = n >> st, t2 = n >> (st + 1);  // twiddle strides of the two stages
    for (int b = threadIdx.x; b < nq; b += blockDim.x) {
      const int c = b % KB, bf = b / KB;
      const int r = bf & (half - 1), grp = bf >> (st - 1);
      const int j = (grp << (st + 1)) + r;
      C2<T> w1 = tw[r * t1], w2 = tw[r * t2];
      if (inverse) {
        w1.y = -w1.y;
        w2.y = -w2.y;
      }
      const C2<T> e0 = s[j * KB + c], e1 = s[(j + half) * KB + c], e2 = s[(j + 2 * half) * KB + c],
                  e3 = s[(j + 3 * half) * KB + c];
      const C2<T> m1 = cmul<T>(w1, e1), m3 = cmul<T>(w1, e3);
      C2<T> a0, a1, a2, a3;
      a0.x = e0.x + m1.x; a0.y = e0.y + m1.y;
      a1.x = e0.x - m1.x; a1.y = e0.y - m1.y;
      a2.x = e2.x + m3.x; a2.y = e2.y + m3.y;
      a3.x = e2.x - m3.x; a3.y = e2.y - m3.y;
      const C2<T> n2 = cmul<T>(w2, a2), n3 = cmul<T>(w2, a3);
      // second-stage twiddle of the odd pair is w2 * exp(-+ i pi / 2): multiply by -i (forward) / +i (inverse)
      C2<T> r3;
      if (inverse) {
        r3.x = -n3.y; r3.y = n3.x;
      } else {
        r3.x = n3.y; r3.y = -n3.x;
      }
      C2<T> o0, o1, o2, o3;
      o0.x = a0.x + n2.x; o0.y = a0.y + n2.y;
      o2.x = a0.x - n2.x; o2.y = a0.y - n2.y;
      o1.x = a1.x + r3.x; o1.y = a1.y + r3.y;
      o3.x = a1.x - r3.x; o3.y = a1.y - r3.y;
      s[j * KB + c] = o0;
      s[(j + half) * KB + c] = o1;
      s[(j + 2 * half) * KB + c] = o2;
      s[(j + 3 * half) * KB + c] = o3;
    }
    __syncthreads();
  }
}

// MODE selects which half of the boundary exists, so that the first and the last step of a trajectory (and the force
// evaluation before the first step) run on the 2-D plans as well:
//   BX_INTERIOR  forward x passes of V^, g^, both half kicks, drift, Psi^' with its inverse x passes (described above);
//   BX_FIRST     no V^ yet: g^ is read from g_in (k_kick_drift_za: one half kick, drift), Psi^' with inverse x passes;
//                g_in == nullptr: Psi^ of q_in only, nothing else read or written (launch_za), no trajectory control;
//   BX_LAST      forward x passes of V^, g^ stored to g_out, p_out = p - (eps/2) g^ (k_step_boundary<LAST>);
//                p_out == nullptr: g^ only (k_assemble<false>), no trajectory control.
enum { BX_INTERIOR = 0, BX_FIRST = 1, BX_LAST = 2 };

// ALPT = true changes what leaves through the two inverse x passes: instead of the Zel'dovich Psi^ (kx B and B, see
// below) the two fields Lag2Eul_non_zeldovich starts from -- delta(1)^ = c_za q^' into component 0 and the Poisson
// solution Phi^ = -delta(1)^ / k^2 (0 at k = 0; PoissonSolver, EqSolvers.cc:29-64: no Nyquist zeroing) into component 1
// of Ck, with c_za = deltaQ_factor / N.  The ALPT model's own pipeline (alpt_x.hpp) takes over from there.
template <typename T, int NT, int PER, int MODE = BX_INTERIOR, bool ALPT = false>
__global__ void __launch_bounds__(NT, BCHMC_BX_WAVES)
k_step_boundary_x(Geo g, int log2n, const C2<T> *__restrict__ twiddle, C2<T> *Ck, const C2<T> *q_in, const C2<T> *p_in,
                  C2<T> *q_out, C2<T> *p_out, const double *__restrict__ wS, const double *__restrict__ wM, double a,
                  double b, double half_eps, double eps, double c_za, double *guard_slot, StepCtl ctl,
                  const C2<T> *g_in = nullptr, C2<T> *g_out = nullptr) {
  constexpr int KB = 128 / (int)sizeof(C2<T>);
  constexpr int kMaxPer = PER;  // elements of one component per thread: n == PER * NT / KB (checked by the host)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_x[];
  __shared__ double red[NT / 64];
  const bool controlled = MODE == BX_INTERIOR || (MODE == BX_FIRST && g_in) || (MODE == BX_LAST && p_out);
  if (controlled) {
    if (*ctl.stop) return;
    if (ctl.guard_prev && fabs(*ctl.guard_prev) > ctl.guard_limit) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        *ctl.steps_done = ctl.step_index;
        __threadfence();
        *ctl.stop = 1;
      }
      return;
    }
  }
  const int n = g.n;
  C2<T> *s = reinterpret_cast<C2<T> *>(s_raw_x);  // n * KB
  C2<T> *tw = s + (size_t)n * KB;                 // n / 2
  for (int t = threadIdx.x; t < n / 2; t += blockDim.x) tw[t] = twiddle[t];
  const int ntk = g.nhp / KB;
  int bid = (int)blockIdx.x;
#if BCHMC_BX_SWIZZLE
  // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous range of (j, k0) columns
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
#endif
  const int j = bid / ntk, k0 = (bid % ntk) * KB;
  const int c = threadIdx.x % KB, irow = threadIdx.x / KB;
  constexpr int rows = NT / KB, per = PER;
  const int k = k0 + c;
  const long long plane = (long long)g.n * g.nhp;  // elements between consecutive i
  const long long col = k + (long long)g.nhp * j;  // element (0, j, k)
  const double ky = kval(j, g.n, g.kfac), kz = kval(k, g.n, g.kfac);
  const int shift = 32 - log2n;
  double2 hk[kMaxPer];
  // ---- forward x passes.  ky and kz are constants of a column, so V^_y and V^_z go through ONE transform as
  // ky V_y + kz V_z:  h^ = (1/k^2) [ kx (Im V^_x, -Re V^_x) + (Im W^, -Re W^) ],  W = ky V_y + kz V_z ----
  for (int pass = 0; pass < (MODE == BX_FIRST ? 0 : 2); pass++) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kMaxPer; m++) {
      const int i = irow + rows * m;
      const long long e = col + plane * i;
      C2<T> v;
      if (pass == 0) {
        v = bx_load(Ck + e);
      } else {
        const C2<T> vy = bx_load(Ck + e + g.Nhp), vz = bx_load(Ck + e + 2 * g.Nhp);
        v.x = (T)(ky * (double)vy.x + kz * (double)vz.x);
        v.y = (T)(ky * (double)vy.y + kz * (double)vz.y);
      }
      s[(int)(__brev((unsigned)i) >> shift) * KB + c] = v;
    }
    __syncthreads();
    xfft_inplace<T>(s, tw, n, log2n, KB, false);
#pragma unroll
    for (int m = 0; m < kMaxPer; m++) {
      const int i = irow + rows * m;
      const C2<T> v = s[i * KB + c];
      if (pass == 0) {
        const double kx = kval(i, g.n, g.kfac);
        hk[m] = make_double2(kx * (double)v.y, -(kx * (double)v.x));
      } else {
        hk[m].x += (double)v.y;
        hk[m].y -= (double)v.x;
      }
    }
  }
  // ---- boundary arithmetic (as k_step_boundary); the first inverse transform's input is filled on the way ----
  // Psi^_j = k_j B with B = (1/k^2)(Im phi^, -Re phi^), phi^ = c_za q^': kx B is transformed on its own, B once for
  // both the y and the z component (ky, kz are constants of the column again).
  double gsum = 0.;
  C2<T> qkeep[kMaxPer];  // q' as stored (rounded to T), for the second inverse pass: registers are free up to 128 here
  __syncthreads();
#pragma unroll
  for (int m = 0; m < kMaxPer; m++) {
    const int i = irow + rows * m;
    const long long idx = col + plane * i;
    const double kx = kval(i, g.n, g.kfac);
    const double ksq = kx * kx + ky * ky + kz * kz;
    const bool nyq = (i == g.n / 2) || (j == g.n / 2) || (k == g.n / 2);
    double2 q;
    if (MODE == BX_INTERIOR) {
      const C2<T> t = bx_load(q_in + idx);
      q = make_double2((double)t.x, (double)t.y);
    } else q = ld2<T>(q_in, idx);
    if (MODE == BX_FIRST) {
      if (g_in) {  // k_kick_drift_za<DRIFT>: the drift uses the kicked momentum before it is rounded to T
        double2 p = ld2<T>(p_in, idx);
        const double2 g0 = ld2<T>(g_in, idx);
        p.x -= half_eps * g0.x;
        p.y -= half_eps * g0.y;
        st2<T>(p_out, idx, p.x, p.y);
        if (wM) {
          const double w = wM[idx];
          q.x += eps * (w * p.x);
          q.y += eps * (w * p.y);
        }
        st2<T>(q_out, idx, q.x, q.y);
      }
    } else {
      double2 gg = make_double2(0., 0.);
      if (ksq > 0 && !nyq) {
        const double f = b * (1 / ksq);
        gg = make_double2(f * hk[m].x, f * hk[m].y);
      }
      if (a != 0.) {
        const double w = a * bx_load(wS + idx);
        gg.x += w * q.x;
        gg.y += w * q.y;
      }
      C2<T> gs;
      gs.x = (T)gg.x;
      gs.y = (T)gg.y;
      if (MODE == BX_LAST) {
        g_out[idx] = gs;
        if (p_out) {
          const double2 p = ld2<T>(p_in, idx);
          C2<T> pe;
          pe.x = (T)(p.x - half_eps * gg.x);
          pe.y = (T)(p.y - half_eps * gg.y);
          const double hw = (k == 0 || ((g.n & 1) == 0 && k == g.n / 2)) ? 1. : 2.;
          if (k < g.nh) gsum += hw * (double)pe.x;
          p_out[idx] = pe;
        }
        continue;
      }
      const C2<T> pt = bx_load(p_in + idx);
      double2 p = make_double2((double)pt.x, (double)pt.y);
      C2<T> pe;
      pe.x = (T)(p.x - half_eps * gg.x);
      pe.y = (T)(p.y - half_eps * gg.y);
      const double hw = (k == 0 || ((g.n & 1) == 0 && k == g.n / 2)) ? 1. : 2.;
      if (k < g.nh) gsum += hw * (double)pe.x;
      p.x = (double)pe.x - half_eps * (double)gs.x;
      p.y = (double)pe.y - half_eps * (double)gs.y;
      {
        C2<T> t;
        t.x = (T)p.x;
        t.y = (T)p.y;
        bx_store(p_out + idx, t);
      }
      if (wM) {
        const double w = bx_load(wM + idx);
        q.x += eps * (w * p.x);
        q.y += eps * (w * p.y);
      }
      {
        C2<T> t;
        t.x = (T)q.x;
        t.y = (T)q.y;
        bx_store(q_out + idx, t);
      }
    }
    qkeep[m].x = (T)q.x;
    qkeep[m].y = (T)q.y;
    C2<T> o;
    o.x = T(0);
    o.y = T(0);
    if (ALPT) {
      o.x = (T)(c_za * q.x);
      o.y = (T)(c_za * q.y);
    } else if (ksq > 1.e-14 && !nyq) {
      const double f = (1. / ksq) * kx;
      o.x = (T)(f * (c_za * q.y));
      o.y = (T)(f * -(c_za * q.x));
    }
    s[(int)(__brev((unsigned)i) >> shift) * KB + c] = o;
  }
  if (MODE == BX_LAST) {
    if (p_out) {
      gsum = block_sum(gsum, red);
      if (threadIdx.x == 0) atomic_add_r(guard_slot, gsum);
    }
    return;
  }
  __syncthreads();
  xfft_inplace<T>(s, tw, n, log2n, KB, true);
  for (int m = 0; m < per; m++) {
    const int i = irow + rows * m;
    bx_store(Ck + col + plane * i, s[i * KB + c]);
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < kMaxPer; m++) {
    const int i = irow + rows * m;
    const double kx = kval(i, g.n, g.kfac);
    const double ksq = kx * kx + ky * ky + kz * kz;
    const bool nyq = (i == g.n / 2) || (j == g.n / 2) || (k == g.n / 2);
    C2<T> o;
    o.x = T(0);
    o.y = T(0);
    if (ALPT) {
      const double f = (ksq > 0.) ? -1. / ksq : 0.;
      o.x = (T)(f * (c_za * (double)qkeep[m].x));
      o.y = (T)(f * (c_za * (double)qkeep[m].y));
    } else if (ksq > 1.e-14 && !nyq) {
      const double2 qn = make_double2((double)qkeep[m].x, (double)qkeep[m].y);
      const double f = 1. / ksq;
      o.x = (T)(f * (c_za * qn.y));
      o.y = (T)(f * -(c_za * qn.x));
    }
    s[(int)(__brev((unsigned)i) >> shift) * KB + c] = o;
  }
  __syncthreads();
  xfft_inplace<T>(s, tw, n, log2n, KB, true);
  for (int m = 0; m < per; m++) {
    const int i = irow + rows * m;
    const C2<T> v = s[i * KB + c];
    if (ALPT) {
      bx_store(Ck + col + plane * i + g.Nhp, v);
      continue;
    }
    C2<T> oy, oz;
    oy.x = (T)(ky * (double)v.x);
    oy.y = (T)(ky * (double)v.y);
    oz.x = (T)(kz * (double)v.x);
    oz.y = (T)(kz * (double)v.y);
    bx_store(Ck + col + plane * i + g.Nhp, oy);
    bx_store(Ck + col + plane * i + 2 * g.Nhp, oz);
  }
  if (MODE == BX_INTERIOR) {
    gsum = block_sum(gsum, red);
    if (threadIdx.x == 0) atomic_add_r(guard_slot, gsum);
  }
}


// ------------------------------------------------------------------------------------------------------
// Interior boundary, second formulation (round 3): both forward transforms at once and both inverse transforms at
// once on TWO LDS tiles, with every operand of a phase requested before the transforms of the previous phase run.
// The first formulation alternates load -> transform -> load -> transform -> load -> arithmetic with one 32 KB tile:
// its memory phases are exposed three times per workgroup and at most 32-64 KB are in flight per workgroup (rocFFT's
// column kernel keeps about twice as much in flight on the same 128-byte pattern, profiles/r02 counters).  Here a
// workgroup requests V^ (96 KB) at once, fills both tiles, requests q^, p^, wS, wM (96 KB) BEFORE the forward
// transforms, and the 8 (instead of 16) barrier-separated stages of the two transform pairs hide that latency.  No
// q' copy is kept in registers: both inverse inputs (kx B and B) are filled in the same pass.  Same arithmetic per
// element as k_step_boundary_x<BX_INTERIOR>.
// ------------------------------------------------------------------------------------------------------
template <typename T, bool INV>
__device__ __forceinline__ void xfft_r4(C2<T> *__restrict__ s, int j, int half, int KB, int c, C2<T> w1, C2<T> w2) {
  const C2<T> e0 = s[j * KB + c], e1 = s[(j + half) * KB + c], e2 = s[(j + 2 * half) * KB + c],
              e3 = s[(j + 3 * half) * KB + c];
  const C2<T> m1 = cmul<T>(w1, e1), m3 = cmul<T>(w1, e3);
  C2<T> a0, a1, a2, a3;
  a0.x = e0.x + m1.x; a0.y = e0.y + m1.y;
  a1.x = e0.x - m1.x; a1.y = e0.y - m1.y;
  a2.x = e2.x + m3.x; a2.y = e2.y + m3.y;
  a3.x = e2.x - m3.x; a3.y = e2.y - m3.y;
  const C2<T> n2 = cmul<T>(w2, a2), n3 = cmul<T>(w2, a3);
  C2<T> r3;
  if (INV) {
    r3.x = -n3.y; r3.y = n3.x;
  } else {
    r3.x = n3.y; r3.y = -n3.x;
  }
  C2<T> o0, o1, o2, o3;
  o0.x = a0.x + n2.x; o0.y = a0.y + n2.y;
  o2.x = a0.x - n2.x; o2.y = a0.y - n2.y;
  o1.x = a1.x + r3.x; o1.y = a1.y + r3.y;
  o3.x = a1.x - r3.x; o3.y = a1.y - r3.y;
  s[j * KB + c] = o0;
  s[(j + half) * KB + c] = o1;
  s[(j + 2 * half) * KB + c] = o2;
  s[(j + 3 * half) * KB + c] = o3;
}

// xfft_inplace on two tiles with shared twiddles and shared barriers
template <typename T, bool INV>
__device__ __forceinline__ void xfft_inplace2(C2<T> *__restrict__ sa, C2<T> *__restrict__ sb,
                                              const C2<T> *__restrict__ tw, int n, int log2n, int KB) {
  int st = 1;
  if (log2n & 1) {
    const int nb = (n >> 1) * KB;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
      const int c = b % KB, i0 = (b / KB) << 1;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        C2<T> *s = u ? sb : sa;
        const C2<T> a = s[i0 * KB + c], x = s[(i0 + 1) * KB + c];
        C2<T> o0, o1;
        o0.x = a.x + x.x; o0.y = a.y + x.y;
        o1.x = a.x - x.x; o1.y = a.y - x.y;
        s[i0 * KB + c] = o0;
        s[(i0 + 1) * KB + c] = o1;
      }
    }
    __syncthreads();
    st = 2;
  }
  const int nq = (n >> 2) * KB;
  for (; st < log2n; st += 2) {
    const int half = 1 << (st - 1);
    const int t1 = n >> st, t2 = n >> (st + 1);
    for (int b = threadIdx.x; b < nq; b += blockDim.x) {
      const int c = b % KB, bf = b / KB;
      const int r = bf & (half - 1), grp = bf >> (st - 1);
      const int j = (grp << (st + 1)) + r;
      C2<T> w1 = tw[r * t1], w2 = tw[r * t2];
      if (INV) {
        w1.y = -w1.y;
        w2.y = -w2.y;
      }
      xfft_r4<T, INV>(sa, j, half, KB, c, w1, w2);
      xfft_r4<T, INV>(sb, j, half, KB, c, w1, w2);
    }
    __syncthreads();
  }
}

template <typename T, int NT, int PER>  // PER <= 4: eight elements per thread do not fit 128 registers with their operands
__global__ void __launch_bounds__(NT, BCHMC_BX_WAVES)
k_step_boundary_x2(Geo g, int log2n, const C2<T> *__restrict__ twiddle, C2<T> *Ck, const C2<T> *q_in, const C2<T> *p_in,
                   C2<T> *q_out, C2<T> *p_out, const double *__restrict__ wS, const double *__restrict__ wM, double a,
                   double b, double half_eps, double eps, double c_za, double *guard_slot, StepCtl ctl) {
  constexpr int KB = 128 / (int)sizeof(C2<T>);
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_x2[];
  __shared__ double red[NT / 64];
  if (*ctl.stop) return;
  if (ctl.guard_prev && fabs(*ctl.guard_prev) > ctl.guard_limit) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      *ctl.steps_done = ctl.step_index;
      __threadfence();
      *ctl.stop = 1;
    }
    return;
  }
  const int n = g.n;
  C2<T> *sa = reinterpret_cast<C2<T> *>(s_raw_x2);  // n * KB: V^_x, then kx B
  C2<T> *sb = sa + (size_t)n * KB;                  // n * KB: ky V^_y + kz V^_z, then B
  C2<T> *tw = sb + (size_t)n * KB;                  // n / 2
  for (int t = threadIdx.x; t < n / 2; t += blockDim.x) tw[t] = twiddle[t];
  const int ntk = g.nhp / KB;
  int bid = (int)blockIdx.x;
#if BCHMC_BX_SWIZZLE
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
#endif
  const int j = bid / ntk, k0 = (bid % ntk) * KB;
  const int c = threadIdx.x % KB, irow = threadIdx.x / KB;
  constexpr int rows = NT / KB;
  const int k = k0 + c;
  const long long plane = (long long)g.n * g.nhp;
  const long long col = k + (long long)g.nhp * j;
  const double ky = kval(j, g.n, g.kfac), kz = kval(k, g.n, g.kfac);
  const int shift = 32 - log2n;
  // One 32-bit byte offset per element serves every array of the kernel (uniform base pointer + lane offset: the
  // `saddr` form of the global instructions), instead of a 64-bit address per array and element -- those cost the
  // registers the requests in flight need.  The host checks that a component is smaller than 4 GB.
  unsigned boff[PER];
#pragma unroll
  for (int m = 0; m < PER; m++) boff[m] = (unsigned)((col + plane * (irow + rows * m)) * (long long)sizeof(C2<T>));
  auto at_c = [](const C2<T> *base, unsigned off) {
    return reinterpret_cast<const C2<T> *>(reinterpret_cast<const char *>(base) + off);
  };
  auto at_cw = [](C2<T> *base, unsigned off) {
    return reinterpret_cast<C2<T> *>(reinterpret_cast<char *>(base) + off);
  };
  auto at_d = [](const double *base, unsigned off) {
    return reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + off);
  };
  C2<T> *const Ck1 = Ck + g.Nhp, *const Ck2 = Ck + 2 * g.Nhp;
  {
    // ---- V^: all three components requested before the first one is used ----
    C2<T> vx[PER], vy[PER], vz[PER];
#pragma unroll
    for (int m = 0; m < PER; m++) {
      vx[m] = bx_load(at_c(Ck, boff[m]));
      vy[m] = bx_load(at_c(Ck1, boff[m]));
      vz[m] = bx_load(at_c(Ck2, boff[m]));
    }
#pragma unroll
    for (int m = 0; m < PER; m++) {
      const int i = irow + rows * m;
      const int r = (int)(__brev((unsigned)i) >> shift) * KB + c;
      sa[r] = vx[m];
      C2<T> w;
      w.x = (T)(ky * (double)vy[m].x + kz * (double)vz[m].x);
      w.y = (T)(ky * (double)vy[m].y + kz * (double)vz[m].y);
      sb[r] = w;
    }
  }
  // ---- the operands of the boundary arithmetic are requested now and arrive while the transforms run ----
  C2<T> qv[PER], pv[PER];
  double ws[PER], wm[PER];
#pragma unroll
  for (int m = 0; m < PER; m++) {
    const unsigned woff = sizeof(C2<T>) == 16 ? boff[m] / 2 : boff[m];  // doubles at the same element index
    qv[m] = bx_load(at_c(q_in, boff[m]));
    pv[m] = bx_load(at_c(p_in, boff[m]));
    ws[m] = (a != 0.) ? bx_load(at_d(wS, woff)) : 0.;
    wm[m] = wM ? bx_load(at_d(wM, woff)) : 0.;
  }
  __syncthreads();
  xfft_inplace2<T, false>(sa, sb, tw, n, log2n, KB);
  // h^ = (1/k^2) [ kx (Im V^_x, -Re V^_x) + (Im W^, -Re W^) ]
  double2 hk[PER];
#pragma unroll
  for (int m = 0; m < PER; m++) {
    const int i = irow + rows * m;
    const C2<T> v = sa[i * KB + c], w = sb[i * KB + c];
    const double kx = kval(i, g.n, g.kfac);
    hk[m] = make_double2(kx * (double)v.y, -(kx * (double)v.x));
    hk[m].x += (double)w.y;
    hk[m].y -= (double)w.x;
  }
  __syncthreads();  // every thread has read its transformed elements: the tiles are free for the inverse inputs
  double gsum = 0.;
#pragma unroll
  for (int m = 0; m < PER; m++) {
    const int i = irow + rows * m;
    const double kx = kval(i, g.n, g.kfac);
    const double ksq = kx * kx + ky * ky + kz * kz;
    const bool nyq = (i == g.n / 2) || (j == g.n / 2) || (k == g.n / 2);
    double2 q = make_double2((double)qv[m].x, (double)qv[m].y);
    double2 gg = make_double2(0., 0.);
    if (ksq > 0 && !nyq) {
      const double f = b * (1 / ksq);
      gg = make_double2(f * hk[m].x, f * hk[m].y);
    }
    if (a != 0.) {
      const double w = a * ws[m];
      gg.x += w * q.x;
      gg.y += w * q.y;
    }
    C2<T> gs;
    gs.x = (T)gg.x;
    gs.y = (T)gg.y;
    double2 p = make_double2((double)pv[m].x, (double)pv[m].y);
    C2<T> pe;
    pe.x = (T)(p.x - half_eps * gg.x);
    pe.y = (T)(p.y - half_eps * gg.y);
    const double hw = (k == 0 || ((g.n & 1) == 0 && k == g.n / 2)) ? 1. : 2.;
    if (k < g.nh) gsum += hw * (double)pe.x;
    p.x = (double)pe.x - half_eps * (double)gs.x;
    p.y = (double)pe.y - half_eps * (double)gs.y;
    {
      C2<T> t;
      t.x = (T)p.x;
      t.y = (T)p.y;
      bx_store(at_cw(p_out, boff[m]), t);
    }
    if (wM) {
      q.x += eps * (wm[m] * p.x);
      q.y += eps * (wm[m] * p.y);
    }
    C2<T> qs;
    qs.x = (T)q.x;
    qs.y = (T)q.y;
    bx_store(at_cw(q_out, boff[m]), qs);
    // Psi^_j = k_j B, B = (1/k^2)(Im phi^, -Re phi^), phi^ = c_za q^': kx B from the stored q' as computed, B from q'
    // as stored (rounded to T) -- the two inputs the first formulation feeds its inverse passes
    C2<T> o, o2;
    o.x = o.y = o2.x = o2.y = T(0);
    if (ksq > 1.e-14 && !nyq) {
      const double f = (1. / ksq) * kx;
      o.x = (T)(f * (c_za * q.y));
      o.y = (T)(f * -(c_za * q.x));
      const double f2 = 1. / ksq;
      o2.x = (T)(f2 * (c_za * (double)qs.y));
      o2.y = (T)(f2 * -(c_za * (double)qs.x));
    }
    const int r = (int)(__brev((unsigned)i) >> shift) * KB + c;
    sa[r] = o;
    sb[r] = o2;
  }
  __syncthreads();
  xfft_inplace2<T, true>(sa, sb, tw, n, log2n, KB);
#pragma unroll
  for (int m = 0; m < PER; m++) {
    const int i = irow + rows * m;
    bx_store(at_cw(Ck, boff[m]), sa[i * KB + c]);
    const C2<T> v = sb[i * KB + c];
    C2<T> oy, oz;
    oy.x = (T)(ky * (double)v.x);
    oy.y = (T)(ky * (double)v.y);
    oz.x = (T)(kz * (double)v.x);
    oz.y = (T)(kz * (double)v.y);
    bx_store(at_cw(Ck1, boff[m]), oy);
    bx_store(at_cw(Ck2, boff[m]), oz);
  }
  gsum = block_sum(gsum, red);
  if (threadIdx.x == 0) atomic_add_r(guard_slot, gsum);
}

}  // namespace bchmc
