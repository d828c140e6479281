// alpt_x.hpp -- ALPT displacement on the 2-D plans: the k-space mix with the x passes of both transforms fused in.
// Part of the bchmc engine's kernel set; include through kernels.hpp (after step_boundary_x.hpp).
#pragma once
#include "alpt.hpp"
#include "step_boundary_x.hpp"

namespace bchmc {

// ======================================================================================================
// Planes-mode twin of k_alpt_mix (alpt.hpp): between the batched 2-D R2C of the two real-space sources
//   A = D1 delta(1) - D2 delta(2)   (Ck component 0)      B = spherical-collapse source   (component 1)
// and the batched 2-D C2R of the three displacement components, one workgroup owns the x-columns of KB adjacent k at
// one j (as k_step_boundary_x does): forward x passes of A^ and B^ in LDS, the mix
//   M = K A^ + B^ - K B^,  K = exp(-k^2 kth^2 / 2) / wtot,       Psi^_j = k_j E,  E = P (1 / (N k^2)) (Im M, -Re M)
// (theta2velcomp EqSolvers.cc:280-368 + convcomp convolution.cpp:327-377; Nyquist planes and k^2 <= 1e-14 -> 0),
// inverse x passes of kx E and of E (k_y, k_z are constants of a column: Psi_y, Psi_z leave through one transform).
// P = (1 + exp(-2 pi i (i + j + k) / n)) / 2 is cellboundcomp (massFunctions.cc:588-658: out[l] = (in[l] +
// in[l - (1,1,1)]) / 2, periodic) by the shift theorem: the averaging pass over the three real-space components
// disappears into this kernel (same numbers to round-off: the shift is exact on a periodic grid).
// Saves, per ALPT evaluation, against the 3-D path: one strided pass in each of the three transforms, the element-wise
// mix pass and the averaging pass.
// ======================================================================================================
template <typename T, int NT, int PER>
__global__ void __launch_bounds__(NT, BCHMC_BX_WAVES)
k_alpt_mix_x(Geo g, int log2n, const C2<T> *__restrict__ twiddle, C2<T> *Ck, double smol, double inv_wtot, double inv_n) {
  constexpr int KB = 128 / (int)sizeof(C2<T>);
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_ax[];
  const int n = g.n;
  C2<T> *s = reinterpret_cast<C2<T> *>(s_raw_ax);  // n * KB
  C2<T> *tw = s + (size_t)n * KB;                  // n / 2
  for (int t = threadIdx.x; t < n / 2; t += blockDim.x) tw[t] = twiddle[t];
  const int ntk = g.nhp / KB;
  const int j = blockIdx.x / ntk, k0 = (blockIdx.x % ntk) * KB;
  const int c = threadIdx.x % KB, irow = threadIdx.x / KB;
  constexpr int rows = NT / KB;
  const int k = k0 + c;
  const long long plane = (long long)g.n * g.nhp;
  const long long col = k + (long long)g.nhp * j;
  const double ky = kval(j, g.n, g.kfac), kz = kval(k, g.n, g.kfac);
  const int shift = 32 - log2n;
  double2 E[PER];
  for (int pass = 0; pass < 2; pass++) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < PER; m++) {
      const int i = irow + rows * m;
      s[(int)(__brev((unsigned)i) >> shift) * KB + c] = bx_load(Ck + col + plane * i + pass * g.Nhp);
    }
    __syncthreads();
    xfft_inplace<T>(s, tw, n, log2n, KB, false);
#pragma unroll
    for (int m = 0; m < PER; m++) {
      const int i = irow + rows * m;
      const C2<T> v = s[i * KB + c];
      if (pass == 0) {
        E[m] = make_double2((double)v.x, (double)v.y);  // A^
        continue;
      }
      const double2 A = E[m], B = make_double2((double)v.x, (double)v.y);
      const double kx = kval(i, g.n, g.kfac);
      const double ksq = kx * kx + ky * ky + kz * kz;
      const bool nyq = (i == g.n / 2) || (j == g.n / 2) || (k == g.n / 2);
      double2 e = make_double2(0., 0.);
      if (ksq > 1.e-14 && !nyq) {
        const double K = exp(-ksq * smol * smol / 2.) * inv_wtot;
        // K o Psi^2LPT + Psi^SC - K o Psi^SC, in the reference's order of operations (Lag2Eul.cc:240-250)
        const double mr = (K * A.x + B.x) - K * B.x, mi = (K * A.y + B.y) - K * B.y;
        const double fac = inv_n / ksq;
        const double ex = fac * mi, ey = fac * -mr;
        // cellboundcomp in k-space: exp(-2 pi i r / n) from the twiddle table (r >= n / 2: minus the entry r - n / 2)
        const int r = (i + j + k) & (n - 1);
        const C2<T> w = tw[r & (n / 2 - 1)];
        const double sg = (r >= n / 2) ? -1. : 1.;
        const double pr = 0.5 * (1. + sg * (double)w.x), pi = 0.5 * (sg * (double)w.y);
        e = make_double2(ex * pr - ey * pi, ex * pi + ey * pr);
      }
      E[m] = e;
    }
  }
  // inverse x passes: kx E -> component 0;  E -> components 1, 2 (times ky, kz)
  __syncthreads();
#pragma unroll
  for (int m = 0; m < PER; m++) {
    const int i = irow + rows * m;
    const double kx = kval(i, g.n, g.kfac);
    C2<T> o;
    o.x = (T)(kx * E[m].x);
    o.y = (T)(kx * E[m].y);
    s[(int)(__brev((unsigned)i) >> shift) * KB + c] = o;
  }
  __syncthreads();
  xfft_inplace<T>(s, tw, n, log2n, KB, true);
  for (int m = 0; m < PER; m++) {
    const int i = irow + rows * m;
    bx_store(Ck + col + plane * i, s[i * KB + c]);
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < PER; m++) {
    const int i = irow + rows * m;
    C2<T> o;
    o.x = (T)E[m].x;
    o.y = (T)E[m].y;
    s[(int)(__brev((unsigned)i) >> shift) * KB + c] = o;
  }
  __syncthreads();
  xfft_inplace<T>(s, tw, n, log2n, KB, true);
  for (int m = 0; m < PER; m++) {
    const int i = irow + rows * m;
    const C2<T> v = s[i * KB + c];
    C2<T> oy, oz;
    oy.x = (T)(ky * (double)v.x);
    oy.y = (T)(ky * (double)v.y);
    oz.x = (T)(kz * (double)v.x);
    oz.y = (T)(kz * (double)v.y);
    bx_store(Ck + col + plane * i + g.Nhp, oy);
    bx_store(Ck + col + plane * i + 2 * g.Nhp, oz);
  }
}

}  // namespace bchmc
