// kernels.hpp -- HIP kernels of the bchmc engine (gfx950 / CDNA4, wave64).
//
// State lives in Fourier space: qk = R2C[q], pk = R2C[p] (unnormalised forward transforms, half-complex
// n x n x (n/2+1), z fastest).  Everything that is diagonal in k (prior force S^-1 q, drift M^-1 p, the
// Zel'dovich displacement kernel, the inverse-Laplacian-divergence of V, kicks) is done pointwise on those
// arrays; only the particle-mesh part (displace, SPH scatter, likelihood partials, SPH-gradient gather)
// runs in real space.  Reference lines restated by each kernel are cited at the kernel.
//
// Every kernel is a template on T, the STORAGE type of the field arrays (double = reference DOUBLE_PREC,
// float = BASELINE config 5).  k-space arithmetic, k-vectors, reductions and the per-cell likelihood are always
// done in double; the particle-mesh kernels (positions, spline evaluations, LDS accumulation) compute in T.
#pragma once
#include "common.hpp"

#include "kspace_step.hpp"  // Conversions, spectrum multipliers, first-step kick + drift + Zel'dovich kernel
#include "forward_model.hpp"  // Particle positions, SPH mass assignment (direct kernels), mean density, likelihood partials, direct SPH-gradient gather
#include "kspace_force.hpp"  // Force assembly, fused step boundary, rollback, Parseval energies, element-wise helpers, GRF likelihood
#include "variants.hpp"  // NGP / CIC / TSC mass assignment and the calc_h 0 / 3 likelihood-force variants
#include "rng.hpp"  // Philox4x32-10 momentum draw
#include "tiles.hpp"  // Tile-sorted particle-mesh path: binning, scan, LDS scatter / gather kernels
#include "tiles_low.hpp"  // NGP / CIC / TSC mass assignment and calc_h = 3's TSC interpolation on the same records
#include "alpt.hpp"  // ALPT displacement (Lag2Eul_non_zeldovich)
#include "spectrum.hpp"  // measure_spectrum
#include "step_boundary_x.hpp"  // Planes mode: step boundary with the x passes of both transforms fused in
#include "alpt_x.hpp"  // Planes mode of the ALPT displacement: mix + cell-boundary average with the x passes fused in
#include "zpass.hpp"  // Planes mode: the y and z passes of the inverse transform by the engine, the z pass ending in the binning
