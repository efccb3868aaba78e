// mh_mfma_ad.hpp -- mh_sweep_mfma_ad<KIND, NG, KX>: kernel_adapt / kernel_ram on the fp64-MFMA evaluation for data sets beyond
// the wave-specialised kernel's registers (round 4: n > 10240 at p <= 3, n > 5120 / 4096 at p <= 7 used to fall to the general
// kernel, 5-6x slower per flop -- profiles/r04_shape_map.md).
#pragma once

namespace {

// The evaluation is mh_sweep_mfma's EXT form (mh_mfma.hpp: v_mfma_f64_4x4x4 with A = [x_1 .. x_p, y], B = [b_1 .. b_p, -1],
// C = b_0 for 16 observations x 4 chains, the canonical tree's levels 1..32 inside the wave): NSR observation slots resident
// in operand registers, the rest streamed every step from the operand-order copy (mfma_build_stream).  All four chains of the
// workgroup are evaluated together by all eight waves; then waves 0..3 run ONE step each of the register-row adaptive owner
// of mh_sweep_spec (spec_owner_adaptive_reg: recursive covariance + Cholesky, or the product-form RAM update, rows in VGPRs)
// between two workgroup barriers.  At these sizes the evaluation is most of a step, so the exposed owner phase costs what it
// costs the normal kernels on mh_sweep_mfma; below them mh_sweep_spec, which overlaps owners and evaluation, stays the kernel.
template <int NG>
struct MfmaAdShape {
  static constexpr int NSR = (NG == 1) ? 12 : (NG == 2 ? 6 : 3);    // resident slots (operand registers: 4 NSR NG doubles per lane)
  static constexpr int RD = (NG == 1) ? 4 : 2;      // streamed slots in flight per wave
};

// the owners' side of a step (spec_owner_adaptive_reg's SYNC policy): they evaluate like every wave, then meet on barriers
template <class EV>
struct MfmaAdSync {
  static constexpr bool PREP_EARLY = true;
  EV& ev; const double* s_fold; int myc; unsigned* s_need;
  __device__ __forceinline__ double fold() const {
    double wsum = s_fold[myc * NW + (threadIdx.x & 7)];
    wsum = wsum + dpp_d<0xB1>(wsum);            // canonical levels 64, 128, 256 over the eight wave sums
    wsum = wsum + dpp_d<0x4E>(wsum);
    return wsum + dpp_d<0x141>(wsum);
  }
  // bounded kernel_ram (spec_owner_adaptive_reg<.., BND>): one more barrier per step tells the workgroup whether any of its
  // chains' proposals was moved by the reflection; if so all eight waves evaluate the four (republished) proposals once more
  template <class W> __device__ __forceinline__ bool second(bool need, W&& republish, double& tot2) const {
    if (need && (threadIdx.x & 63) == 0) lds_st_u32(s_need, 1u);
    lds_barrier();
    const bool again = __builtin_amdgcn_readfirstlane((int)lds_ld_u32(s_need)) != 0;
    if (!again) return false;
    republish();
    lds_barrier();                              // (every wave has read the flag; the proposals are in place)
    if (threadIdx.x == 0) lds_st_u32(s_need, 0u);
    ev();
    lds_barrier();
    tot2 = fold();
    return true;
  }
  __device__ __forceinline__ void second_idle() const {
    double t;
    (void)second(false, []() {}, t);
  }
  template <class F> __device__ __forceinline__ double total(int, F&& prep) const {
    ev();
    prep();                                     // the sigma-only half of the closed form, in the slack before the barrier
    lds_barrier();
    return fold();
  }
  __device__ __forceinline__ void publish(int) const { lds_barrier(); }
  __device__ __forceinline__ void final() const { lds_barrier(); }
};

// (the mirror kernels' owner, mfma_owner_mirror<KIND, SYNC>: mh_spec.hpp -- it also serves mh_sweep_spec since round 5)

// NSV: resident slots -- MfmaAdShape<NG>::NSR, or 1 for short data (512 < n <= 512 NSR: everything else streamed)
template <int KIND, int NG, int KX, bool BND = false, int NSV = MfmaAdShape<NG>::NSR>
__global__ __launch_bounds__(NT) void mh_sweep_mfma_ad(const SweepArgs A) {
  constexpr int CW = 4, NS = NSV, RD = MfmaAdShape<NG>::RD, TN = NS * 4, MB = TN < 12 ? TN : 12;
  static_assert(TN % MB == 0, "batches of MB pairs");
  static_assert(KX >= 0 || !BND, "the bounded kernel_ram has the register-row owner only");   // (KX = -1: matrices in LDS, -2: the mirror kernels' owner)
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = A.k;
  double* s_th1 = smem;                            // [CW][PIPE_KMAX]
  double* s_fold = s_th1 + CW * PIPE_KMAX;         // [CW][NW] per-wave sums of every chain (canonical levels 1..32 done)
  unsigned* s_need = reinterpret_cast<unsigned*>(s_fold + CW * NW);   // BND: "a reflection moved a proposal of this workgroup in this step"
  double* s_par = s_fold + CW * NW + 1;            // KX < 0: [4][PIPE_KMAX] mu | scale | lb | ub, then [CW][SPEC_ADS] the owners' matrices
  double* s_ad = s_par + 4 * PIPE_KMAX;
  if (tid == 0) lds_st_u32(s_need, 0u);
  if (KX == -1 && tid < PIPE_KMAX) {
    const bool in = tid < A.k;
    s_par[0 * PIPE_KMAX + tid] = in ? A.mu[tid] : 0.0;
    s_par[1 * PIPE_KMAX + tid] = in ? A.scale[tid] : 0.0;
    s_par[2 * PIPE_KMAX + tid] = in ? A.lb[tid] : 0.0;
    s_par[3 * PIPE_KMAX + tid] = in ? A.ub[tid] : 0.0;
  }
  const long long cg0 = (long long)blockIdx.x * CW;
  const int ncw = (int)((A.nchains - cg0 < CW) ? (A.nchains - cg0) : CW);
  const int nsteps = (int)A.nsteps, ic = A.intercept, P = A.p, next = A.mf_next;

  // ---- A operands of the resident slots (see mh_sweep_mfma)
  const int feat = lane >> 4, o16 = lane & 15;
  const int cl_a = 16 * (o16 & 3) + 4 * (o16 >> 2);
  double areg[NG][TN];
#pragma unroll
  for (int q = 0; q < NG; q++) {
    const int f = 4 * q + feat;
#pragma unroll
    for (int t = 0; t < TN; t++) {
      const int sl = t >> 2, g = t & 3;
      const long long i = (long long)(64 * wave + cl_a + g) + (long long)NT * sl;
      double a = 0.0;
      if (i < A.n) {
        if (f < P) a = A.X[(long long)f * A.n + i];
        else if (f == P) a = A.y[i];
      }
      areg[q][t] = a;
    }
  }
  const int jch = lane & 3;
  const int cl_d = 16 * (lane >> 4) + 4 * ((lane >> 2) & 3);
  unsigned vbits = 0;    // validity of this lane's 4 results in the LAST (streamed) slot; every earlier slot is full
#pragma unroll
  for (int g = 0; g < 4; g++)
    if ((long long)(64 * wave + cl_d + g) + (long long)NT * (NS - 1 + next) < A.n) vbits |= 1u << g;
  if (tid < CW * PIPE_KMAX) {
    const int c = tid / PIPE_KMAX, j = tid - c * PIPE_KMAX;
    s_th1[tid] = (c < ncw && j < k) ? A.theta0[(cg0 + c) * k + j] : 0.0;
  }
  lds_barrier();

  typedef double mf_d4 __attribute__((ext_vector_type(4)));
  const mf_d4* sp = reinterpret_cast<const mf_d4*>(A.mf_stream) + ((long long)wave * next * NG) * 64 + lane;
  // evaluation of the published proposals of all 4 chains; leaves this wave's sums in s_fold
  auto evaluate = [&]() {
    const double* tj = s_th1 + jch * PIPE_KMAX;
    double bop[NG];
#pragma unroll
    for (int q = 0; q < NG; q++) {
      const int f = 4 * q + feat;
      bop[q] = (f < P) ? tj[ic + f] : (f == P ? -1.0 : 0.0);
    }
    const double cop = ic ? tj[0] : 0.0;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    mf_d4 ring[RD][NG];                          // the first streamed slots are requested before the resident ones run
    // (round 5: next == 0 -- up to 512 observations -- streams nothing, the one resident slot is the last, with its padding: these requests
    //  then read slot 0 of a stand-in buffer and nobody uses them; a branch around them cost the streamed forms 7 - 10 %)
    const int rlast = next > 0 ? next - 1 : 0;
#pragma unroll
    for (int r = 0; r < RD; r++)
#pragma unroll
      for (int q = 0; q < NG; q++) ring[r][q] = sp[(((r < next) ? r : rlast) * NG + q) * 64];
    double cml[4];                               // C operands of the LAST slot: 0 where it is padding (-r == 0 exactly)
#pragma unroll
    for (int g = 0; g < 4; g++) cml[g] = ((vbits >> g) & 1u) ? cop : 0.0;
#pragma unroll
    for (int t0 = 0; t0 < TN; t0 += MB) {
      double d[MB];
#pragma unroll
      for (int u = 0; u < MB; u++) d[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(areg[0][t0 + u], bop[0], (next == 0 && t0 + u >= TN - 4) ? cml[(t0 + u) & 3] : cop, 0, 0, 0);
#pragma unroll
      for (int q = 1; q < NG; q++)
#pragma unroll
        for (int u = 0; u < MB; u++) d[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(areg[q][t0 + u], bop[q], d[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < MB; u++) acc[u & 3] = fmh_fma(d[u], d[u], acc[u & 3]);
    }
    for (int e0 = 0; e0 < next; e0 += RD) {
#pragma unroll
      for (int r = 0; r < RD; r++) {
        const int e = e0 + r;
        if (e < next) {                          // (uniform)
          mf_d4 a[NG];
#pragma unroll
          for (int q = 0; q < NG; q++) a[q] = ring[r][q];
          const int en = (e + RD < next) ? e + RD : next - 1;
#pragma unroll
          for (int q = 0; q < NG; q++) ring[r][q] = sp[(en * NG + q) * 64];
          const bool last = (e == next - 1);
          double d[4];
#pragma unroll
          for (int g = 0; g < 4; g++) d[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0][g], bop[0], last ? cml[g] : cop, 0, 0, 0);
#pragma unroll
          for (int q = 1; q < NG; q++)
#pragma unroll
            for (int g = 0; g < 4; g++) d[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[q][g], bop[q], d[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; g++) acc[g] = fmh_fma(d[g], d[g], acc[g]);
        }
      }
    }
    double fs = (acc[0] + acc[1]) + (acc[2] + acc[3]);        // canonical levels 1, 2 ... 32 (mh_sweep_mfma)
    fs = fs + dpp_d<0x114>(fs);
    fs = fs + dpp_d<0x118>(fs);
    {
      const unsigned long long u = (unsigned long long)__double_as_longlong(fs);
      const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
      const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
      const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      fs = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0])) +
           __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
    }
    {
      const unsigned long long u = (unsigned long long)__double_as_longlong(fs);
      const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
      const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      fs = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0])) +
           __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
    }
    if (lane >= 60) s_fold[jch * NW + wave] = fs;
  };

  const int myc = wave;
  MfmaAdSync<decltype(evaluate)> sync{evaluate, s_fold, myc < ncw ? myc : 0, s_need};
  if (myc >= ncw) {                               // waves without a chain: evaluate, meet the barriers of every step
    for (int v = 1; v <= nsteps; v++) {
      evaluate();
      lds_barrier();
      if constexpr (BND) sync.second_idle();
      lds_barrier();
    }
    return;
  }
  const int cl = __builtin_amdgcn_readfirstlane((int)cg0 + myc);
  // KX < 0: the owners with their matrices in LDS (spec_owner_adaptive: any k <= 16, fixed parameters) -- 8 .. 15 covariates, or a
  // fixed parameter, beyond the wave-specialised kernel's range
  // (measured and dropped at the end of round 4: the register-row owner with rows of 16 -- level at 8 .. 11 covariates, 15.0 against
  //  15.3 us per step at n = 1e4, and 1.3 - 1.7x SLOWER at 12 .. 15, where four operand groups leave the rows no registers)
  if constexpr (KIND == FMCMC_KERNEL_NMIRROR || KIND == FMCMC_KERNEL_UMIRROR) mfma_owner_mirror<KIND>(A, myc, cl, s_th1, sync);
  else if constexpr (KX < 0) spec_owner_adaptive<KIND>(A, myc, cl, s_th1, s_par, sync, s_ad + myc * SPEC_ADS);
  else spec_owner_adaptive_reg<KIND, KX, decltype(sync), BND>(A, myc, cl, s_th1, sync);
}

size_t mfma_ad_lds_bytes(bool lds_owner) { return sizeof(double) * ((size_t)4 * PIPE_KMAX + 4 * NW + 1 + (lds_owner ? 4 * PIPE_KMAX + 4 * SPEC_ADS : 0)); }

}  // namespace
