// mh_spec.hpp -- mh_sweep_spec<P, OPT, KIND>: 8 compute + 4 owner wavefronts meeting on LDS sequence words; owner roles
// for the normal kernels, kernel_adapt and kernel_ram (matrix rows in LDS or in registers).
#pragma once

namespace {

// ==============================================================================================
// Wave-specialised resident kernel (the headline path).
//
//   768 threads = 12 wavefronts per workgroup, 3 per SIMD:
//     waves 0..7  COMPUTE: hold the x columns of their 64 canonical lanes in VGPRs (y in LDS) and do nothing
//                 but evaluate: for version v, for chain c: wait ready[c] >= v, read theta1[c], 20 observations
//                 x (3 fma + sub + fma), write the lane partial, arrive on done[c].
//     waves 8..11 OWNERS (one per chain): wait done[c] == 8 v, fold the 512 partials (canonical tree), closed
//                 form, accept, propose, prefetch, publish theta1[c] (ready[c] = v + 1), then store the row.
//   No s_barrier in the steady state: producers/consumers meet on LDS sequence words, so an owner's
//   latency-bound phase overlaps the evaluation of the OTHER three chains, and on every SIMD the owner's
//   dependency stalls are filled by the two compute waves' independent FMAs (hardware multithreading instead
//   of compiler interleaving).  Register budget: 12 waves -> 168 VGPRs; one chain per evaluation pass keeps the
//   compute role at 120 (data) + ~30: 148 VGPRs, no scratch.
//   FMCMC_AMD_DEBUG=mode=8 stamps (s_memtime) flag-wait / work time per wave into the draws buffer.
// ==============================================================================================
constexpr int SPEC_NT = 768;
constexpr int SPEC_NCW = 8;   // compute wavefronts

__device__ __forceinline__ unsigned lds_ld_u32(const unsigned* p) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
  return v;
}

__device__ __forceinline__ void lds_st_u32(unsigned* p, unsigned v) {   // (completed by the s_waitcnt of the next lds_barrier)
  asm volatile("ds_write_b32 %0, %1" : : "v"((unsigned)(size_t)p), "v"(v) : "memory");
}

__device__ __forceinline__ double lds_ld_f64(const double* p) {   // ordered after a preceding flag poll
  double v;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
  return v;
}



constexpr int SPEC_ALD = PIPE_KMAX + 1;                       // row stride of the k x k matrices in LDS
constexpr int SPEC_ADS = 7 * PIPE_KMAX + 2 * PIPE_KMAX * SPEC_ALD;  // doubles of adaptive state per chain

// Owner role of the specialised kernel for kernel_adapt (R/kernel_adapt.R:117-180) and kernel_ram
// (R/kernel_ram.R:123-158, unbounded parameters): same wave-collective arithmetic as mh_sweep_kernel (lanes = rows
// of Sigma / S, twin of the oracle's propose_adapt / propose_ram), state in LDS, variates from the HBM stream.
template <int KIND, class SYNC>   // SYNC: how the owner meets the evaluation (SpecSync below / MfmaAdSync, mh_mfma_ad.hpp)
__device__ __forceinline__ void spec_owner_adaptive(const SweepArgs& A, int myc, int cl, double* s_th1, const double* s_par,
                                                    SYNC& sync, double* ad) {
  const int lane = threadIdx.x & 63;
  const int k = A.k, kz = A.kz, nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  constexpr int LD = SPEC_ALD;
  double* th0 = ad;                    // [k]
  double* th1 = th0 + PIPE_KMAX;       // [k]
  double* vz = th1 + PIPE_KMAX;        // [kf] z / U of the pending proposal
  double* vv = vz + PIPE_KMAX;         // [kf] S U, or x
  double* vmp = vv + PIPE_KMAX;        // [kf] mean_prev
  double* vmt = vmp + PIPE_KMAX;       // [kf] mean_t
  double* vrs = vmt + PIPE_KMAX;       // [kf] running sum of ans rows
  double* SigA = vrs + PIPE_KMAX;      // [kf][LD]
  double* SigB = SigA + PIPE_KMAX * LD;
  __shared__ int s_which[4][PIPE_KMAX];
  int kf = 0;
  for (int j = 0; j < k; j++)
    if (!A.fixed[j]) { if (lane == 0) s_which[myc][kf] = j; kf++; }
  const int* which = s_which[myc];
  const double* s_mu = s_par, *s_lb = s_par + 2 * PIPE_KMAX, *s_ub = s_par + 3 * PIPE_KMAX;
  double f0 = 0.0;
  long long abs_iter = 0;
  // (a continuation window of a long call -- launch_sweep, SweepArgs.win_cont -- takes over what the windows before it left:
  //  chain status, thinning counter, the running sum of this call's rows; the step-dependent rules read the CALL's step
  //  ioff + i: `i > 2`, the mean of the rows so far, eta(i, k), `i %% freq`)
  int nacc = 0, status = A.win_cont ? A.status[cl] : FMCMC_CHAIN_OK, thin_ctr = A.thin_ctr0, have_mean = 0, nerr = 0;
  const int ioff = (int)A.step_off;
  unsigned int bitword = 0;
  double* const Scur = SigA;
  if (lane < k) { double t = A.theta0[(long long)cl * k + lane]; th0[lane] = t; th1[lane] = t; }
  for (int e = lane; e < kf * LD; e += 64) {
    const int a = e / LD, b = e % LD;   // (ram: S is a LOWER factor, anything above its diagonal is not part of it)
    SigA[e] = A.fresh ? ((a == b) ? 1.0 * A.eps : 0.0) : ((b < kf && (b <= a || KIND == FMCMC_KERNEL_ADAPT)) ? A.Sigma[((long long)cl * kf + a) * kf + b] : 0.0);
    SigB[e] = 0.0;
  }
  if (!A.fresh) {
    abs_iter = A.abs_iter[cl];
    if (A.nerrors) nerr = A.nerrors[cl];
    if (KIND == FMCMC_KERNEL_ADAPT) {
      have_mean = A.have_mean[cl];
      if (lane < kf) vmp[lane] = A.mean_prev[(long long)cl * kf + lane];
    }
  }
  wave_sync_lds();
  const int jl = (lane < k) ? lane : 0;
  // (row stores and variates: wave-uniform 64-bit bases of the chain's own blocks + 32-bit offsets -- 32-bit offsets from the
  //  BUFFER bases capped a call at 4 GiB)
  // (wave-uniform 64-bit bases -- the chain's block -- plus a 32-bit byte offset per lane: one scalar-base store each)
  char* const s_base = reinterpret_cast<char*>(A.samples) + ((long long)cl * k) * A.ldS * 8;
  char* const d_base = A.draws ? reinterpret_cast<char*>(A.draws) + ((long long)cl * k) * A.ldS * 8 : nullptr;
  char* const l_base = A.logpost ? reinterpret_cast<char*>(A.logpost) + (long long)cl * A.ldS * 8 : nullptr;
  const unsigned int lane_off = (unsigned int)((long long)jl * A.ldS * 8);
  unsigned int srow8 = 0;
  const char* const z_base = reinterpret_cast<const char*>(A.fed_z) + ((long long)cl * nsteps) * kz * 8;   // wave-uniform
  const unsigned int z_lane = (unsigned int)((lane < kz ? lane : 0)) * 8u;
  const double* const lu_row = A.fed_logu + (long long)cl * nsteps;
  auto ld_z = [&](int row) -> double {
    return *reinterpret_cast<const double*>(z_base + (z_lane + (unsigned int)row * (unsigned int)(kz * 8)));
  };
  double z_nx = (lane < kz && nsteps >= 2) ? ld_z(1) : 0.0;
  double lu_nx = (nsteps >= 2) ? lu_row[1] : 0.0;
  bool ram_gate = false;   // gate of the PENDING proposal (evaluated when it was made)
  auto flush_bits = [&](int i) {
    if (A.accept_bits && lane == 0) {   // (bits_stride: words per chain of the whole call's bitmap; the first word of a continuation window also holds the last bit of the window before it)
      unsigned int* w = A.accept_bits + ((long long)cl * A.bits_stride + ((i - 1) >> 5));
      *w = (A.win_cont && i <= 32) ? (*w | bitword) : bitword;
    }
    bitword = 0;
  };

  for (int v = 1; v <= nsteps; v++) {
    const double tot = sync.total(v, []() {});
    const double f1 = finish_logpost(A, th1, tot);
    bool keep_row = false, st_row = false;
    double st_th0 = 0.0, st_dr = 0.0;
    if (v == 1) {
      f0 = f1;
      if (lane < kf) vrs[lane] = (A.win_cont && A.win_sum) ? A.win_sum[(long long)cl * kf + lane] : th0[which[lane]];
      keep_row = true;
    } else if (status == FMCMC_CHAIN_OK) {
      const int i = v;
      if (KIND == FMCMC_KERNEL_RAM) {   // adaptation with f(theta1) of the pending (un-reflected) proposal :129-152
        if (ram_gate) {
          double a_n = fmh_exp(f1 - f0);
          if (fmh_isnan(a_n)) a_n = 0.0;
          else if (a_n > 1.0) a_n = 1.0;
          double eta = (double)kf * fmh_exp(A.ram_neg_exp * fmh_log((double)(i + ioff)));
          if (eta > 1.0) eta = 1.0;
          // S <- S T in place (mh_common.hpp, ram_coef; twin of the oracle's ram_factor_update_canon)
          const double zl = (lane < kf) ? vz[lane] : 0.0;
          const double Pj1 = lane_scan_row16(zl * zl);
          const double Pj = dpp_d<0x111>(Pj1);                 // row_shr:1, lane 0 reads 0
          const double nrm2 = readlane_d(Pj1, kf - 1);
          double cp = (eta * (a_n - A.arate)) / nrm2;
          if (cp != 0.0 && fmh_isfinite(cp)) {
            double dl, kl;
            const bool okl = ram_coef(cp, Pj, Pj1, zl, dl, kl);
            if (__any(lane < kf && !okl)) {
              nerr += 1;
            } else {
              double* row = Scur + ((lane < kf) ? lane : 0) * LD;
              const double* grow = SigB + ((lane < kf) ? lane : 0) * LD;   // G_ij, kept by the proposal
              for (int j = 0; j < kf; j++) {
                const double dj = readlane_d(dl, j), kj = readlane_d(kl, j);
                const double sij = row[j];
                const double nw = fmh_fma(grow[j], kj, sij * dj);
                if (lane >= j && lane < kf) row[j] = nw;
              }
              wave_sync_lds();
            }
          }
        }
        abs_iter += 1;
      }
      if (fmh_isnan(f1)) status = FMCMC_CHAIN_NAN_LOGPOST;
      const double ratio = f1 - f0;
      if (status == FMCMC_CHAIN_OK && fmh_isnan(ratio)) status = FMCMC_CHAIN_NAN_RATIO;
      if (status != FMCMC_CHAIN_OK) {
        if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i + ioff; }
        if (lane < k) A.status_theta[(long long)cl * k + lane] = th1[lane];
        flush_bits(i);
      } else {
        const double lu = lu_nx;
        lu_nx = lu_row[v < nsteps ? v : nsteps - 1];
        keep_row = true;
        // row i of ans / draws / logpost (the proposal th1 is still the evaluated one here)
        if (i > burnin) {
          thin_ctr += 1;
        }
        const bool acc = lu < ratio;
        const double dr = (lane < k) ? th1[lane] : 0.0;
        if (acc) {
          if (lane < k) th0[lane] = dr;
          f0 = f1;
          nacc += 1;
          bitword |= (1u << ((i - 1) & 31));
        }
        wave_sync_lds();
        if (i > burnin && thin_ctr == thin) {   // stored after the proposal is published (below)
          thin_ctr = 0;
          st_row = true;
          st_th0 = (lane < k) ? th0[lane] : 0.0;
          st_dr = dr;
        }
        if (KIND == FMCMC_KERNEL_ADAPT && lane < kf) vrs[lane] = vrs[lane] + th0[which[lane]];
        if (((i - 1) & 31) == 31 || i == nsteps) flush_bits(i);
      }
    }
    if (v == 1 && keep_row && 1 > burnin) {   // row 1 (R/mcmc.R:737-743)
      thin_ctr += 1;
      if (thin_ctr == thin) {
        thin_ctr = 0;
        if (lane < k) {
          *reinterpret_cast<double*>(s_base + (lane_off + srow8)) = th0[lane];
          if (d_base) *reinterpret_cast<double*>(d_base + (lane_off + srow8)) = th1[lane];
        }
        if (l_base && lane == 0) *reinterpret_cast<double*>(l_base + srow8) = f1;
        srow8 += 8;
      }
    }
    // ---- proposal of loop step i = v + 1
    if (v < nsteps) {
      if (status == FMCMC_CHAIN_OK) {
        const int i = v + 1;
        wave_sync_lds();
        if (lane < kz) vz[lane] = z_nx;
        z_nx = (lane < kz) ? ld_z(v + 1 < nsteps ? v + 1 : nsteps - 1) : 0.0;
        wave_sync_lds();
        if (KIND == FMCMC_KERNEL_ADAPT) {
          if (A.until > (double)abs_iter && abs_iter > A.warmup && i + ioff > 2) {
            const double t = (double)(abs_iter - 1);
            double x = 0, mp = 0, mt = 0;
            if (lane < kf) {
              x = th0[which[lane]];
              mp = have_mean ? vmp[lane] : (vrs[lane] / (double)(i + ioff - 1));
              mt = (mp * t + x) / (t + 1);
              vv[lane] = x; vmp[lane] = mp; vmt[lane] = mt;
            }
            wave_sync_lds();
            if (lane < kf) {
              // (round 5: four columns' LDS reads in flight -- one element per LDS round trip made this loop and the factor's ~7 us
              //  of a step at k = 14; same operations per element, same bits)
              const double c1 = (t - 1) / t, c2 = 1.0 / t;
              int b = 0;
              for (; b + 4 <= kf; b += 4) {
                double pb[4], tb[4], xb[4], sb[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { pb[u] = vmp[b + u]; tb[u] = vmt[b + u]; xb[u] = vv[b + u]; sb[u] = SigA[lane * LD + b + u]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                  const double ik = (b + u == lane) ? 1.0 * A.eps : 0.0;
                  const double inner = t * (mp * pb[u]) - (t + 1) * (mt * tb[u]) + x * xb[u] + 1e-5 * ik;
                  SigA[lane * LD + b + u] = c1 * sb[u] + c2 * inner;
                }
              }
              for (; b < kf; b++) {
                double ik = (b == lane) ? 1.0 * A.eps : 0.0;
                double inner = t * (mp * vmp[b]) - (t + 1) * (mt * vmt[b]) + x * vv[b] + 1e-5 * ik;
                SigA[lane * LD + b] = c1 * SigA[lane * LD + b] + c2 * inner;
              }
            }
            wave_sync_lds();
            if (lane < kf) vmp[lane] = mt;
            have_mean = 1;
          }
          abs_iter += 1;
          // root-free factor Sigma = L D L^T (twin of the oracle's ldl_lower_canon): L in the lower triangle of SigB, the numerators
          // W_ib = L_ib D_b in its upper one (W_ib at [b][i]), D in vv (the x of the covariance update is done with)
          bool notpd = false;
          for (int j = 0; j < kf; j++) {
            double sacc = 0.0;
            if (lane >= j && lane < kf) {
              sacc = SigA[lane * LD + j];
              int b = 0;
              for (; b + 4 <= j; b += 4) {
                double lb4[4], wb4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { lb4[u] = SigB[lane * LD + b + u]; wb4[u] = SigB[(b + u) * LD + j]; }
#pragma unroll
                for (int u = 0; u < 4; u++) sacc = fmh_fma(-lb4[u], wb4[u], sacc);
              }
              for (; b < j; b++) sacc = fmh_fma(-SigB[lane * LD + b], SigB[b * LD + j], sacc);
            }
            const double d = readlane_d(sacc, j);     // (the pivot as a scalar: no LDS round trip of a ds_bpermute)
            if (!(d > 0.0) || !fmh_isfinite(d)) { notpd = true; break; }
            if (lane == j) { SigB[j * LD + j] = 1.0; vv[j] = d; }
            else if (lane > j && lane < kf) { SigB[lane * LD + j] = sacc / d; SigB[j * LD + lane] = sacc; }
            wave_sync_lds();
          }
          if (notpd) {
            status = FMCMC_CHAIN_NOT_PD;
            if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i + ioff; }
            if (lane < k) A.status_theta[(long long)cl * k + lane] = th1[lane];
            if (((v - 1) & 31) != 31) flush_bits(v);   // (the accept bits of the steps decided so far; a full word has just been flushed)
          } else {
            if (lane < k) th1[lane] = th0[lane];
            if (lane < kf) vz[lane] = fmh_sqrt(vv[lane]) * vz[lane];    // u = sqrt(D) z
            wave_sync_lds();
            if (lane < kf) {
              double sacc = 0.0;
              for (int b = 0; b < kf; b += 4) {     // (terms b <= lane, in order; four columns' reads in flight, clamped inside the row)
                double lb4[4], zb4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int bb = (b + u < kf) ? b + u : kf - 1; lb4[u] = SigB[lane * LD + bb]; zb4[u] = vz[bb]; }
#pragma unroll
                for (int u = 0; u < 4; u++) { const double nx = fmh_fma(lb4[u], zb4[u], sacc); sacc = (b + u <= lane) ? nx : sacc; }
              }
              const int j = which[lane];
              th1[j] = reflect1(th0[j] + (s_mu[j] + sacc), s_lb[j], s_ub[j]);
            }
          }
        } else {  // RAM P1 :123-126 (theta1 keeps its previous values in fixed coordinates)
          if (lane < kf) {
            double sacc = 0.0;
            for (int b = lane; b >= 0; b--) {   // from the diagonal down to column 0, keeping the partial sums G_ib
              SigB[lane * LD + b] = sacc;
              sacc = fmh_fma(Scur[lane * LD + b], vz[b], sacc);
            }
            const int j = which[lane];
            th1[j] = th0[j] + sacc;
          }
          ram_gate = (A.until > (double)abs_iter && abs_iter > A.warmup && ((v + 1 + ioff) % A.freq) == 0);
        }
        wave_sync_lds();
        if (lane < k) s_th1[myc * PIPE_KMAX + lane] = th1[lane];
      }
      sync.publish(v + 1);
    } else {
      sync.final();
    }
    if (st_row) {   // row v of ans / draws / logpost, off the compute waves' critical path
      if (lane < k) {
        *reinterpret_cast<double*>(s_base + (lane_off + srow8)) = st_th0;
        if (d_base) *reinterpret_cast<double*>(d_base + (lane_off + srow8)) = st_dr;
      }
      if (l_base && lane == 0) *reinterpret_cast<double*>(l_base + srow8) = f1;
      srow8 += 8;
    }
  }
  // ---- write state back
  wave_sync_lds();
  if (lane < k) A.theta0[(long long)cl * k + lane] = th0[lane];
  if (lane == 0) {
    A.f0[cl] = f0;
    A.accept_count[cl] = nacc;
    if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    A.abs_iter[cl] = abs_iter;
    if (A.nerrors) A.nerrors[cl] = nerr;
    if (KIND == FMCMC_KERNEL_ADAPT) A.have_mean[cl] = have_mean;
  }
  if (KIND == FMCMC_KERNEL_ADAPT && A.win_sum && lane < kf) A.win_sum[(long long)cl * kf + lane] = vrs[lane];
  const double* Sfin = (KIND == FMCMC_KERNEL_RAM) ? Scur : SigA;
  for (int e = lane; e < kf * kf; e += 64) {
    const int a = e / kf, b = e % kf;
    A.Sigma[((long long)cl * kf + a) * kf + b] = Sfin[a * LD + b];
  }
  if (KIND == FMCMC_KERNEL_ADAPT && lane < kf) A.mean_prev[(long long)cl * kf + lane] = vmp[lane];
}

// Register-row variant of the adaptive owner for k <= SPEC_KA parameters, none fixed (C3, kernel_ram at k = 5):
// lane a keeps ROW a of Sigma / S (and of the Cholesky factor) in VGPRs and other rows' entries arrive by v_readlane
// (statically unrolled indices), so the Cholesky, the rank-1 update, the recursive covariance and the L z / S U
// products run without a single LDS round trip or ds_bpermute.  Same operations in the same order per matrix
// element as spec_owner_adaptive / the oracle, hence the same bits.
constexpr int SPEC_KA = 8;

// KX > 0: the number of parameters is the compile-time constant KX (the statically unrolled loops then carry no `b < kf`
// predicates and no dead iterations: at k = 5 the owner loop was 1800 instructions per step with KA = 8, and an owner
// wave issues one instruction per ~6.5 cycles: the chain-step rate of C3 IS this instruction count); KX == 0: k <= 8.
// How the owner meets the evaluation is a policy (round 4): SpecSync = mh_sweep_spec's LDS sequence words (the evaluation runs
// in other waves, chain by chain); the MFMA kernel's owners (mh_mfma_ad.hpp) evaluate themselves and meet on barriers.
//   total(v, prep): the canonical total of version v's evaluation of this chain (uniform); PREP_EARLY policies run prep() -- what
//                   the decision needs and the evaluation does not enter -- inside it, in front of their wait
//   publish(vn):    theta1 of version vn is in s_th1;   final(): behind the last decision (no proposal)
struct SpecSync {
  static constexpr bool PREP_EARLY = false;
  unsigned* s_ready; unsigned* s_done; const double* s_tr; int myc;
  template <class F> __device__ __forceinline__ double total(int v, F&&) const {
    const int lane = threadIdx.x & 63;
    while (lds_ld_u32(&s_done[myc]) < 8u * (unsigned)v) __builtin_amdgcn_s_sleep(1);
    const double* src = s_tr + myc * (8 * PIPE_TRS) + lane;
    const double v0 = src[0 * PIPE_TRS], v1 = src[1 * PIPE_TRS], v2 = src[2 * PIPE_TRS], v3 = src[3 * PIPE_TRS];
    const double v4 = src[4 * PIPE_TRS], v5 = src[5 * PIPE_TRS], v6 = src[6 * PIPE_TRS], v7 = src[7 * PIPE_TRS];
    return wave_xor_sum(((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)));
  }
  __device__ __forceinline__ void publish(int vn) const {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(&s_ready[myc], (unsigned)vn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ void final() const {}
};
// The bounded kernel_ram on this kernel (round 5; it ran on the barrier-synchronised MFMA kernel, or -- up to 512 observations -- on
// the general one: 5.4 us per step at the README's size where the unbounded kernel takes 2.1): evaluations are REQUESTS, numbered per
// chain -- s_ready[c] = requests published, s_done[c] = 8 x requests served --; a step asks once, and once more when the reflection
// moved its proposal; the compute waves serve whatever request a chain has next (spec_compute<.., BNDC>) until the owner's final().
// A step without a reflection costs what the unbounded kernel's costs.
constexpr unsigned SPEC_FINAL = 0xFFFFFFFFu;
struct SpecSyncB {
  static constexpr bool PREP_EARLY = false;
  unsigned* s_ready; unsigned* s_done; const double* s_tr; int myc;
  mutable unsigned nreq = 1u;          // (request 1: the initial vector, published by the kernel's set-up)
  __device__ __forceinline__ double served() const {
    const int lane = threadIdx.x & 63;
    while (lds_ld_u32(&s_done[myc]) < 8u * nreq) __builtin_amdgcn_s_sleep(1);
    const double* src = s_tr + myc * (8 * PIPE_TRS) + lane;
    const double v0 = src[0 * PIPE_TRS], v1 = src[1 * PIPE_TRS], v2 = src[2 * PIPE_TRS], v3 = src[3 * PIPE_TRS];
    const double v4 = src[4 * PIPE_TRS], v5 = src[5 * PIPE_TRS], v6 = src[6 * PIPE_TRS], v7 = src[7 * PIPE_TRS];
    return wave_xor_sum(((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)));
  }
  __device__ __forceinline__ void ask() const {
    nreq += 1u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(&s_ready[myc], nreq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  template <class F> __device__ __forceinline__ double total(int, F&&) const { return served(); }
  template <class W> __device__ __forceinline__ bool second(bool need, W&& republish, double& tot2) const {
    if (!need) return false;
    republish();
    ask();
    tot2 = served();
    return true;
  }
  __device__ __forceinline__ void second_idle() const {}
  __device__ __forceinline__ void publish(int) const { ask(); }
  __device__ __forceinline__ void final() const {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(&s_ready[myc], SPEC_FINAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
};
// BND (kernel_ram with bounds, R/kernel_ram.R:123-157 + R/mcmc.R:749-753): the adaptation takes f of the proposal as drawn, the
// decision f of the REFLECTED proposal -- a second evaluation, needed only in the steps in which a reflection moved something.  The
// policy decides it for its workgroup: second(need, republish, tot2) -> true when a second evaluation ran (tot2: its total for this
// chain), second_idle() for an owner that has nothing to ask in this step but must keep in step (MfmaAdSync; SpecSync has neither).
constexpr int SPEC_FREQMAX = 8;    // kernel_adapt(freq = 2 .. 8, bw = 0) on the register owner: the last `freq` rows of the chain in an LDS ring
// RING: kernel_adapt(freq > 1) -- a template parameter because the run-time stride (its modulo, the fold loop, the quotients it cannot
// prepare) in the one owner cost kernel_adapt(freq = 1) 15 % in the latency form at n = 10,000 (2.20 -> 2.55 us per step, found by the
// perf guard's record at the end of round 5; `tools/ab_lib.py`)
template <int KIND, int KX, class SYNC, bool BND = false, int FAM = FMCMC_FAM_GAUSSIAN_LINREG, bool RING = false>
__device__ __forceinline__ void spec_owner_adaptive_reg(const SweepArgs& A, int myc, int cl, double* s_th1, SYNC& sync,
                                                        double* ring = nullptr /* LDS [SPEC_FREQMAX][PIPE_KMAX]; needed for freq > 1 */) {
  static_assert(!BND || KIND == FMCMC_KERNEL_RAM, "BND is the bounded kernel_ram");
  constexpr bool LG = FAM == FMCMC_FAM_LOGISTIC;
  constexpr int KA = KX > 0 ? KX : SPEC_KA;
  const int lane = threadIdx.x & 63;
  const int k = KX > 0 ? KX : A.k, kz = KX > 0 ? KX : A.kz, kf = (KX > 0 || LG) ? k : kz;
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  const bool rl = lane < k;                 // a lane with a parameter (its row of the chain, its entry of the published vector)
  const int jl = rl ? lane : 0;
  // KX == 0 (run-time width, k <= 8) takes FIXED parameters too (round 5: they ran on the owners with their matrices in LDS): the kf free
  // parameters on lanes 0 .. kf - 1 -- all the arithmetic, in the space the reference's `which` spans --, the fixed ones behind them as
  // passengers that keep their value (pj: a lane's parameter; sig_lane: the lane that holds sigma).  Compile-time widths: the identity.
  const bool fl = lane < kf;                // a lane with a FREE parameter
  int pj = jl, sig_lane = k - 1;
  if constexpr (KX == 0 && !LG) {
    int nfree = 0, nfix = 0, mine = 0;
    for (int j = 0; j < k; j++) {
      const bool fx = A.fixed[j] != 0;
      const int pos = fx ? (kf + nfix) : nfree;
      if (pos == lane) mine = j;
      if (j == k - 1) sig_lane = pos;
      nfree += fx ? 0 : 1; nfix += fx ? 1 : 0;
    }
    pj = rl ? mine : 0;
  }
  const double mu_l = A.mu[pj], lb_l = A.lb[pj], ub_l = A.ub[pj];
  double Srow[KA], Lrow[KA];                // Sigma (adapt) or S (ram) row `lane`; row of the unit lower factor L (adapt)
  double Wrow[KIND == FMCMC_KERNEL_ADAPT ? KA : 1];   // adapt: the numerators W_ib = L_ib D_b of the factor's row
  double (&Grow)[KA] = Lrow;                // ram: G_ib, the partial sums of the proposal's (S z)_lane chain
#pragma unroll
  for (int b = 0; b < (KIND == FMCMC_KERNEL_ADAPT ? KA : 1); b++) Wrow[b] = 0.0;
#pragma unroll
  for (int b = 0; b < KA; b++) {
    Lrow[b] = 0.0;
    Srow[b] = (fl && b < kf) ? (A.fresh ? ((b == lane) ? 1.0 * A.eps : 0.0)
                                        : ((b <= lane || KIND == FMCMC_KERNEL_ADAPT) ? A.Sigma[((long long)cl * kf + lane) * kf + b] : 0.0)) : 0.0;
  }
  double th0 = rl ? A.theta0[(long long)cl * k + pj] : 0.0, th1 = th0;
  double f0 = 0.0, mean_prev = 0.0, run_sum = 0.0, zcur = 0.0;
  double Dl = 0.0;                          // adapt: D_lane of the factor (kept between the steps that form it)
  long long abs_iter = 0;
  // (continuation windows of a long call: see spec_owner_adaptive)
  int nacc = 0, status = A.win_cont ? A.status[cl] : FMCMC_CHAIN_OK, thin_ctr = A.thin_ctr0, have_mean = 0, nerr = 0;
  const int ioff = (int)A.step_off;
  unsigned int bitword = 0;
  if (!A.fresh) {
    abs_iter = A.abs_iter[cl];
    if (A.nerrors) nerr = A.nerrors[cl];
    if (KIND == FMCMC_KERNEL_ADAPT) {
      have_mean = A.have_mean[cl];
      if (fl) mean_prev = A.mean_prev[(long long)cl * kf + lane];
    }
  }
  // (row stores and variates: wave-uniform 64-bit bases of the chain's own blocks + 32-bit offsets -- 32-bit offsets from the
  //  BUFFER bases capped a call at 4 GiB)
  // (wave-uniform 64-bit bases -- the chain's block -- plus a 32-bit byte offset per lane: one scalar-base store each)
  char* const s_base = reinterpret_cast<char*>(A.samples) + ((long long)cl * k) * A.ldS * 8;
  char* const d_base = A.draws ? reinterpret_cast<char*>(A.draws) + ((long long)cl * k) * A.ldS * 8 : nullptr;
  char* const l_base = A.logpost ? reinterpret_cast<char*>(A.logpost) + (long long)cl * A.ldS * 8 : nullptr;
  const unsigned int lane_off = (unsigned int)((long long)pj * A.ldS * 8);
  unsigned int srow8 = 0;
  const char* const z_base = reinterpret_cast<const char*>(A.fed_z) + ((long long)cl * nsteps) * kz * 8;   // wave-uniform
  const unsigned int z_lane = (unsigned int)(fl ? lane : 0) * 8u;
  const double* const lu_row = A.fed_logu + (long long)cl * nsteps;
  const double dn = uniform_d((double)A.n);
  auto ld_z = [&](int row) -> double {
    return *reinterpret_cast<const double*>(z_base + (z_lane + (unsigned int)row * (unsigned int)(kz * 8)));
  };
  double z_nx = (fl && nsteps >= 2) ? ld_z(1) : 0.0;
  double lu_nx = (nsteps >= 2) ? lu_row[1] : 0.0;
  bool ram_gate = false;
  auto flush_bits = [&](int i) {
    if (A.accept_bits && lane == 0) {
      unsigned int* w = A.accept_bits + ((long long)cl * A.bits_stride + ((i - 1) >> 5));
      *w = (A.win_cont && i <= 32) ? (*w | bitword) : bitword;
    }
    bitword = 0;
  };
  auto logpost_of = [&](double tot, double sigma) -> double {   // Gaussian linreg closed form (same as the normal owners)
    double f;
    if (sigma < 0.0 || fmh_isnan(sigma)) f = fmh_nan();
    else if (sigma == 0.0) f = -fmh_inf();
    else {
      double t1 = fmh_log(sigma) + FMH_K(FMH_LN_SQRT_2PI);
      double q = (0.5 * tot) / (sigma * sigma);
      f = -(dn * t1) - q;
    }
    if (A.guard && !fmh_isfinite(f)) f = -fmh_inf();
    return f;
  };

  // What the decision of a step needs and f(theta1) does not enter is prepared while the compute waves evaluate the
  // pending proposal (this wave would only poll): the sigma-only part of the closed form -- n (log sigma + ln sqrt 2 pi),
  // sigma^2 and the reciprocal half of the division (div_recip / div_finish: bit for bit the plain quotient in the safe
  // range, anything else takes logpost_of) -- and for kernel_ram eta(i, k) and the prefix sums of z^2.
  double pre_nt1 = 0.0, pre_ss = 1.0, pre_rs = 1.0, pre_eta = 0.0, pre_Pj = 0.0, pre_Pj1 = 0.0, pre_nrm2 = 1.0;
  double pre_c1 = 0.0, pre_c2 = 0.0;   // kernel_adapt: (t - 1) / t and 1 / t of the NEXT covariance update (t = abs_iter - 1: the step's own)
  double pre_lin = 0.0, pre_pri = 0.0; // logistic: sum_j b_j hs_j and sum_j b_j^2 / prior_div of the pending proposal
  const int nbl = A.intercept + A.p;
  const double hs_l = (LG && lane < nbl && A.lg_hs) ? A.lg_hs[lane] : 0.0;
  bool pre_ok = false;
  auto prepare = [&](int i) {   // i: loop step whose proposal is pending (1: the initial state)
    if (KIND == FMCMC_KERNEL_ADAPT) {
      const double t = uniform_d((double)(abs_iter - 1));
      pre_c1 = (t - 1) / t;
      pre_c2 = 1.0 / t;
    }
    if constexpr (LG) {   // (finish_logpost<LOGISTIC>: the same fma chains, the same bits; no sigma in this family)
      double lin = 0.0, ss = 0.0;
      for (int j = 0; j < nbl; j++) {
        const double bj = readlane_d(th1, j);
        lin = fmh_fma(bj, readlane_d(hs_l, j), lin);
        ss = fmh_fma(bj, bj, ss);
      }
      pre_lin = lin;
      pre_pri = (A.prior_div != 0.0) ? ss / A.prior_div : 0.0;
      if (KIND == FMCMC_KERNEL_RAM && i >= 2) {
        double eta = (double)kf * fmh_exp(A.ram_neg_exp * fmh_log((double)(i + ioff)));
        if (eta > 1.0) eta = 1.0;
        pre_eta = eta;
        pre_Pj1 = lane_scan_row16(zcur * zcur);
        pre_Pj = dpp_d<0x111>(pre_Pj1);
        pre_nrm2 = readlane_d(pre_Pj1, kf - 1);
      }
      return;
    }
    const double sigma = readlane_d(th1, sig_lane);
    const unsigned sg_hi = (unsigned)(fmh_d2u(sigma) >> 32);
    const bool sg_fast = (sg_hi - 0x00100000u) < 0x7fe00000u;            // positive, finite, normal
    const double sg = sg_fast ? sigma : 1.0;
    pre_nt1 = dn * (fmh_log(sg) + FMH_K(FMH_LN_SQRT_2PI));
    pre_ss = sg * sg;
    pre_ok = sg_fast && mfr_div_safe(pre_ss);
    pre_rs = div_recip(pre_ok ? pre_ss : 1.0);
    if (KIND == FMCMC_KERNEL_RAM && i >= 2) {
      double eta = (double)kf * fmh_exp(A.ram_neg_exp * fmh_log((double)(i + ioff)));
      if (eta > 1.0) eta = 1.0;
      pre_eta = eta;
      pre_Pj1 = lane_scan_row16(zcur * zcur);                // (zcur is 0 beyond the parameters)
      pre_Pj = dpp_d<0x111>(pre_Pj1);                        // row_shr:1, lane 0 reads 0
      pre_nrm2 = readlane_d(pre_Pj1, kf - 1);
    }
  };
  if (!SYNC::PREP_EARLY) prepare(1);
#ifdef SPEC_STAMP   /* diagnostic build (tools/exp_spec.hip): s_memtime shares of the owner's phases */
  unsigned long long stt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stp = clk();
#define SPEC_ST(i) do { const unsigned long long t_ = clk(); stt[i] += t_ - stp; stp = t_; } while (0)
#else
#define SPEC_ST(i) do { } while (0)
#endif

  for (int v = 1; v <= nsteps; v++) {
    const double tot = sync.total(v, [&]() { prepare(v); });
    SPEC_ST(0);
    const double h = 0.5 * tot;
    double f1;
    if constexpr (LG) {
      f1 = pre_lin - tot;
      if (A.prior_div != 0.0) f1 = f1 - pre_pri;
      if (A.guard && !fmh_isfinite(f1)) f1 = -fmh_inf();
    } else if (pre_ok && mfr_div_safe(h)) f1 = -pre_nt1 - div_finish(h, pre_ss, pre_rs);   // (finite: the guard has nothing to do)
    else f1 = logpost_of(tot, readlane_d(th1, sig_lane));
    bool st_row = false;
    double st_th0 = 0.0;
    double st_dr = th1;
    SPEC_ST(1);
    if (v == 1) {
      f0 = f1;
      run_sum = (A.win_cont && A.win_sum && fl) ? A.win_sum[(long long)cl * kf + lane] : th0;
      if (1 > burnin) { thin_ctr += 1; if (thin_ctr == thin) { thin_ctr = 0; st_row = true; st_th0 = th0; } }
      if constexpr (BND) sync.second_idle();
    } else if (status != FMCMC_CHAIN_OK) {
      if constexpr (BND) sync.second_idle();
    } else {
      const int i = v;
      if (KIND == FMCMC_KERNEL_RAM) {   // adaptation with f(theta1) of the pending proposal (R/kernel_ram.R:129-152)
        if (ram_gate) {
          double a_n = fmh_exp(f1 - f0);
          if (fmh_isnan(a_n)) a_n = 0.0;
          else if (a_n > 1.0) a_n = 1.0;
          // S <- S T (mh_common.hpp, ram_coef): square root and divisions once per update, lane = column; then row `lane`
          // in registers, two fma per element, column values by v_readlane
          const double eta = pre_eta, Pj1 = pre_Pj1, Pj = pre_Pj, nrm2 = pre_nrm2;
          const double cp = (eta * (a_n - A.arate)) / nrm2;
          if (cp != 0.0 && fmh_isfinite(cp)) {
            double dl, kl;
            const bool okl = ram_coef(cp, Pj, Pj1, zcur, dl, kl);
            if (__any(fl && !okl)) {
              nerr += 1;
            } else {
              static_for<KA>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                if (j < kf) {
                  const double dj = row_bcast<j>(dl), kj = row_bcast<j>(kl);
                  Srow[j] = fmh_fma(Grow[j], kj, Srow[j] * dj);    // (G_ij kept by the proposal; +0 above the diagonal stays +0)
                }
              });
            }
          }
        }
        abs_iter += 1;
        if constexpr (BND) {
          const double th1r = fl ? reflect1(th1, lb_l, ub_l) : th1;
          const bool moved = __any(fl && !(th1r == th1));
          th1 = th1r;
          st_dr = th1r;                               // (the row of draws is the proposal the kernel returns: reflected)
          double tot2 = 0.0;
          const bool again = sync.second(moved, [&]() { if (rl) s_th1[myc * PIPE_KMAX + pj] = th1; }, tot2);
          if (again && moved) {                       // (not moved: the same vector, the same f)
            if constexpr (LG) {                       // (the linear part and the prior term of the REFLECTED proposal: prepare()'s chains)
              double lin = 0.0, ss = 0.0;
              for (int j = 0; j < nbl; j++) {
                const double bj = readlane_d(th1, j);
                lin = fmh_fma(bj, readlane_d(hs_l, j), lin);
                ss = fmh_fma(bj, bj, ss);
              }
              f1 = lin - tot2;
              if (A.prior_div != 0.0) f1 = f1 - ss / A.prior_div;
              if (A.guard && !fmh_isfinite(f1)) f1 = -fmh_inf();
            } else {
              f1 = logpost_of(tot2, readlane_d(th1, sig_lane));
            }
          }
        }
      }
      if (fmh_isnan(f1)) status = FMCMC_CHAIN_NAN_LOGPOST;
      const double ratio = f1 - f0;
      if (status == FMCMC_CHAIN_OK && fmh_isnan(ratio)) status = FMCMC_CHAIN_NAN_RATIO;
      if (status != FMCMC_CHAIN_OK) {
        if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i + ioff; }
        if (rl) A.status_theta[(long long)cl * k + pj] = th1;
        flush_bits(i);
      } else {
        const double lu = lu_nx;
        lu_nx = lu_row[v < nsteps ? v : nsteps - 1];
        if (lu < ratio) {
          th0 = th1;
          f0 = f1;
          nacc += 1;
          bitword |= (1u << ((i - 1) & 31));
        }
        if (i > burnin) { thin_ctr += 1; if (thin_ctr == thin) { thin_ctr = 0; st_row = true; st_th0 = th0; } }
        if (KIND == FMCMC_KERNEL_ADAPT) run_sum = run_sum + th0;
        if (((i - 1) & 31) == 31 || i == nsteps) flush_bits(i);
      }
    }
    SPEC_ST(2);
    // kernel_adapt(freq > 1): row v of the chain into the ring (rows (i - freq) .. (i - 1) are folded in together at step i)
    const int afreq = (KIND == FMCMC_KERNEL_ADAPT && RING) ? A.freq : 1;
    if (KIND == FMCMC_KERNEL_ADAPT && afreq > 1 && ring && fl) ring[(v & (SPEC_FREQMAX - 1)) * PIPE_KMAX + lane] = th0;
    // ---- proposal of loop step i = v + 1
    if (v < nsteps) {
      if (status == FMCMC_CHAIN_OK) {
        const int i = v + 1;
        zcur = z_nx;
        z_nx = fl ? ld_z(v + 1 < nsteps ? v + 1 : nsteps - 1) : 0.0;
        bool sig_dirty = (afreq == 1) || v == 1;         // (freq > 1: the factor is formed again only when Sigma has moved)
        if (KIND == FMCMC_KERNEL_ADAPT) {
          if (A.until > (double)abs_iter && abs_iter > A.warmup && i + ioff > 2 && (afreq == 1 || ((i + ioff) % afreq) == 0)) {   // R/kernel_adapt.R:118-166
            if (afreq > 1 && i - afreq < 1) {
              status = FMCMC_CHAIN_BAD_WINDOW;           // R: ans[0:(i-1), ] has fewer than freq rows, `[, , freq]` is out of bounds
            } else {
              // rows (i - freq):(i - 1) folded in one by one, t = abs_iter - freq + (row - 1) (R/recursive.R:79-108,:129-136)
              for (int jr = 0; jr < afreq; jr++) {
                const double t = (double)(abs_iter - afreq + jr);
                const double x = (afreq == 1) ? th0 : (fl ? ring[((i - afreq + jr) & (SPEC_FREQMAX - 1)) * PIPE_KMAX + lane] : 0.0);
                const double mp = have_mean ? mean_prev : (run_sum / (double)(i + ioff - 1));
                const double mt = (mp * t + x) / (t + 1);
                // (freq = 1: the two step-only quotients were prepared while this wave waited for the partials)
                const double c1 = (afreq == 1) ? pre_c1 : (t - 1) / t, c2 = (afreq == 1) ? pre_c2 : 1.0 / t;
                static_for<KA>([&](auto b_) {
                  constexpr int b = decltype(b_)::value;
                  if (b < kf) {
                    const double mpb = row_bcast<b>(mp), mtb = row_bcast<b>(mt), xb = row_bcast<b>(x);
                    const double ik = (b == lane) ? 1.0 * A.eps : 0.0;
                    const double inner = t * (mp * mpb) - (t + 1) * (mt * mtb) + x * xb + 1e-5 * ik;
                    Srow[b] = c1 * Srow[b] + c2 * inner;
                  }
                });
                mean_prev = mt;
                have_mean = 1;
              }
              sig_dirty = true;
            }
          }
          abs_iter += 1;
          SPEC_ST(3);
          // root-free factor Sigma = L D L^T, left-looking: column j, lane = row (twin of the oracle's ldl_lower_canon; round 5).  The
          // Cholesky factor it replaces put a square root AND a division into every column's dependent chain (5 x ~27 of the
          // owner's ~550 vector instructions at k = 5); here a column is its fma chain and one division, W = the numerators
          // before the division stand in for L D in the sums, and sqrt(D) is one element-wise square root at proposal time.
          bool notpd = false;
          if (!RING || (status == FMCMC_CHAIN_OK && sig_dirty)) {   // (without the stride: every step, no test)
          Dl = 0.0;
          static_for<KA>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            // (a compile-time row count: NO branch per column -- a pivot that fails marks the factor, the columns behind it compute
            //  values nobody reads; every `if (!notpd)` was a scalar branch with its exec bookkeeping in the owner's dependent pass)
            if (j < kf && (KX > 0 || !notpd)) {
              double sacc = Srow[j];
              static_for<j>([&](auto b_) {
                constexpr int b = decltype(b_)::value;
                sacc = fmac_row_bcast<j, true>(sacc, Wrow[KIND == FMCMC_KERNEL_ADAPT ? b : 0], Lrow[b]);   // fma(-L_lane,b, W_jb, sacc)
              });
              const double d = readlane_d(sacc, j);
              const bool badp = !(d > 0.0) || !fmh_isfinite(d);
              if (KX > 0) notpd = notpd || badp;
              if (KX <= 0 && badp) {
                notpd = true;
              } else {
                // selects, not exec-mask regions: every `if (lane ...)` costs three scalar instructions and a branch, and
                // the owner's rate is its instruction count (rows above j keep their 0)
                const double qj = sacc / d;
                Lrow[j] = (lane == j) ? 1.0 : ((lane > j) ? qj : Lrow[j]);
                Wrow[KIND == FMCMC_KERNEL_ADAPT ? j : 0] = sacc;
                Dl = (lane == j) ? d : Dl;
              }
            }
          });
          }
          SPEC_ST(4);
          if (notpd || (RING && status != FMCMC_CHAIN_OK)) {
            if (notpd) status = FMCMC_CHAIN_NOT_PD;
            if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i + ioff; }
            if (rl) A.status_theta[(long long)cl * k + pj] = th1;
            if (((v - 1) & 31) != 31) flush_bits(v);   // (the accept bits of the steps decided so far; a full word has just been flushed)
          } else {
            const double ul = fmh_sqrt(Dl) * zcur;    // (beyond the parameters: D = 0, z = 0)
            double sacc = 0.0;
            static_for<KA>([&](auto b_) {   // (row `lane` of the factor is 0 beyond its diagonal and sacc starts at +0: the terms
              constexpr int b = decltype(b_)::value;
              if (b < kf) sacc = fmac_row_bcast<b>(sacc, ul, Lrow[b]);   // b > lane add +-0 exactly)
            });
            th1 = (KX != 0 || LG || fl) ? reflect1(th0 + (mu_l + sacc), lb_l, ub_l) : th0;
          }
        } else {  // RAM P1 (R/kernel_ram.R:123-126)
          double sacc = 0.0;
          static_for<KA>([&](auto r_) {   // last column first; the partial sums are the G_ib of the factor update
            constexpr int b = KA - 1 - decltype(r_)::value;
            if (b < kf) { Grow[b] = sacc; sacc = fmac_row_bcast<b>(sacc, zcur, Srow[b]); }   // (+0 above the diagonal adds +0)
          });
          th1 = (KX != 0 || LG || fl) ? th0 + sacc : th0;
          ram_gate = (A.until > (double)abs_iter && abs_iter > A.warmup && ((v + 1 + ioff) % A.freq) == 0);
        }
        if (rl) s_th1[myc * PIPE_KMAX + pj] = th1;
      }
      sync.publish(v + 1);
    } else {
      sync.final();
    }
    SPEC_ST(5);
    if (st_row) {   // row v of ans / draws / logpost, off the compute waves' critical path
      if (rl) {
        *reinterpret_cast<double*>(s_base + (lane_off + srow8)) = st_th0;
        if (d_base) *reinterpret_cast<double*>(d_base + (lane_off + srow8)) = st_dr;
      }
      if (l_base && lane == 0) *reinterpret_cast<double*>(l_base + srow8) = f1;
      srow8 += 8;
    }
    SPEC_ST(6);
    if (!SYNC::PREP_EARLY && v < nsteps) prepare(v + 1);
    SPEC_ST(7);
  }
#ifdef SPEC_STAMP
  __builtin_amdgcn_s_waitcnt(0);
  if (lane < 16 && A.draws) A.draws[(long long)cl * 16 + lane] = (lane < 8) ? (double)stt[lane & 7] : (double)nsteps;
#endif
#undef SPEC_ST
  // ---- write state back
  if (rl) A.theta0[(long long)cl * k + pj] = th0;
  if (lane == 0) {
    A.f0[cl] = f0;
    A.accept_count[cl] = nacc;
    if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    A.abs_iter[cl] = abs_iter;
    if (A.nerrors) A.nerrors[cl] = nerr;
    if (KIND == FMCMC_KERNEL_ADAPT) A.have_mean[cl] = have_mean;
  }
#pragma unroll
  for (int b = 0; b < KA; b++)
    if (fl && b < kf) A.Sigma[((long long)cl * kf + lane) * kf + b] = (b <= lane || KIND == FMCMC_KERNEL_ADAPT) ? Srow[b] : 0.0;
  if (KIND == FMCMC_KERNEL_ADAPT && fl) A.mean_prev[(long long)cl * kf + lane] = mean_prev;
  if (KIND == FMCMC_KERNEL_ADAPT && A.win_sum && fl) A.win_sum[(long long)cl * kf + lane] = run_sum;
}

// The mirror kernels' owner (kernel_nmirror / kernel_umirror, R/kernel_mirror.R:66-131, :203-262; twin of the general kernel's
// mirror branch, mh_streamed.hpp, and of the oracle's propose_mirror): joint scheme, no fixed parameter, lane = parameter with
// theta, the running mean mu and the scale in registers.  Until round 4 these kernels ran on the all-family kernel only
// (tools/option_audit.py: 17.8 us per step at C2's shape).
template <int KIND, class SYNC>
__device__ __forceinline__ void mfma_owner_mirror(const SweepArgs& A, int myc, int cl, double* s_th1, SYNC& sync) {
  const int lane = threadIdx.x & 63;
  const int k = A.k, kz = A.kz;
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  const bool rl = lane < k;
  const int jl = rl ? lane : 0;
  const double lb_l = A.lb[jl], ub_l = A.ub[jl];
  double th0 = rl ? A.theta0[(long long)cl * k + lane] : 0.0, th1 = th0;
  double mmu = rl ? (A.fresh ? A.mu[jl] : A.mirror_mu[(long long)cl * k + jl]) : 0.0;
  double msc = rl ? (A.fresh ? A.scale[jl] : A.mirror_scale[(long long)cl * k + jl]) : 0.0;
  double f0 = 0.0, obs_arate = fmh_nan(), th_prev = 0.0;   // (obs_arate: lane = parameter -- R's turns into a k-vector through warm-up; th_prev: ans[i-2, ])
  long long abs_iter = 0, nzero = 0;
  if (!A.fresh) { abs_iter = A.abs_iter[cl]; obs_arate = A.obs_arate[(long long)cl * k + jl]; }
  int nacc = 0, status = FMCMC_CHAIN_OK, thin_ctr = 0;
  unsigned int bitword = 0;
  char* const s_base = reinterpret_cast<char*>(A.samples) + ((long long)cl * k) * A.ldS * 8;
  char* const d_base = A.draws ? reinterpret_cast<char*>(A.draws) + ((long long)cl * k) * A.ldS * 8 : nullptr;
  char* const l_base = A.logpost ? reinterpret_cast<char*>(A.logpost) + (long long)cl * A.ldS * 8 : nullptr;
  const unsigned int lane_off = (unsigned int)((long long)jl * A.ldS * 8);
  unsigned int srow8 = 0;
  const char* const z_base = reinterpret_cast<const char*>(A.fed_z) + ((long long)cl * nsteps) * kz * 8;
  const unsigned int z_lane = (unsigned int)(jl) * 8u;
  const double* const lu_row = A.fed_logu + (long long)cl * nsteps;
  const double dn = uniform_d((double)A.n);
  auto ld_z = [&](int row) -> double { return *reinterpret_cast<const double*>(z_base + (z_lane + (unsigned int)row * (unsigned int)(kz * 8))); };
  double z_nx = (rl && nsteps >= 2) ? ld_z(1) : 0.0;
  double lu_nx = (nsteps >= 2) ? lu_row[1] : 0.0;
  auto flush_bits = [&](int i) {
    if (A.accept_bits && lane == 0) A.accept_bits[(long long)cl * ((nsteps + 31) >> 5) + ((i - 1) >> 5)] = bitword;
    bitword = 0;
  };
  auto logpost_of = [&](double tot, double sigma) -> double {   // Gaussian linreg closed form (as the other owners')
    double f;
    if (sigma < 0.0 || fmh_isnan(sigma)) f = fmh_nan();
    else if (sigma == 0.0) f = -fmh_inf();
    else {
      double t1 = fmh_log(sigma) + FMH_K(FMH_LN_SQRT_2PI);
      double q = (0.5 * tot) / (sigma * sigma);
      f = -(dn * t1) - q;
    }
    if (A.guard && !fmh_isfinite(f)) f = -fmh_inf();
    return f;
  };
  // What a step needs and the evaluation does not enter, formed while the owner waits for the total (round 5, as the register owner's
  // prepare()): the sigma-only half of the closed form with the reciprocal of sigma^2 (div_finish: bit for bit the division), and the
  // reciprocal of abs_iter + 1, the denominator of the two running means of the proposal.
  double pre_nt1 = 0.0, pre_ss = 1.0, pre_rs = 1.0, pre_den = 1.0, pre_rd = 1.0;
  bool pre_ok = false;
  auto prepare = [&]() {
    const double sigma = readlane_d(th1, k - 1);
    const unsigned sg_hi = (unsigned)(fmh_d2u(sigma) >> 32);
    const bool sg_fast = (sg_hi - 0x00100000u) < 0x7fe00000u;            // positive, finite, normal
    const double sg = sg_fast ? sigma : 1.0;
    pre_nt1 = dn * (fmh_log(sg) + FMH_K(FMH_LN_SQRT_2PI));
    pre_ss = sg * sg;
    pre_ok = sg_fast && mfr_div_safe(pre_ss);
    pre_rs = div_recip(pre_ok ? pre_ss : 1.0);
    pre_den = (double)abs_iter + 1;
    pre_rd = div_recip(pre_den);
  };
  auto over_den = [&](double num) -> double {   // num / (abs_iter + 1)
    if (__all(!rl || mfr_div_safe(num))) return div_finish(num, pre_den, pre_rd);
    return num / pre_den;
  };
  if (!SYNC::PREP_EARLY) prepare();
  for (int v = 1; v <= nsteps; v++) {
    const double tot = sync.total(v, [&]() { prepare(); });
    const double hq = 0.5 * tot;
    const double f1 = (pre_ok && mfr_div_safe(hq)) ? (-pre_nt1 - div_finish(hq, pre_ss, pre_rs)) : logpost_of(tot, readlane_d(th1, k - 1));
    bool st_row = false;
    double st_th0 = 0.0;
    const double st_dr = th1;
    if (v == 1) {
      f0 = f1;
      if (1 > burnin) { thin_ctr += 1; if (thin_ctr == thin) { thin_ctr = 0; st_row = true; st_th0 = th0; } }
    } else if (status == FMCMC_CHAIN_OK) {
      const int i = v;
      if (fmh_isnan(f1)) status = FMCMC_CHAIN_NAN_LOGPOST;
      const double ratio = f1 - f0;
      if (status == FMCMC_CHAIN_OK && fmh_isnan(ratio)) status = FMCMC_CHAIN_NAN_RATIO;
      if (status != FMCMC_CHAIN_OK) {
        if (lane == 0) { A.status[cl] = status; A.status_step[cl] = i; }
        if (rl) A.status_theta[(long long)cl * k + lane] = th1;
        flush_bits(i);
      } else {
        const double lu = lu_nx;
        lu_nx = lu_row[v < nsteps ? v : nsteps - 1];
        bool moved = false;
        th_prev = th0;                          // (row i - 1, the row before the one decided now)
        if (lu < ratio) {
          const double d = th1 - th0;           // rowSums(diff(ans)^2) != 0 of the row about to be stored: a sum of squares is 0 exactly
          moved = __any(rl && (d * d != 0.0));  // when every term is (a NaN term makes it NaN: != 0 either way)
          th0 = th1;
          f0 = f1;
          nacc += 1;
          bitword |= (1u << ((i - 1) & 31));
        }
        if (!moved) nzero += 1;
        if (i > burnin) { thin_ctr += 1; if (thin_ctr == thin) { thin_ctr = 0; st_row = true; st_th0 = th0; } }
        if (((i - 1) & 31) == 31 || i == nsteps) flush_bits(i);
      }
    }
    // ---- proposal of loop step i = v + 1
    if (v < nsteps) {
      if (status == FMCMC_CHAIN_OK) {
        const int i = v + 1;
        const double z = z_nx;
        z_nx = rl ? ld_z(v + 1 < nsteps ? v + 1 : nsteps - 1) : 0.0;
        if (abs_iter >= 1 && abs_iter <= A.warmup) mmu = over_den(mmu * (double)abs_iter + th0);   // mean_recursive(ans[i-1, ], mu, abs_iter)
        if (abs_iter == A.nadapt) {   // the one-off scale adaptation (the closure reads its argument `nadapt`)
          obs_arate = 1.0 - (double)nzero / (double)(i - 2);
          const double num = fmh_tan_0_halfpi(1.5707963267948966 * obs_arate);
          const double den = fmh_tan_0_halfpi(1.5707963267948966 * A.arate);
          msc = msc * num / den;
        } else if (abs_iter > A.nadapt && abs_iter <= A.warmup) {
          // obs_arate <<- mean_recursive(as.double(ans[i-1, ] != ans[i-2, ]), obs_arate, abs_iter), element-wise (R/kernel_mirror.R:108-118,
          // :246-253); the first proposal of a call has no ans[i-2, ]: numeric(0) in R, NaN here (twin of the oracle's propose_mirror)
          obs_arate = (i < 3) ? fmh_nan() : over_den(obs_arate * (double)abs_iter + ((th0 != th_prev) ? 1.0 : 0.0));
        }
        double t;
        if (KIND == FMCMC_KERNEL_NMIRROR) {
          t = (2.0 * mmu - th0) + msc * z;
        } else {   // runif(k, 2 mu - theta -+ sqrt3 scale)
          const double sqrt3 = fmh_sqrt(3.0);
          const double c = 2.0 * mmu - th0;
          const double lo = c - sqrt3 * msc, hi = c + sqrt3 * msc;
          t = lo + (hi - lo) * z;
        }
        th1 = reflect1(t, lb_l, ub_l);
        abs_iter += 1;
        if (rl) s_th1[myc * PIPE_KMAX + lane] = th1;
      }
      sync.publish(v + 1);
    } else {
      sync.final();
    }
    if (st_row) {
      if (rl) {
        *reinterpret_cast<double*>(s_base + (lane_off + srow8)) = st_th0;
        if (d_base) *reinterpret_cast<double*>(d_base + (lane_off + srow8)) = st_dr;
      }
      if (l_base && lane == 0) *reinterpret_cast<double*>(l_base + srow8) = f1;
      srow8 += 8;
    }
    if (!SYNC::PREP_EARLY && v < nsteps) prepare();
  }
  if (rl) {
    A.theta0[(long long)cl * k + lane] = th0;
    A.mirror_mu[(long long)cl * k + lane] = mmu;
    A.mirror_scale[(long long)cl * k + lane] = msc;
    A.obs_arate[(long long)cl * k + lane] = obs_arate;
  }
  if (lane == 0) {
    A.f0[cl] = f0;
    A.accept_count[cl] = nacc;
    if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
    A.abs_iter[cl] = abs_iter;
  }
}

// The compute role of mh_sweep_spec for OPT (even) observation slots of P covariates per lane: one instantiation per slot count,
// selected at run time by the kernel (round 4: the kernel used to exist for n in (9728, 10240] at p = 3 and (512, 1024] at p = 1
// only, every other shape fell to the general kernel).
template <int P, int OPT, bool BNDC = false>
__device__ __forceinline__ void spec_compute(const SweepArgs& A, const double* s_y, const double* s_th1, unsigned* s_ready, unsigned* s_done,
                                             double* s_tr, int ncw, int nsteps, int ic, bool dbg, int wave, int tid, int lane,
                                             const unsigned* = nullptr) {
    double xr[OPT][P > 0 ? P : 1];
    double wlast = 1.0, wprev = 1.0;   // validity of this lane's observation in the last two slots (an odd slot count leaves the last one empty)
#pragma unroll
    for (int s = 0; s < OPT; s++) {
      const long long i = (long long)tid + (long long)NT * s;
      const bool valid = i < A.n;
#pragma unroll
      for (int j = 0; j < P; j++) xr[s][j] = valid ? A.X[(long long)j * A.n + i] : 0.0;
      if (s == OPT - 1) wlast = valid ? 1.0 : 0.0;
      if (s == OPT - 2) wprev = valid ? 1.0 : 0.0;
    }
    const int tr_slot = (tid & 7) * PIPE_TRS + (tid >> 3);
    const double2* yp = reinterpret_cast<const double2*>(s_y) + tid;
    unsigned long long tw = 0, te = 0;
    // (BNDC, the bounded kernel_ram: numbered requests per chain, SpecSyncB -- a round serves the next request of every live chain)
    unsigned srv0 = 0u, srv1 = 0u, srv2 = 0u, srv3 = 0u;
    unsigned live = BNDC ? ((1u << ncw) - 1u) : 0u;
    for (int v = 1; BNDC ? (live != 0u) : (v <= nsteps); v++) {
      for (int c = 0; c < ncw; c++) {
        unsigned long long t_a = dbg ? clk() : 0;
        if constexpr (BNDC) {
          if (!((live >> c) & 1u)) continue;
          const unsigned mine = c == 0 ? srv0 : (c == 1 ? srv1 : (c == 2 ? srv2 : srv3));
          unsigned w;
          while ((w = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_ld_u32(&s_ready[c]))) <= mine) __builtin_amdgcn_s_sleep(1);
          if (w == SPEC_FINAL) { live &= ~(1u << c); continue; }
          srv0 += (c == 0) ? 1u : 0u; srv1 += (c == 1) ? 1u : 0u; srv2 += (c == 2) ? 1u : 0u; srv3 += (c == 3) ? 1u : 0u;
        } else {
          while (lds_ld_u32(&s_ready[c]) < (unsigned)v) __builtin_amdgcn_s_sleep(1);
        }
        unsigned long long t_b = dbg ? clk() : 0;
        const double* t0 = s_th1 + c * PIPE_KMAX;
        const double m00 = ic ? t0[0] : 0.0;
        double b0[P > 0 ? P : 1];
#pragma unroll
        for (int j = 0; j < P; j++) b0[j] = t0[ic + j];
        double a0 = 0.0;
        // y pairs come from LDS three pairs (~30 FMAs) ahead of their use: LDS latency is ~130 cycles and only two
        // compute waves share the SIMD, so a one-pair lookahead leaves the FMA pipe waiting on lgkmcnt
        constexpr int YD = 3;
        double2 yq[YD];
#pragma unroll
        for (int d = 0; d < YD; d++) yq[d] = yp[(d < OPT / 2 ? d : OPT / 2 - 1) * NT];
#pragma unroll
        for (int s2 = 0; s2 < OPT / 2; s2++) {
          const double2 yy = yq[s2 % YD];
          if (s2 + YD < OPT / 2) yq[s2 % YD] = yp[(s2 + YD) * NT];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int s = 2 * s2 + h;
            const double yv = h ? yy.y : yy.x;
            double m0 = m00;
#pragma unroll
            for (int j = 0; j < P; j++) m0 = fmh_fma(xr[s][j], b0[j], m0);
            const double r0 = yv - m0;
            if (s == OPT - 1) a0 = fmh_fma(r0 * wlast, r0, a0);
            else if (s == OPT - 2) a0 = fmh_fma(r0 * wprev, r0, a0);   // (r0 * 1 == r0: the same bits where the slot is full)
            else a0 = fmh_fma(r0, r0, a0);
          }
        }
        s_tr[c * (8 * PIPE_TRS) + tr_slot] = a0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // partials landed before the arrival is visible
        if (lane == 0) __hip_atomic_fetch_add(&s_done[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (dbg) { unsigned long long t_c = clk(); tw += t_b - t_a; te += t_c - t_b; }
      }
    }
    if (dbg && lane == 0 && A.draws) {
      double* d = A.draws + ((long long)blockIdx.x * 12 + wave) * 4;
      d[0] = (double)tw; d[1] = (double)te; d[2] = 0.0; d[3] = (double)nsteps;
    }
}

// The compute role for the LOGISTIC family (round 5: the workflow vignette's own model -- mcmc::logit, 100 observations -- ran on the
// general kernel at 2.6 / 5.9 us per step, kernel_normal / kernel_adapt): a lane's OPT observations in VGPRs as above, per observation
// 64 eta as the fma chain over the scaled coefficients and g(|eta|) off the table in LDS (logit_g_vec: the oracle's fmh_logit_g_scaled),
// added in slot order; the linear part sum_j b_j hs_j and the prior term never enter a loop (the owners form them).  y is not read.
template <int P, int OPT, bool BNDC = false>
__device__ __forceinline__ void spec_compute_logit(const SweepArgs& A, const double* s_tab, const double* s_th1, unsigned* s_ready, unsigned* s_done,
                                                   double* s_tr, int ncw, int nsteps, int ic, int wave, int tid, int lane) {
    double xr[OPT][P > 0 ? P : 1];
    double wlast = 1.0, wprev = 1.0;
#pragma unroll
    for (int s = 0; s < OPT; s++) {
      const long long i = (long long)tid + (long long)NT * s;
      const bool valid = i < A.n;
#pragma unroll
      for (int j = 0; j < P; j++) xr[s][j] = valid ? A.X[(long long)j * A.n + i] : 0.0;
      if (s == OPT - 1) wlast = valid ? 1.0 : 0.0;
      if (s == OPT - 2) wprev = valid ? 1.0 : 0.0;
    }
    const int tr_slot = (tid & 7) * PIPE_TRS + (tid >> 3);
    unsigned srv0 = 0u, srv1 = 0u, srv2 = 0u, srv3 = 0u;         // (BNDC, the bounded kernel_ram: numbered requests per chain, as spec_compute)
    unsigned live = BNDC ? ((1u << ncw) - 1u) : 0u;
    for (int v = 1; BNDC ? (live != 0u) : (v <= nsteps); v++) {
      for (int c = 0; c < ncw; c++) {
        if constexpr (BNDC) {
          if (!((live >> c) & 1u)) continue;
          const unsigned mine = c == 0 ? srv0 : (c == 1 ? srv1 : (c == 2 ? srv2 : srv3));
          unsigned w;
          while ((w = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_ld_u32(&s_ready[c]))) <= mine) __builtin_amdgcn_s_sleep(1);
          if (w == SPEC_FINAL) { live &= ~(1u << c); continue; }
          srv0 += (c == 0) ? 1u : 0u; srv1 += (c == 1) ? 1u : 0u; srv2 += (c == 2) ? 1u : 0u; srv3 += (c == 3) ? 1u : 0u;
        } else {
          while (lds_ld_u32(&s_ready[c]) < (unsigned)v) __builtin_amdgcn_s_sleep(1);
        }
        const double* t0 = s_th1 + c * PIPE_KMAX;
        const double m00 = ic ? t0[0] * FMH_LG_SCALE : 0.0;      // (times 64: exact)
        double b0[P > 0 ? P : 1];
#pragma unroll
        for (int j = 0; j < P; j++) b0[j] = t0[ic + j] * FMH_LG_SCALE;
        double a0 = 0.0;
        constexpr int G = 4;
#pragma unroll
        for (int s0 = 0; s0 < OPT; s0 += G) {
          constexpr int GG = G;
          double us[GG], gv[GG];
#pragma unroll
          for (int u = 0; u < GG; u++) {
            double es = m00;
            if (s0 + u < OPT) {
#pragma unroll
              for (int j = 0; j < P; j++) es = fmh_fma(xr[s0 + u][j], b0[j], es);
            }
            us[u] = __builtin_fabs(es);
          }
          logit_g_vec<GG, true>(us, gv, s_tab);
#pragma unroll
          for (int u = 0; u < GG; u++) {
            const int s = s0 + u;
            if (s < OPT) {
              if (s == OPT - 1) a0 = a0 + gv[u] * wlast;        // (g * 1 == g; a slot without an observation adds +0)
              else if (s == OPT - 2) a0 = a0 + gv[u] * wprev;
              else a0 = a0 + gv[u];
            }
          }
        }
        s_tr[c * (8 * PIPE_TRS) + tr_slot] = a0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // partials landed before the arrival is visible
        if (lane == 0) __hip_atomic_fetch_add(&s_done[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
}

// OPTMAX: the most observation slots a compute lane holds (x in VGPRs: OPTMAX P doubles); the launch's (even) slot count
// A.spec_opt <= OPTMAX selects the compute loop.  FAM: the Gaussian linear model, or (round 5) the logistic one -- the g table in LDS
// where the linear model keeps y, the closed form sum_j b_j hs_j - total - prior in the owners.
// RING: kernel_adapt(freq = 2 .. 8) -- its own kernel (k_spec_r.hip): as one more owner inside the freq = 1 kernel it cost that
// kernel's owner loop scalar-register spills (kernel_adapt in the latency form at n = 10,000: 2.20 -> 2.33 us per step).
template <int P, int OPTMAX, int KIND, int FAM = FMCMC_FAM_GAUSSIAN_LINREG, bool RING = false>
__global__ __launch_bounds__(SPEC_NT) void mh_sweep_spec(const SweepArgs A) {
  static_assert(!RING || KIND == FMCMC_KERNEL_ADAPT, "RING is kernel_adapt(freq > 1)");
  constexpr bool LG = FAM == FMCMC_FAM_LOGISTIC;
  constexpr int CW = 4;
  static_assert(OPTMAX % 2 == 0, "slot counts are even (y is read back in pairs)");
  const int OPT = A.spec_opt;                      // (uniform) even, 2 .. OPTMAX
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = A.k, kz = A.kz;
  double* s_th1 = smem;                            // [CW][PIPE_KMAX] proposals read by the evaluation
  double* s_par = s_th1 + CW * PIPE_KMAX;          // [4][PIPE_KMAX]  mu, scale, lb, ub
  unsigned* s_ready = (unsigned*)(s_par + 4 * PIPE_KMAX);  // [CW] version of theta1[c] that is published
  unsigned* s_done = s_ready + CW;                         // [CW] partial arrivals (8 per version)
  double* s_tr = s_par + 4 * PIPE_KMAX + CW;       // [CW][8][PIPE_TRS] lane partials, transposed
  double* s_y = s_tr + CW * 8 * PIPE_TRS;          // [OPT/2][NT][2] this workgroup's copy of y (logistic: the g table, 16-byte aligned)
  double* s_ad = s_y + (LG ? LG_LDS_DOUBLES + 2 : OPT * NT);   // KIND >= 3: [CW][SPEC_ADS] adaptive per-chain state
  double* const s_tab = LG ? logit_table_align(s_y) : nullptr;
  const bool s_need = KIND == FMCMC_KERNEL_RAM && A.ram_bounded;   // the bounded kernel_ram: numbered requests (SpecSyncB)
  // chains of this workgroup: A.spec_cw = 4, or 2 / 1 in the LATENCY form (fewer than 4 x CUs chains per GPU: every chain gets
  // more of a compute unit -- all eight compute waves evaluate the one or two chains there are, and an owner's turn-around
  // is no longer queued behind the evaluation of three other chains; same canonical lanes, same tree, same bits)
  const int cwl = A.spec_cw;
  const long long cg0 = (long long)blockIdx.x * cwl;
  const int ncw = (int)((A.nchains - cg0 < cwl) ? (A.nchains - cg0) : cwl);
  const int nsteps = (int)A.nsteps, burnin = (int)A.burnin, thin = (int)A.thin;
  const int ic = A.intercept;
  const bool dbg = (A.debug & 8) != 0;

  // ---- cooperative set-up (all 12 waves): y -> LDS, kernel constants, initial theta1, flags
  if constexpr (LG) {
    logit_stage_table(s_tab);
  } else {
    for (int e = tid; e < OPT * NT; e += SPEC_NT) {
      const int s = e / NT, t = e - s * NT;
      const long long i = (long long)t + (long long)NT * s;
      s_y[((s >> 1) * NT + t) * 2 + (s & 1)] = (i < A.n) ? A.y[i] : 0.0;
    }
  }
  if (tid < k) {
    s_par[0 * PIPE_KMAX + tid] = A.mu[tid];
    s_par[1 * PIPE_KMAX + tid] = A.scale[tid];
    s_par[2 * PIPE_KMAX + tid] = A.lb[tid];
    s_par[3 * PIPE_KMAX + tid] = A.ub[tid];
  }
  if (tid < CW * PIPE_KMAX) {
    const int c = tid / PIPE_KMAX, j = tid - c * PIPE_KMAX;
    s_th1[tid] = (c < ncw && j < k) ? A.theta0[(cg0 + c) * k + j] : 0.0;
  }
  if (tid < CW) { s_ready[tid] = 1u; s_done[tid] = 0u; }
  __syncthreads();

  if (wave < SPEC_NCW) {
    // =========================== COMPUTE ROLE ===========================
    switch (OPT) {
#define SPEC_CC(O_) case O_: if constexpr (O_ <= OPTMAX) { if constexpr (LG) { if (KIND == FMCMC_KERNEL_RAM && s_need) { if constexpr (KIND == FMCMC_KERNEL_RAM) spec_compute_logit<P, O_, true>(A, s_tab, s_th1, s_ready, s_done, s_tr, ncw, nsteps, ic, wave, tid, lane); } \
                                                                             else spec_compute_logit<P, O_>(A, s_tab, s_th1, s_ready, s_done, s_tr, ncw, nsteps, ic, wave, tid, lane); } \
                                                            else if (KIND == FMCMC_KERNEL_RAM && s_need) { if constexpr (KIND == FMCMC_KERNEL_RAM) spec_compute<P, O_, true>(A, s_y, s_th1, s_ready, s_done, s_tr, ncw, nsteps, ic, dbg, wave, tid, lane); } \
                                                            else spec_compute<P, O_>(A, s_y, s_th1, s_ready, s_done, s_tr, ncw, nsteps, ic, dbg, wave, tid, lane); } break;
      SPEC_CC(2) SPEC_CC(4) SPEC_CC(6) SPEC_CC(8) SPEC_CC(10) SPEC_CC(12) SPEC_CC(14) SPEC_CC(16) SPEC_CC(18) SPEC_CC(20)
#undef SPEC_CC
      default: break;
    }
    return;
  }

  // =========================== OWNER ROLE ===========================
  const int myc = wave - SPEC_NCW;
  if (myc >= ncw) return;
  // Owners issue ahead of the compute waves of their SIMD (round 3).  The s_memtime shares of an adaptive owner's phases
  // (tools/exp_spec.hip -DSPEC_STAMP) showed every one of its instructions waiting ~26 cycles: two compute waves keep the
  // SIMD's fp64 pipe full and the arbiter serves the oldest wave first.  With the priority raised its work per step falls
  // from 8760 to 5270 ticks (the compute waves fill what it leaves): C3 3.55 -> 3.34 us per step, kernel_ram k = 5
  // 3.67 -> 3.41; priority 3 measures the same as 1.
#ifndef SPEC_OWNER_PRIO
#define SPEC_OWNER_PRIO 1
#endif
  if (SPEC_OWNER_PRIO > 0) __builtin_amdgcn_s_setprio(SPEC_OWNER_PRIO);
  const int cl = __builtin_amdgcn_readfirstlane((int)cg0 + myc);
  if constexpr (KIND == FMCMC_KERNEL_NMIRROR || KIND == FMCMC_KERNEL_UMIRROR) {   // (joint scheme, no fixed parameter: the host's conditions)
    SpecSync sync{s_ready, s_done, s_tr, myc};
    mfma_owner_mirror<KIND>(A, myc, cl, s_th1, sync);
    return;
  }
  if constexpr (KIND == FMCMC_KERNEL_ADAPT || KIND == FMCMC_KERNEL_RAM) {
    bool nofixed = true;
    for (int j = 0; j < k; j++) nofixed = nofixed && (A.fixed[j] == 0);
    SpecSync sync{s_ready, s_done, s_tr, myc};
    if constexpr (LG) {   // (the host takes this kernel for k = P + intercept, no fixed parameter: the register owner)
      if constexpr (KIND == FMCMC_KERNEL_RAM && P <= 7) {   // (8 .. 15 covariates: no bounded kernel_ram here, the host's condition)
        if (s_need) {
          SpecSyncB syncb{s_ready, s_done, s_tr, myc};
          if (k == P + 1) spec_owner_adaptive_reg<KIND, P + 1, SpecSyncB, true, FMCMC_FAM_LOGISTIC>(A, myc, cl, s_th1, syncb);
          else spec_owner_adaptive_reg<KIND, (P > 0 ? P : 1), SpecSyncB, true, FMCMC_FAM_LOGISTIC>(A, myc, cl, s_th1, syncb);
          return;
        }
      }
      if constexpr (RING) {   // (the stride: the generic-width owner with the ring of the chain's last rows)
        spec_owner_adaptive_reg<KIND, 0, SpecSync, false, FMCMC_FAM_LOGISTIC, true>(A, myc, cl, s_th1, sync, s_ad + myc * SPEC_ADS);
        return;
      }
      if (k == P + 1) spec_owner_adaptive_reg<KIND, P + 1, SpecSync, false, FMCMC_FAM_LOGISTIC>(A, myc, cl, s_th1, sync);
      else spec_owner_adaptive_reg<KIND, (P > 0 ? P : 1), SpecSync, false, FMCMC_FAM_LOGISTIC>(A, myc, cl, s_th1, sync);
      return;
    }
    if constexpr (P >= 8) {
      // 8 .. 14 covariates on up to 2048 observations (round 5): the register owner at the compile-time width k = P + 2 / P + 1 <= 16 -- the
      // owner waves of THIS kernel hold no operands (rows of 16 lost their registers to the operand groups in mh_sweep_mfma_ad); a fixed
      // parameter: the owners with their matrices in LDS.  (No bounded kernel_ram, no stride here: the host's conditions.)
      if (k == P + 2 && nofixed && A.kz == k) spec_owner_adaptive_reg<KIND, P + 2>(A, myc, cl, s_th1, sync);
      else if (k == P + 1 && nofixed && A.kz == k) spec_owner_adaptive_reg<KIND, P + 1>(A, myc, cl, s_th1, sync);
      else spec_owner_adaptive<KIND>(A, myc, cl, s_th1, s_par, sync, s_ad + myc * SPEC_ADS);
      return;
    }
    if constexpr (KIND == FMCMC_KERNEL_RAM) {
      if (s_need) {   // the bounded kernel_ram (the host takes this kernel for k <= 8, no fixed parameter)
        SpecSyncB syncb{s_ready, s_done, s_tr, myc};
        if (k == P + 2) spec_owner_adaptive_reg<KIND, P + 2, SpecSyncB, true>(A, myc, cl, s_th1, syncb);
        else spec_owner_adaptive_reg<KIND, 0, SpecSyncB, true>(A, myc, cl, s_th1, syncb);
        return;
      }
    }
    if constexpr (RING) {   // (the stride: the generic-width owner with the ring of the chain's last rows; k <= 8, none fixed: the host's conditions)
      spec_owner_adaptive_reg<KIND, 0, SpecSync, false, FMCMC_FAM_GAUSSIAN_LINREG, true>(A, myc, cl, s_th1, sync, s_ad + myc * SPEC_ADS);
      return;
    }
    if (k == P + 2 && nofixed && A.kz == k && !(A.debug & 16))        // intercept + P covariates + sigma (C3: k = 5)
      spec_owner_adaptive_reg<KIND, P + 2>(A, myc, cl, s_th1, sync);
    else if (k == P + 1 && nofixed && A.kz == k && !(A.debug & 16))   // no intercept
      spec_owner_adaptive_reg<KIND, P + 1>(A, myc, cl, s_th1, sync);
    else if (k <= SPEC_KA && A.kz >= 1 && !(A.debug & 16))       // (run-time width: fixed parameters too)
      spec_owner_adaptive_reg<KIND, 0>(A, myc, cl, s_th1, sync);
    else
      spec_owner_adaptive<KIND>(A, myc, cl, s_th1, s_par, sync, s_ad + myc * SPEC_ADS);
    return;
  }
  const bool plane = (lane < k);
  const int jl = plane ? lane : 0;
  const bool fixed_l = A.fixed[jl] != 0;
  int zidx = 0;
  for (int j = 0; j < jl; j++) zidx += A.fixed[j] ? 0 : 1;
  double th0 = plane ? A.theta0[(long long)cl * k + lane] : 0.0;
  double th1 = th0;
  double f0 = 0.0;
  // (a continuation window of a long call takes over what the windows before it left: SweepArgs.win_cont, mh_common.hpp)
  int nacc = 0;                               // (per launch: launch_sweep adds the windows up)
  int status = A.win_cont ? A.status[cl] : FMCMC_CHAIN_OK, thin_ctr = A.thin_ctr0;
  unsigned int bitword = 0;
  // (row stores and variates: wave-uniform 64-bit bases of the chain's own blocks + 32-bit offsets -- 32-bit offsets from the
  //  BUFFER bases capped a call at 4 GiB)
  // (wave-uniform 64-bit bases -- the chain's block -- plus a 32-bit byte offset per lane: one scalar-base store each)
  char* const s_base = reinterpret_cast<char*>(A.samples) + ((long long)cl * k) * A.ldS * 8;
  char* const d_base = A.draws ? reinterpret_cast<char*>(A.draws) + ((long long)cl * k) * A.ldS * 8 : nullptr;
  char* const l_base = A.logpost ? reinterpret_cast<char*>(A.logpost) + (long long)cl * A.ldS * 8 : nullptr;
  const unsigned int lane_off = (unsigned int)((long long)jl * A.ldS * 8);
  unsigned int srow8 = 0;
  const char* const z_base = reinterpret_cast<const char*>(A.fed_z) + ((long long)cl * nsteps) * kz * 8;   // wave-uniform
  const unsigned int z_lane = (unsigned int)(zidx) * 8u;
  const double* const lu_row = A.fed_logu + (long long)cl * nsteps;
  const double dn = uniform_d((double)A.n);
  auto ld_z = [&](int row) -> double {
    return *reinterpret_cast<const double*>(z_base + (z_lane + (unsigned int)row * (unsigned int)(kz * 8)));
  };
  double z_nx = (plane && !fixed_l && nsteps >= 2) ? ld_z(1) : 0.0;   // variates of the NEXT proposal / decision
  double lu_nx = (nsteps >= 2) ? lu_row[1] : 0.0;

  auto logpost_of = [&](double tot, double sigma) -> double {
    double f;
    if (sigma < 0.0 || fmh_isnan(sigma)) {
      f = fmh_nan();
    } else if (sigma == 0.0) {
      f = -fmh_inf();
    } else {
      double t1 = fmh_log(sigma) + FMH_K(FMH_LN_SQRT_2PI);
      double q = (0.5 * tot) / (sigma * sigma);
      f = -(dn * t1) - q;
    }
    if (A.guard && !fmh_isfinite(f)) f = -fmh_inf();
    return f;
  };
  auto flush_bits = [&](int i) {
    if (A.accept_bits && lane == 0) {   // (bits_stride: words per chain of the whole call's bitmap, set by launch_sweep for every launch)
      unsigned int* w = A.accept_bits + ((long long)cl * A.bits_stride + ((i - 1) >> 5));
      // the first word of a continuation window also holds the last bit of the window before it
      *w = (A.win_cont && i <= 32) ? (*w | bitword) : bitword;
    }
    bitword = 0;
  };

  // logistic: sum_j b_j hs_j and the prior term of the pending proposal, formed while the compute waves evaluate it
  const int nbl = A.intercept + A.p;
  const double hs_l = (LG && lane < nbl && A.lg_hs) ? A.lg_hs[lane] : 0.0;
  double lin1 = 0.0, pri1 = 0.0;
  unsigned long long tw = 0, tp = 0, tst = 0;
  for (int v = 1; v <= nsteps; v++) {
    if constexpr (LG) {   // (finish_logpost<LOGISTIC>: the same fma chains, the same bits)
      double lin = 0.0, ss = 0.0;
      for (int j = 0; j < nbl; j++) {
        const double bj = readlane_d(th1, j);
        lin = fmh_fma(bj, readlane_d(hs_l, j), lin);
        ss = fmh_fma(bj, bj, ss);
      }
      lin1 = lin;
      pri1 = (A.prior_div != 0.0) ? ss / A.prior_div : 0.0;
    }
    // ---- wait for the 8 compute waves' partials of version v
    unsigned long long t_a = dbg ? clk() : 0;
    while (lds_ld_u32(&s_done[myc]) < 8u * (unsigned)v) __builtin_amdgcn_s_sleep(1);
    unsigned long long t_b = dbg ? clk() : 0;
    const double* src = s_tr + myc * (8 * PIPE_TRS) + lane;  // this lane folds canonical lanes 8*lane .. 8*lane+7
    const double v0 = src[0 * PIPE_TRS], v1 = src[1 * PIPE_TRS], v2 = src[2 * PIPE_TRS], v3 = src[3 * PIPE_TRS];
    const double v4 = src[4 * PIPE_TRS], v5 = src[5 * PIPE_TRS], v6 = src[6 * PIPE_TRS], v7 = src[7 * PIPE_TRS];
    const double tot = wave_xor_sum(((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)));
    double f1;
    if constexpr (LG) {
      f1 = lin1 - tot;
      if (A.prior_div != 0.0) f1 = f1 - pri1;
      if (A.guard && !fmh_isfinite(f1)) f1 = -fmh_inf();
    } else {
      f1 = logpost_of(tot, readlane_d(th1, k - 1));
    }
    const double th1_eval = th1;
    bool keep_row = false;
    if (v == 1) {                       // row 1: f0 = f(initial)
      f0 = uniform_d(f1);
      keep_row = true;
    } else if (status == FMCMC_CHAIN_OK) {
      const double ratio = f1 - f0;
      if (fmh_isnan(f1) || fmh_isnan(ratio)) {
        status = fmh_isnan(f1) ? FMCMC_CHAIN_NAN_LOGPOST : FMCMC_CHAIN_NAN_RATIO;
        if (lane == 0) { A.status[cl] = status; A.status_step[cl] = v + A.step_off; }
        if (plane) A.status_theta[(long long)cl * k + lane] = th1;
        flush_bits(v);
      } else {
        const double lu = lu_nx;
        lu_nx = lu_row[v < nsteps ? v : nsteps - 1];   // log u of step v + 1 (clamped), consumed a step later
        if (lu < ratio) {
          th0 = th1;
          f0 = uniform_d(f1);
          nacc += 1;
          bitword |= (1u << ((v - 1) & 31));
        }
        keep_row = true;
      }
    }
    const double th0_row = th0;
    // ---- proposal of step v + 1, published for the compute waves
    if (v < nsteps) {
      if (status == FMCMC_CHAIN_OK && plane) {
        double t = th0;
        if (!fixed_l) {
          t = th0 + (s_par[0 * PIPE_KMAX + lane] + s_par[1 * PIPE_KMAX + lane] * z_nx);
          if (KIND == FMCMC_KERNEL_NORMAL_REFLECTIVE) t = reflect1(t, s_par[2 * PIPE_KMAX + lane], s_par[3 * PIPE_KMAX + lane]);
          z_nx = ld_z(v + 1 < nsteps ? v + 1 : nsteps - 1);   // row of step v + 2 (clamped), awaited a step later
        }
        th1 = t;
        s_th1[myc * PIPE_KMAX + lane] = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&s_ready[myc], (unsigned)(v + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    unsigned long long t_c = dbg ? clk() : 0;
    // ---- stores last (off the compute waves' critical path)
    if (keep_row && v > burnin && !(dbg && A.draws)) {
      thin_ctr += 1;
      if (thin_ctr == thin) {
        thin_ctr = 0;
        if (plane) {
          *reinterpret_cast<double*>(s_base + (lane_off + srow8)) = th0_row;
          if (d_base) *reinterpret_cast<double*>(d_base + (lane_off + srow8)) = th1_eval;
        }
        if (l_base && lane == 0) *reinterpret_cast<double*>(l_base + srow8) = f1;
        srow8 += 8;
      }
    }
    if (status == FMCMC_CHAIN_OK && v >= 2 && (((v - 1) & 31) == 31 || v == nsteps)) flush_bits(v);
    if (dbg) { unsigned long long t_d = clk(); tw += t_b - t_a; tp += t_c - t_b; tst += t_d - t_c; }
  }
  if (dbg && lane == 0 && A.draws) {
    double* d = A.draws + ((long long)blockIdx.x * 12 + wave) * 4;
    d[0] = (double)tw; d[1] = (double)tp; d[2] = (double)tst; d[3] = (double)nsteps;
  }
  // ---- write state back
  if (plane) A.theta0[(long long)cl * k + lane] = th0;
  if (lane == 0) {
    A.f0[cl] = f0;
    A.accept_count[cl] = nacc;   // (of THIS launch: launch_sweep adds the windows up)
    if (status == FMCMC_CHAIN_OK) { A.status[cl] = FMCMC_CHAIN_OK; A.status_step[cl] = 0; }
  }
}

size_t spec_logit_lds_bytes(bool adaptive) {
  return sizeof(double) * ((size_t)8 * PIPE_KMAX + 4 + 4 * 8 * PIPE_TRS + (size_t)(LG_LDS_DOUBLES + 2) + (adaptive ? 4 * SPEC_ADS : 0));
}
size_t spec_lds_bytes(int opt, bool adaptive) {
  return sizeof(double) * ((size_t)8 * PIPE_KMAX + 4 + 4 * 8 * PIPE_TRS + (size_t)opt * NT + (adaptive ? 4 * SPEC_ADS : 0));
}

size_t pipe_lds_bytes(int opt) { return sizeof(double) * ((size_t)8 * PIPE_KMAX + 4 * 8 * PIPE_TRS + (size_t)opt * NT); }

}  // namespace
