// awsem_device.h -- device-side descriptors shared by the kernels (awsem_kernels.h) and the host API (awsem_hip.hip).
//
// One `DevSys<real>` describes a finalized AWSEM system: index arrays and parameter tables of every term, shared
// by all replicas and read-only during evaluation.  Replica state (positions / velocities) lives in separate
// SoA arrays and, during a launch, in the LDS of the one workgroup that owns the replica.
#pragma once
#include <stdint.h>

#define AWSEM_NGROUPS 32
#define AWSEM_MAX_WELLS 4
#define AWSEM_FIXED_SCALE 4294967296.0   // 2^32: forces, energies and densities are accumulated as int64 fixed point

template <class real> struct HarmItem { int i, j; real r0, k; };
template <class real> struct HelicalItem { int o, n, h; real coef; };
template <class real> struct DhItem { int i, j; real qq, pad; };
template <class real> struct MembItem { int atom; real scale; };
struct Int4 { int x, y, z, w; };
struct Int5 { int a[5]; };

template <class real> struct HarmD { int group, n; const HarmItem<real>* items; };
template <class real> struct ChiD { int group, n; const Int4* atoms; real k, chi0; };
template <class real> struct RamaD {
    int group, n, nwell;
    const Int5* atoms;
    const real* bw;                       // n x nwell: k * w_well * bond weight (folded on the host)
    real sigma[AWSEM_MAX_WELLS], omega_phi[AWSEM_MAX_WELLS], omega_psi[AWSEM_MAX_WELLS];
    real cphi0[AWSEM_MAX_WELLS], sphi0[AWSEM_MAX_WELLS], cpsi0[AWSEM_MAX_WELLS], spsi0[AWSEM_MAX_WELLS];
};
template <class real> struct ExclD {
    int group, nsites, nclass;
    const int* atom;                      // sites sorted by class
    const Int4* partners;                 // per site: up to 4 excluded partner ATOM indices (-1 padded)
    int class_start[9];                   // class c occupies sites [class_start[c], class_start[c+1])
    real k, r0;
};
template <class real> struct ContactD {
    int group, nsites, env, min_seq_sep, flags, has_burial;
    const int* atom; const int* restype; const int* resid; const int* chain;
    const real* gamma;                    // [2][3][400]
    const real* bgamma;                   // [20][3] (pre-multiplied by nothing; k_burial applied in kernel)
    real k, rmin, rmax, rminII, rmaxII, eta2, eta_s2, rho0, cutoff2, kappa2;
    real rho_lo[3], rho_hi[3];
    real eta_sw2, z_m, zc, kb;
};
template <class real> struct BetaD {
    int group, two_theta, nacc, ndon, nres;
    const Int4* acc;                      // a1, a2, a3, residue
    const Int4* don;                      // d1, d2, d3, residue
    const real* lambda;                   // nres x nres, row = donor residue, pre-multiplied by -k
    real r_ON, r_OH, inv_sNO2, inv_sHO2, cutoff2;   // inv_s*2 = 1/sigma^2
};
template <class real> struct PapD {
    int group, nacc, ndon, nres;
    const Int4* acc;                      // a1, a2, residue, -
    const Int4* don;                      // d1, d2, residue, -
    const real* gamma;                    // nres x nres, row = donor residue, pre-multiplied by -k
    real eta2, r0, d_i4, cutoff2;
};
template <class real> struct HelicalD { int group, n; const HelicalItem<real>* items; real r_ON, r_OH, inv_sNO2, inv_sHO2; };
template <class real> struct FmD { int group, n, ntab; const int* pairs; const real* table; real k, dr, inv_dr; };
template <class real> struct MembD { int group, n; const MembItem<real>* items; real k, k_m2, z_m, center; };
template <class real> struct DhD { int group, n; const DhItem<real>* items; real k, inv_lam; };

template <class real> struct DevSys {
    int natoms;
    int nvs;
    const Int4* vs;                       // site, p1, p2, p3
    const real* vs_w;                     // 3 per site
    const real* inv_mass;                 // natoms (0 for massless)
    const real* sqrt_inv_mass;
    int n_harm, n_chi, n_rama, n_excl, n_contact, n_beta, n_pap, n_helical, n_fm, n_memb, n_dh;
    const HarmD<real>* harm; const ChiD<real>* chi; const RamaD<real>* rama; const ExclD<real>* excl;
    const ContactD<real>* contact; const BetaD<real>* beta; const PapD<real>* pap; const HelicalD<real>* helical;
    const FmD<real>* fm; const MembD<real>* memb; const DhD<real>* dh;
    int max_contact_sites;                // LDS scratch sizing
    int any_zdep;
};
