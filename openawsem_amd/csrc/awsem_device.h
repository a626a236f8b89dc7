// awsem_device.h -- device-side descriptors shared by the kernels (awsem_kernels.h) and the host API (awsem_hip.hip).
//
// One `DevSys<real>` describes a finalized AWSEM system: index arrays and parameter tables of every term, shared
// by all replicas and read-only during evaluation.  It is passed to the kernels BY VALUE: the whole descriptor
// block lives in the kernarg segment, so every scalar parameter and table pointer is a scalar (s_load) operand in
// the constant address space, hoisted out of the loops, and the tables behind the pointers are known-global
// (global_load, not flat).  Replica state (positions / velocities) lives in separate SoA arrays and, during a
// launch, in the LDS of the one workgroup that owns the replica.
#pragma once
#include <stdint.h>

#define AWSEM_NGROUPS 32
#define AWSEM_MAX_WELLS 4
// instances of each term kind a system may hold (the kernarg block is limited to 4 KB)
#define AWSEM_MAX_HARM 4
#define AWSEM_MAX_CHI 2
#define AWSEM_MAX_RAMA 4
#define AWSEM_MAX_EXCL 2
#define AWSEM_MAX_CONTACT 2
#define AWSEM_MAX_BETA 4
#define AWSEM_MAX_PAP 4
#define AWSEM_MAX_HELICAL 2
#define AWSEM_MAX_FM 2
#define AWSEM_MAX_MEMB 4
#define AWSEM_MAX_DH 2
#define AWSEM_MAX_GAUSS 16
#define AWSEM_FIXED_SCALE 4294967296.0   // 2^32: forces, energies and densities are accumulated as int64 fixed point
#define AWSEM_MAX_LISTS (AWSEM_MAX_EXCL + AWSEM_MAX_CONTACT + AWSEM_MAX_BETA + AWSEM_MAX_PAP)
#define AWSEM_MAX_STAGES 16   // temperature stages in one step launch (awsem_step_stages splits longer schedules)

// Every pair term (excl, contact, beta, pap) owns one Verlet pair list per replica: 32-bit entries in a per-replica
// global buffer (`DevSys::lists + replica * list_stride + list_off`), at most `list_cap` of them (= the number of
// candidate pairs, so a list cannot overflow), counted by LDS counter `list_id`.  A pair is listed when its screening
// distance is below cutoff + skin (`list_rc2` = (cutoff + skin)^2).
struct ListSlot { int list_off, list_cap, list_id; };

template <class real> struct HarmItem { int i, j; real r0, k; };
template <class real> struct HelicalItem { int o, n, h; real coef; };
// z_dependent_helical_term: coef blends towards coef + dcoef with the membrane switch alpha(z of CA_i)
template <class real> struct HelicalZItem { int ca; real dcoef; };
template <class real> struct DhItem { int i, j; real qq, pad; };
template <class real> struct GaussItem { int i, j; real r0, a, c; };   // a = 1/(2 sigma^2), c = k gamma (bond value g = -c exp(-a (r-r0)^2), or c r^2)
template <class real> struct MembItem { int atom; real scale; };
template <class real> struct alignas(16) MassiveItem { int atom; real inv_mass, sqrt_inv_mass; int id; };   // id: the caller's index of the atom (random streams are keyed by it)
// the virtual sites a massive particle parents (-1 padded) and its weight in each: the step kernel gathers their forces
#define AWSEM_MAX_GATHER 8
template <class real> struct alignas(16) GatherItem { int vs[AWSEM_MAX_GATHER]; real w[AWSEM_MAX_GATHER]; };
struct Int4 { int x, y, z, w; };
template <class real> struct alignas(2 * sizeof(real)) real2_t { real x, y; };
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
// Every Ramachandran-shaped term of one force group in ONE item list (replica-resident path): a term with two items (rama_proline_term
// of a protein with two prolines) otherwise costs a whole wave pass of ~360 instructions.  An item carries the index of its term's
// parameter set; the sets are staged in LDS ([set][well][8]: sigma, omega_phi, omega_psi, cos/sin phi0, cos/sin psi0, pad) and read
// per lane; wells a set does not have are padded with zero weights and contribute exact zeros.
struct RamaItem { int a[5]; int set; };
template <class real> struct RamaMergedD {
    int group, n, nwell, nset;            // n == 0: not merged (the kernels walk DevSys::rama term by term)
    const RamaItem* items;
    const real* bw;                       // n x nwell, as in RamaD (zero where the item's term has fewer wells)
    const real* params;                   // nset x nwell x 8
};
template <class real> struct ExclD {
    int group, nsites, nclass;
    const int* atom;                      // sites sorted by class
    const Int4* partners;                 // per site: up to 4 excluded partner ATOM indices (-1 padded)
    int class_start[9];                   // class c occupies sites [class_start[c], class_start[c+1])
    real k, r0;
    ListSlot ls; real list_rc2;           // entry = atom_a << 16 | atom_b
};
template <class real> struct ContactD {
    int group, nsites, env, min_seq_sep, flags, has_burial;
    const int* atom;
    const int* info;                      // per site: resid (bits 0-15) | chain (16-26) | residue type (27-31)
    const real* gamma;                    // [2][3][400]
    const real* bgamma;                   // [20][3] (pre-multiplied by nothing; k_burial applied in kernel)
    real k, rmin, rmax, rminII, rmaxII, eta2, eta_s2, rho0, cutoff2, kappa2;
    real rho_lo[3], rho_hi[3];
    real eta_sw2, z_m, zc, kb;
    ListSlot ls; real list_rc2;           // entry = site_i | site_j << 10 | (type_i*20+type_j) << 20 | in_density << 30 | has_pair_energy << 31
};
// Functional form of a lambda / gamma table (include/awsem_hip.h: awsem_add_hbond_beta_fn): lives in device memory, one per term.
template <class real> struct HbFn {
    int kind, pad;
    const int* rinfo;                     // per residue: type | type(i-1) << 5 | type(i+1) << 10 | chain edge << 15 | helix << 16 | chain << 17
    real scale;                           // -k
    real cc[5][3], g[3];
    real t1[400], t2[400], ob[20];
};
// Pair-list entry of the hydrogen-bond terms on the replica-resident path: everything the pair needs -- its atoms (15 bits each; the
// term's two_theta flag in bit 15 of w0) and the coefficient -k lambda / -k gamma -- so that a walk reads one record per pair instead of
// chasing list word -> acceptor / donor records -> table entry through three dependent loads, and terms of one shape and force group
// (beta_term_1/2/3; pap_term_1/2) share ONE list and one walk.
//   w0 = a1 | flag << 15 | d1 << 16,   w1 = a2 | a3 << 16,   w2 = d2 | d3 << 16
template <class real> struct HbEntry;
template <> struct HbEntry<float> { unsigned w0, w1, w2; float c; };
template <> struct HbEntry<double> { unsigned w0, w1, w2, pad; double c; };
template <class real> struct BetaD {
    static constexpr int res_slot = 3;    // acc/don records carry the residue index in .w
    __host__ __device__ const real* table() const { return lambda; }
    __host__ __device__ unsigned entry_flag() const { return (unsigned)two_theta; }
    int group, two_theta, nacc, ndon, nres;
    int scan_mode;                        // 1: the members of this term's list share one distance scan (build_hbond_group); 0: build_hbond per term
    const Int4* acc;                      // a1, a2, a3, residue
    const Int4* don;                      // d1, d2, d3, residue
    const real* lambda;                   // nres x nres, row = donor residue, pre-multiplied by -k (null when fn is set)
    const HbFn<real>* fn;                 // functional form of the same table, or null
    const unsigned* nzmask;               // bit (don_res * nres + acc_res) set where lambda != 0
    int nz_lds;                           // the mask fits the list builder's LDS scratch next to the acceptor / donor words
    real r_ON, r_OH, inv_sNO2, inv_sHO2, cutoff2;   // inv_s*2 = 1/sigma^2
    real screen2, max_expo;               // fp32 pre-screen: r_ON^2 bound and largest Gaussian exponent evaluated
    int band_lo, band_hi;                 // lambda[q][p] == 0 whenever band_lo < q - p < band_hi (zero band around the diagonal)
    ListSlot ls; real list_rc2;           // HbEntry records; terms of one force group with the same constants share the slot
    int list_owner;                       // this term walks the shared list (the first of its members)
};
template <class real> struct PapD {
    static constexpr int res_slot = 2;    // acc/don records carry the residue index in .z
    __host__ __device__ const real* table() const { return gamma; }
    __host__ __device__ unsigned entry_flag() const { return 0u; }
    int group, nacc, ndon, nres;
    int scan_mode;                        // as BetaD's
    const Int4* acc;                      // a1, a2, residue, -
    const Int4* don;                      // d1, d2, residue, -
    const real* gamma;                    // nres x nres, row = donor residue, pre-multiplied by -k (null when fn is set)
    const HbFn<real>* fn;                 // functional form of the same table, or null
    const unsigned* nzmask;               // bit (don_res * nres + acc_res) set where gamma != 0
    int nz_lds;
    real eta2, r0, d_i4, cutoff2;
    real min_arg;                         // fp32 pre-screen: switches with argument below this are treated as 0
    int band_lo, band_hi;                 // gamma[q][p] == 0 whenever band_lo < q - p < band_hi
    ListSlot ls; real list_rc2;           // HbEntry records, shared like BetaD's
    int list_owner;
};
template <class real> struct HelicalD { int group, n; const HelicalItem<real>* items; real r_ON, r_OH, inv_sNO2, inv_sHO2;
                                        const HelicalZItem<real>* zitems; real k_m2, z_m, center; };   // zitems: null unless z-dependent
// table2[b][i] = (t[b][i], t[b][i+1]) for i < ntab-1: the two interpolation nodes of a bin in one aligned load
// pairs[b] = atom_i | atom_j << 16 on the replica-resident path (atoms < 65536); pairs2 = (atom_i, atom_j) for the cell path
// Replica-resident path: the bonds are grouped into BUNDLES of up to AWSEM_FM_BUNDLE bonds that share one atom (the
// "owner": the lower-numbered end); one lane takes a bundle, keeps the owner's force in integer registers and adds it once.
// bundles[k] = (owner | count << 16, first bond); partners[u * nb_pad + k] = the other end of the bundle's bond u.
#define AWSEM_FM_BUNDLE 7
struct FmBundle { unsigned head, first; };
template <class real> struct FmD {
    int group, n, ntab, crop_len; const unsigned* pairs; const int* pairs2; const real2_t<real>* table2; real k, dr, inv_dr;
    const FmBundle* bundles; const unsigned* partners; int nbundles, nb_pad;
    // fp32 force walks of the bundled form read CROPPED rows (templateTerms.py:273-281 is the full row): per bond only the `crop_len` nodes
    // crop[b][0 .. crop_len) = t[b][lo_b .. lo_b + crop_len) around the bins whose slope can show in a Q19.12 force accumulator (outside
    // them k |slope| < 2^-15, which rounds to exactly zero there); lo_b rides in the high half of the bond's partner word.  One float per
    // node (the two nodes of a bin are one 8-byte load at 4-byte alignment).  Null in fp64 mode and without bundles.
    const float* crop;
};
template <class real> struct MembD { int group, n; const MembItem<real>* items; real k, k_m2, z_m, center; int kind, pad; };   // kind 0: slab; AWSEM_EXT_*: awsem_add_external
template <class real> struct DhD { int group, n; const DhItem<real>* items; real k, inv_lam; };
// A sum of bond values S = sum_b g_b(r_b) and an energy E = f(S):
//   kind 0: g = -c exp(-a (r - r0)^2)   kind 1: g = c r^2
//   cv 0: E = S (plain bond force)   1: E = k/2 (S - x0)^2   2: E = k sqrt(S)   3: E = k (sqrt(S) - x0)^2
// cv != 0 is OpenMM's CustomCVForce over one CustomBondForce; S is summed in LDS accumulator `slot` first.
template <class real> struct GaussD { int group, n; const GaussItem<real>* items; int kind, cv, slot, pad; real cv_k, cv_x0; };

template <class real> struct DevSys {
    int natoms;
    int nvs;
    const Int4* vs;                       // site, p1, p2, p3
    const real* vs_w;                     // 3 per site
    const real* inv_mass;                 // natoms (0 for massless)
    const real* sqrt_inv_mass;
    const MassiveItem<real>* massive;     // the nmassive particles the integrator moves
    const GatherItem<real>* gather;       // per massive particle (same order)
    int nmassive;
    int nfixed;                           // massless particles that are not virtual sites (they never move)
    const int* fixed;
    const int* orig;                      // internal atom index -> the caller's (null: identical); see Engine::finalize
    int n_harm, n_chi, n_rama, n_excl, n_contact, n_beta, n_pap, n_helical, n_fm, n_memb, n_dh, n_gauss;
    HarmD<real> harm[AWSEM_MAX_HARM]; ChiD<real> chi[AWSEM_MAX_CHI]; RamaD<real> rama[AWSEM_MAX_RAMA];
    RamaMergedD<real> ramam;
    ExclD<real> excl[AWSEM_MAX_EXCL]; ContactD<real> contact[AWSEM_MAX_CONTACT]; BetaD<real> beta[AWSEM_MAX_BETA];
    PapD<real> pap[AWSEM_MAX_PAP]; HelicalD<real> helical[AWSEM_MAX_HELICAL]; FmD<real> fm[AWSEM_MAX_FM];
    MembD<real> memb[AWSEM_MAX_MEMB]; DhD<real> dh[AWSEM_MAX_DH];
    const GaussD<real>* gauss;            // n_gauss descriptors in device memory (read with scalar loads): bond sums come in any number,
                                          // and run-time indexing of a large array inside the by-value block sends it to scratch
    int fuse_vs;                          // the first force phase (density, fragment memory, collective-variable sums) touches massive particles only:
                                          // the step kernel places the virtual sites in that phase instead of a phase of their own
    int n_cv;                             // Gaussian-bond terms with cv != 0 (each owns one LDS accumulator)
    int max_contact_sites;                // LDS scratch sizing
    int any_zdep;
    unsigned* lists;                      // [n_replicas][list_stride] pair-list entries
    long long list_stride;
    int n_lists;
    real skin_half2;                      // (skin/2)^2: lists are rebuilt once an atom has moved further than skin/2
    int remove_cm;                        // CMMotionRemover: subtract the mass-weighted mean velocity at the top of every step
    real inv_total_mass;                  // 1 / sum of the masses
    real* list_ref;                       // [R][nmassive][3]: where the massive particles stood when the replica's lists were last built, and
    unsigned* list_counts;                // [R][AWSEM_MAX_LISTS]: the lists' lengths -- what a step launch needs to go on with its predecessor's lists
    const BetaD<real>* beta_dev;          // copies of beta[] / pap[] in device memory (static fields only): build_hbond_group reads its members'
    const PapD<real>* pap_dev;            // descriptors from there, in a loop, instead of unrolling over the kernarg block
    int prio;                             // step kernel: how the replicas sharing a CU take turns at issue priority (0: hardware default, oldest wave first;
                                          // 1: rotating with the step count; 2: rotating with the clock; 3 | gain*16 << 8: rotation shifted by a
                                          // per-replica bias learnt from launch to launch, see pace_bias)
    double* vel_scale;                    // step kernel: [R] factors the velocities are multiplied by as they are loaded, then set back to 1 (the rescale of a
                                          // replica-exchange sweep, taken in by the next launch); nullptr: none pending
    int pace_group;                       // number of CUs: workgroups b, b + pace_group, ... of a one-wave grid share a CU
    int* status;                          // device error word: bit 0 = a pair list overflowed
    long long* prof;                      // 16 phase-cycle accumulators (written by -DAWSEM_PROF builds only)
};
// Not a hardware limit (the kernarg segment is sized per kernel: the fp64 cell-list kernels take DevSys + CellSys = 4.4 KB); a guard against growing
// the block unnoticed -- past some size the compiler may copy a by-value block to scratch (rejected (vii)), which build() checks.
static_assert(sizeof(DevSys<double>) <= 4096, "DevSys grew: check the step kernel's scratch size in the build remarks");
