/*
 * awsem_hip.h -- C ABI of libawsem_hip.so, the MI355X (gfx950) AWSEM force + Langevin engine.
 *
 * This is the drop-in boundary for the openawsem hot path.  The reference (epochbio/openawsem) has no
 * native/FFI interface of its own: every term factory builds an OpenMM Force object and OpenMM's Context
 * evaluates it.  Each entry point below therefore cites the reference *call site* whose OpenMM object it
 * replaces (paths relative to the reference checkout).  All arguments are plain pointers and sizes; the caller
 * owns every host buffer; the library owns all device memory behind the opaque handle.
 *
 * Conventions
 *   - every function returns 0 on success, a negative AWSEM_E* code on failure; awsem_last_error() returns a
 *     thread-local, human readable message for the last failure on the calling thread.
 *   - units: nm, ps, amu, kJ/mol, K (OpenMM's unit system).
 *   - atom indices are 0-based into the natoms particles of the system; a handle carries `n_replicas`
 *     independent copies of the coordinates/velocities (one trajectory each) sharing one set of terms.
 *   - a handle is bound to one HIP device and is not thread-safe.
 *   - `group` is the OpenMM force group (0..31) the factory assigned (Force.setForceGroup); energy/force
 *     queries take a 32-bit mask of groups, as Context.getState(groups=...) does.
 *   - precision: AWSEM_FP32 runs the production kernels (fp32 arithmetic, 64-bit fixed-point force
 *     accumulation); AWSEM_FP64 runs the same kernels instantiated in double (the "check mode").
 */
#ifndef AWSEM_HIP_H
#define AWSEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct awsem_context* awsem_handle;

enum { AWSEM_FP32 = 0, AWSEM_FP64 = 1 };

enum {
    AWSEM_OK = 0,
    AWSEM_EINVAL = -1,   /* bad argument (index out of range, null pointer, wrong size ...) */
    AWSEM_ESTATE = -2,   /* call made in the wrong phase (e.g. add_* after finalize) */
    AWSEM_EDEVICE = -3,  /* HIP runtime error or no usable gfx950 device */
    AWSEM_ELIMIT = -4,   /* system too large for the replica-resident (LDS) path */
    AWSEM_ENOMEM = -5
};

const char* awsem_last_error(void);
/* version string "awsem_hip <major>.<minor> gfx950" */
const char* awsem_version(void);

/* ---- system construction --------------------------------------------------------------------------------
 * Replaces openmm.System creation in OpenMMAWSEMSystem.__init__ (openawsem/openAWSEM.py:820-914):
 * ForceField(awsem.xml).createSystem(topology) -> per-particle masses (topology/awsem.xml:4-9; mass 0 marks
 * the virtual N/H/C sites). */
int awsem_create(int natoms, int n_replicas, int device, int precision, awsem_handle* out);
void awsem_destroy(awsem_handle h);
int awsem_set_masses(awsem_handle h, const double* mass_amu, int natoms);

/* CMMotionRemover(frequency 1): ForceField.createSystem(topology) (openAWSEM.py:826,878) is called with its default
 * removeCMMotion=True, so every OpenMM System the reference builds carries one.  At the start of every awsem_step step the
 * mass-weighted mean velocity of the massive particles is subtracted from each of them (OpenMM's updateContextState order).
 * Off by default at this level, as in a bare openmm.System; the Python mirror of OpenMMAWSEMSystem switches it on. */
int awsem_set_cm_motion_removal(awsem_handle h, int enabled);

/* ThreeParticleAverageSite(p1,p2,p3,w1,w2,w3) per virtual atom -- setup_virtual_sites, openAWSEM.py:666-696.
 * parents[3*i..] and weights[3*i..] belong to site[i]. */
int awsem_set_virtual_sites(awsem_handle h, int n, const int32_t* site, const int32_t* parents,
                            const double* weights);

/* ---- terms (one call per reference factory product) ----------------------------------------------------- */

/* HarmonicBondForce: E = 1/2 k (r-r0)^2.  con_term basicTerms.py:24-58, chain_term basicTerms.py:61-95. */
int awsem_add_harmonic_bonds(awsem_handle h, int group, int n, const int32_t* pairs, const double* r0,
                             const double* k);

/* chi_term basicTerms.py:98-138: E = k_chi (chi - chi0)^2 over atoms (CA, C, N, CB). */
int awsem_add_chi(awsem_handle h, int group, int n, const int32_t* atoms4, double k_chi, double chi0);

/* rama_term / rama_proline_term / rama_ssweight_term basicTerms.py:266-440.
 * atoms5 = (C_{i-1}, N_i, CA_i, C_i, N_{i+1}); well_params = nwell rows of
 * (w, sigma, omega_phi, phi0, omega_psi, psi0); bond_well_weight is n*nwell (row per bond; the ssweight(well,res)
 * factor) or NULL for all ones.  E = -k sum_w weight*w*exp(-sigma(omega_phi(cos(phi-phi0)-1)^2+omega_psi(cos(psi-psi0)-1)^2)) */
int awsem_add_rama(awsem_handle h, int group, int n, const int32_t* atoms5, int nwell,
                   const double* well_params, const double* bond_well_weight, double k);

/* excl_term basicTerms.py:141-191 (CustomNonbondedForce, CutoffNonPeriodic, interaction groups).
 * Sites of equal class interact (class 0 = CA and CB, class 1 = O); excl_pairs are atom-index pairs that
 * createExclusionsFromBonds(oa.bonds, 1) removes.  E = k (r - r_excl)^2 for r < r_excl. */
int awsem_add_excl(awsem_handle h, int group, int nsites, const int32_t* atom, const int32_t* cls,
                   int nexcl, const int32_t* excl_pairs, double k, double r_excl);

/* contact_term contactTerms.py:99-375 (CustomGBForce: density rho, direct + mediated pair energy, burial)
 * and burial_term contactTerms.py:1772-1840 (flags & AWSEM_CONTACT_BURIAL_ONLY).
 * Sites are the nres "CB, or CA for glycine" atoms.  gamma is [env 0..1][kind 0=direct,1=protein,2=water][20][20],
 * burial_gamma is [20][3] ([2][20][3] with AWSEM_CONTACT_MEMBRANE_BURIAL) or NULL (burialPartOn=False).
 * dparams: see awsem_contact_dparam below. */
enum {
    AWSEM_CONTACT_ZDEP = 1,         /* z_dependent=True: blend env 0 / env 1 with alphaMembrane(z) */
    AWSEM_CONTACT_BURIAL_ONLY = 2,  /* standalone burial_term: no pair energy */
    AWSEM_CONTACT_NOCUTOFF = 4,     /* burial_term never calls setNonbondedMethod -> NoCutoff */
    AWSEM_CONTACT_INDEX_BASED = 16, /* index_based_contact_term contactTerms.py:668-876: gamma is [3][nsites][nsites] and burial_gamma
                                       [nsites][3], both by site (= residue) index; the pair (i < j) reads gamma[kind][i][j];
                                       restype is ignored; with AWSEM_CONTACT_ZDEP (:812-831) gamma is [2][3][nsites][nsites] -- the water
                                       tables, then the membrane tables -- blended per pair with alpha_i alpha_j; not combinable with
                                       BURIAL_ONLY / MEMBRANE_BURIAL */
    AWSEM_CONTACT_MEMBRANE_BURIAL = 8  /* hybrid_contact_term contactTerms.py:1283-1289: burial_gamma is [water, membrane][20][3] and
                                          the wells of site i are blended with (1 - alpha_i, alpha_i); needs AWSEM_CONTACT_ZDEP */
};
enum awsem_contact_dparam {
    AWSEM_CP_K = 0, AWSEM_CP_RMIN, AWSEM_CP_RMAX, AWSEM_CP_RMINII, AWSEM_CP_RMAXII, AWSEM_CP_ETA,
    AWSEM_CP_ETA_SIGMA, AWSEM_CP_RHO0, AWSEM_CP_CUTOFF, AWSEM_CP_BURIAL_KAPPA,
    AWSEM_CP_BURIAL_RHO_MIN0, AWSEM_CP_BURIAL_RHO_MIN1, AWSEM_CP_BURIAL_RHO_MIN2,
    AWSEM_CP_BURIAL_RHO_MAX0, AWSEM_CP_BURIAL_RHO_MAX1, AWSEM_CP_BURIAL_RHO_MAX2,
    AWSEM_CP_ETA_SWITCH, AWSEM_CP_Z_M, AWSEM_CP_MEMBRANE_CENTER, AWSEM_CP_K_BURIAL,
    AWSEM_CP_COUNT
};
int awsem_add_contact(awsem_handle h, int group, int nsites, const int32_t* atom, const int32_t* restype,
                      const int32_t* resid, const int32_t* chain, int env, int min_seq_sep, int flags,
                      const double* dparams, const double* gamma, const double* burial_gamma);

/* beta_term_1/2/3 hydrogenBondTerms.py:418-611 (CustomHbondForce, CutoffNonPeriodic on distance(a1,d1)).
 * acc_atoms = 3 per acceptor (a1,a2,a3), don_atoms = 3 per donor (d1,d2,d3); -1 marks unused slots.
 *   theta(x;y,z) = exp(-(|x-y|-r_ON)^2/(2 s_NO^2) - (|x-z|-r_OH)^2/(2 s_HO^2))
 *   two_theta==0:  E = -k lambda[don_res][acc_res] theta(a1;d1,d2)                        (beta_1)
 *   two_theta==1:  E = -k lambda[don_res][acc_res] theta(a1;d1,d2) theta(d3;a2,a3)        (beta_2, beta_3)
 * lambda is the dense nres x nres table the factory builds (row = donor residue). */
int awsem_add_hbond_beta(awsem_handle h, int group, int two_theta, int nacc, const int32_t* acc_atoms,
                         const int32_t* acc_res, int ndon, const int32_t* don_atoms, const int32_t* don_res,
                         int nres, const double* lambda, double k, double r_ON, double sigma_NO, double r_OH,
                         double sigma_HO, double cutoff);

/* pap_term_1/2 hydrogenBondTerms.py:614-761.  acc_atoms = (a1,a2) = (CA_i, CA_{i+4}); don_atoms = (d1,d2).
 *   E = -k gamma[don_res][acc_res] 1/2(1+tanh(eta(r0-|a1-d1|))) 1/2(1+tanh(eta(r0-|a2-d2|))) 1/2(1+tanh(eta(|a1-a2|-d_i4))) */
int awsem_add_hbond_pap(awsem_handle h, int group, int nacc, const int32_t* acc_atoms, const int32_t* acc_res,
                        int ndon, const int32_t* don_atoms, const int32_t* don_res, int nres,
                        const double* gamma, double k, double eta, double r0, double d_i4, double cutoff);

/* The same two terms with the lambda / gamma table given as a FUNCTION of the residue pair instead of a dense nres x nres
 * array (which is the reference's own limit on system size: hydrogenBondTerms.py:441,460,513-519 build five such tables,
 * 3.2 GB each in fp64 at 20 000 residues).  The kernels evaluate the entry of a pair when they need it:
 *   kind AWSEM_HB_LAMBDA_1   lambda_1  get_lambda_by_index(i, j, 0)                       hydrogenBondTerms.py:160-184, 441-447
 *        AWSEM_HB_LAMBDA_2   lambda_2  get_Lambda_2, zero on chain edges (n = 1)          :300-328, 513-519
 *        AWSEM_HB_LAMBDA_3   lambda_3  get_Lambda_3, zero on chain edges (n = 1)          :331-359, 577-583
 *        AWSEM_HB_GAMMA_PAP_1  get_pap_gamma_APH + get_pap_gamma_AP                       :216-267, 649-658
 *        AWSEM_HB_GAMMA_PAP_2  get_pap_gamma_P                                            :270-297, 728-736
 * with i the donor residue, j the acceptor residue, c the class of |i - j| (4..17, 18..44, >= 45; no entry below 4):
 *   LAMBDA_1: C[0][c]
 *   LAMBDA_2: C[0][c] + C[1][c] T1[a_i][a_j] + C[2][c] (T2[a_i+1][a_j-1] + T2[a_i-1][a_j+1]) + C[3][c] (O[a_i] + O[a_j])
 *   LAMBDA_3: C[0][c] + C[1][c] T1[a_i+1][a_j] + C[2][c] O[a_i+1] + C[1][c] O[a_j]
 *   PAP_1:    (13 <= i-j <= 16 and same chain ? G[0] : 0) + (i-j >= 17 or other chain ? s G[1] : 0),  s = 1.5 if both helical else 1
 *   PAP_2:    (i-j >= 9 or other chain ? s G[2] : 0)
 * res_info is nres x 4 int32: residue type a (0..19, the H-bond ordering se_map_1_letter :15-18), chain index, chain-edge flag
 * (isChainEdge(i, n=1)), helix flag (ssweight[i][1] == 1).  coef is 838 doubles: C[5][3], G[3], T1[20][20], T2[20][20], O[20]
 * (the factory folds the signs and the 0.5 / 0.25 factors of get_Lambda_2/3 into C). */
enum { AWSEM_HB_LAMBDA_1 = 1, AWSEM_HB_LAMBDA_2 = 2, AWSEM_HB_LAMBDA_3 = 3, AWSEM_HB_GAMMA_PAP_1 = 4, AWSEM_HB_GAMMA_PAP_2 = 5 };
#define AWSEM_HB_FN_COEFS 838
int awsem_add_hbond_beta_fn(awsem_handle h, int group, int two_theta, int nacc, const int32_t* acc_atoms,
                            const int32_t* acc_res, int ndon, const int32_t* don_atoms, const int32_t* don_res, int nres,
                            int kind, const int32_t* res_info, const double* coef, double k, double r_ON, double sigma_NO,
                            double r_OH, double sigma_HO, double cutoff);
int awsem_add_hbond_pap_fn(awsem_handle h, int group, int nacc, const int32_t* acc_atoms, const int32_t* acc_res,
                           int ndon, const int32_t* don_atoms, const int32_t* don_res, int nres, int kind,
                           const int32_t* res_info, const double* coef, double k, double eta, double r0, double d_i4,
                           double cutoff);

/* helical_term hydrogenBondTerms.py:790-832: E = -coef theta(O_i; N_{i+4}, H_{i+4}), coef = k(fa_i + fa_{i+4}). */
int awsem_add_helical(awsem_handle h, int group, int n, const int32_t* atoms3, const double* coef, double r_ON,
                      double sigma_NO, double r_OH, double sigma_HO);
/* z_dependent_helical_term hydrogenBondTerms.py:835-887 (CustomCompoundBondForce(4): O_i, N_{i+4}, H_{i+4}, CA_i):
 * E = -theta ((1 - alpha) coef + alpha coef_membrane), alpha = 1/2 tanh(eta (z + z_m)) + 1/2 tanh(eta (z_m - z)),
 * z = z(CA_i) - membrane_center; coef = k(fa_i + fa_{i+4}) with the water propensities, coef_membrane with the membrane ones.
 * atoms4: n x (O, N, H, CA). */
int awsem_add_helical_z(awsem_handle h, int group, int n, const int32_t* atoms4, const double* coef, const double* coef_membrane,
                        double r_ON, double sigma_NO, double r_OH, double sigma_HO, double eta_switching, double z_m,
                        double membrane_center);

/* fragment_memory_term templateTerms.py:122-294 (CustomCompoundBondForce(2) + Discrete2DFunction):
 * E = -k_fm lerp(table[b][i], table[b][i+1]) with i = min(ntab-2, floor(r/dr)); table is nbonds x ntab, row major. */
int awsem_add_fragment_memory(awsem_handle h, int group, int nbonds, const int32_t* pairs, int ntab,
                              const double* table, double k_fm, double dr);

/* membrane_term membraneTerms.py:12-54 and membrane_preassigned_term membraneTerms.py:115-156
 * (CustomExternalForce): E = k scale (1/2 tanh(k_m((z-c)+z_m)) + 1/2 tanh(k_m(z_m-(z-c)))). */
int awsem_add_membrane(awsem_handle h, int group, int n, const int32_t* atoms, const double* scale, double k,
                       double k_m, double z_m, double center);

/* Other CustomExternalForce terms of the reference, one parameter per particle:
 *   AWSEM_EXT_LINEAR_X/Y/Z  pulling_term biasTerms.py:574-605:                      E = -k param_i x   (or y, z)
 *   AWSEM_EXT_ABS_Z         membrane_preassigned_side_term membraneTerms.py:177-215: E = k |z - center - param_i| */
enum { AWSEM_EXT_LINEAR_X = 1, AWSEM_EXT_LINEAR_Y = 2, AWSEM_EXT_LINEAR_Z = 3, AWSEM_EXT_ABS_Z = 4 };
int awsem_add_external(awsem_handle h, int group, int kind, int n, const int32_t* atoms, const double* param, double k,
                       double center);

/* debye_huckel_term debyeHuckelTerms.py:10-76 (CustomBondForce): E = k_dh qq / r exp(-r/screening_length). */
int awsem_add_debye_huckel(awsem_handle h, int group, int n, const int32_t* pairs, const double* qq,
                           double k_dh, double screening_length);

/* er_term templateTerms.py:643-693 and additive_amhgo_term templateTerms.py:600-640 (CustomBondForce with per-bond
 * parameters): E = -kgamma exp(-(r - r0)^2 / (2 sigma^2)), kgamma = k_er gamma_ij (resp. k_amhgo gamma_ij). */
int awsem_add_gaussian_bonds(awsem_handle h, int group, int n, const int32_t* pairs, const double* r0,
                             const double* sigma, const double* kgamma);

/* A sum over bonds S = sum_b g_b(r_b) and an energy E = f(S): OpenMM's CustomBondForce, alone or as the one
 * collective variable of a CustomCVForce.
 *   bond_kind AWSEM_BOND_GAUSS: g = coef exp(-(r - r0)^2 / (2 sigma^2))     AWSEM_BOND_R2: g = coef r^2 (r0, sigma unused)
 *   cv_kind   AWSEM_CV_SUM:          E = S                 q_value biasTerms.py:146-188, qc_value :191-223, partial_q_value :226-267
 *             AWSEM_CV_HARMONIC:     E = cv_k/2 (S-x0)^2   qbias_term biasTerms.py:270-302 ("0.5*k_qbias*(q-q0)^2")
 *             AWSEM_CV_SQRT:         E = cv_k sqrt(S)      rg_term biasTerms.py:461-486 ("unit*rg_square^0.5")
 *             AWSEM_CV_SQRT_SQUARE:  E = cv_k (sqrt(S)-x0)^2   rg_bias_term biasTerms.py:489-528
 * Available on both paths. */
#define AWSEM_BOND_GAUSS 0
#define AWSEM_BOND_R2 1
#define AWSEM_BOND_R2_XY 2   /* g = coef ((x1-x2)^2 + (y1-y2)^2): cylindrical_rg_bias_term biasTerms.py:531-571 */
#define AWSEM_CV_SUM 0
#define AWSEM_CV_HARMONIC 1
#define AWSEM_CV_SQRT 2
#define AWSEM_CV_SQRT_SQUARE 3
int awsem_add_bond_sum(awsem_handle h, int group, int bond_kind, int n, const int32_t* pairs, const double* r0,
                       const double* sigma, const double* coef, int cv_kind, double cv_k, double cv_x0);

/* Uploads tables and index arrays to the device; no add_* / set_masses / set_virtual_sites afterwards.
 * Replaces Simulation(topology, system, integrator, platform) -> Context creation (scripts/mm_run.py:148). */
int awsem_finalize(awsem_handle h);

/* ---- state ---------------------------------------------------------------------------------------------
 * context.setPositions scripts/mm_run.py:150, scripts/mm_analyze.py:163.  xyz is natoms x 3 (AoS, nm);
 * replica == -1 writes the same coordinates to every replica.  Stored coordinates of virtual sites are used
 * as given (OpenMM does not re-derive them on setPositions); awsem_compute_virtual_sites re-derives them. */
int awsem_set_positions(awsem_handle h, int replica, const double* xyz, int natoms);
int awsem_get_positions(awsem_handle h, int replica, double* xyz, int natoms);
int awsem_set_velocities(awsem_handle h, int replica, const double* vxyz, int natoms);
int awsem_get_velocities(awsem_handle h, int replica, double* vxyz, int natoms);
/* Whole ensemble at once (what a reporter or a checkpoint of an R-replica run needs: one transfer instead of R):
 * xyz / vxyz are n_replicas x natoms x 3; either may be NULL.  CheckpointReporter / loadCheckpoint, scripts/mm_run.py:92-94,124-127. */
int awsem_set_state(awsem_handle h, const double* xyz, const double* vxyz);
int awsem_get_state(awsem_handle h, double* xyz, double* vxyz);
/* Kinetic energy of every replica, sum over the massive particles of 1/2 m v^2 (kJ/mol): the temperature column of
 * StateDataReporter (scripts/mm_run.py:157-158), reduced on the device. */
int awsem_kinetic_energy(awsem_handle h, double* ke_per_replica);
/* Asynchronous snapshot for the reporters of a run (StateDataReporter, DCDReporter, CheckpointReporter: scripts/mm_run.py:157-164, which
 * OpenMM serves synchronously inside simulation.step).  awsem_snapshot_begin queues, BEHIND everything the handle's stream already holds:
 * the potential energy of the masked groups (of every replica when energy_replica is -1, else of that replica alone -- what
 * StateDataReporter prints; the other replicas' entries are then NaN) and the kinetic energy of every replica, a copy of the positions
 * of `replica` (-1: none) and, if with_state != 0, of the whole ensemble state -- all into pinned host memory the handle owns (slot 0 or 1) -- and returns
 * without waiting.  awsem_snapshot_end waits for that slot only and hands the values out, converted exactly as awsem_energy,
 * awsem_kinetic_energy, awsem_get_positions and awsem_get_state convert them (any output may be NULL; xyz_all / vxyz_all need
 * with_state).  A caller that queues the next awsem_step between the two calls writes its files while the GPU computes.  A slot holds
 * one snapshot at a time: begin on a slot that has not been collected is AWSEM_ESTATE. */
int awsem_snapshot_begin(awsem_handle h, int slot, uint32_t group_mask, int energy_replica, int replica, int with_state);
int awsem_snapshot_end(awsem_handle h, int slot, double* e_per_replica, double* ke_per_replica, double* xyz_replica,
                       double* xyz_all, double* vxyz_all);
int awsem_compute_virtual_sites(awsem_handle h);
/* context.setVelocitiesToTemperature scripts/mm_run.py:151; T is per replica (n_replicas). */
int awsem_set_velocities_to_temperature(awsem_handle h, const double* T_kelvin, uint64_t seed);

/* Multiply the velocities of replica r by scale[r] (n_replicas factors).  Used after a replica-exchange swap of
 * temperature labels, v *= sqrt(T_new/T_old); the reference delegates REMD to openmmtools' ReplicaExchangeSampler
 * (scripts/mm_remd.py:133-149), which exchanges thermodynamic states between replicas in the same way. */
int awsem_scale_velocities(awsem_handle h, const double* scale);

/* ---- evaluation ----------------------------------------------------------------------------------------
 * context.getState(getEnergy=True, groups=...) scripts/mm_analyze.py:165-178, tests/test_energies.py:119.
 * awsem_energy: one potential energy per replica for the masked groups.
 * awsem_energy_by_group: n_replicas x 32 energies in one pass.
 * awsem_forces: n_replicas x natoms x 3; redistribute != 0 folds virtual-site forces onto their parents
 * (what getState(getForces=True) reports), 0 returns the raw per-atom gradient. */
int awsem_energy(awsem_handle h, uint32_t group_mask, double* e_per_replica);
int awsem_energy_by_group(awsem_handle h, double* e_replica_by_group);
int awsem_forces(awsem_handle h, uint32_t group_mask, int redistribute, double* f);

/* Pair set of one pair term for one replica, from the engine's own list builder (the step kernel evaluates the
 * pair terms from such lists, rebuilt whenever an atom has moved more than half the skin).  No counterpart in the
 * reference: OpenMM keeps its neighbour lists private; this exists for the bit-exact list check against a
 * brute-force O(N^2) scan.  kind/index select the term (index = order of the add_* calls of that kind).  A pair is
 * listed when its screening distance r satisfies r < cutoff + skin, with r^2 = fma(dz,dz,fma(dy,dy,dx*dx))
 * evaluated in the handle's precision from the stored coordinates:
 *   AWSEM_LIST_EXCL     same-class site pairs minus the bonded exclusions, cutoff = r_excl;   pairs = atom indices (a < b)
 *   AWSEM_LIST_CONTACT  site pairs that enter the density (|resid_i - resid_j| >= 2) or the pair energy (different chains
 *                       or |resid_i - resid_j| >= min_seq_sep), cutoff = the CustomGB cutoff; pairs = site indices (i < j)
 *   AWSEM_LIST_BETA     (acceptor, donor) index pairs with lambda != 0, r = |a1 - d1|, cutoff = the reference's
 *                       cutoff (fp64) or the fp32 pre-screen r_ON + sqrt(60) sigma_NO, whichever is smaller
 *   AWSEM_LIST_PAP      (acceptor, donor) index pairs with gamma != 0, r = |a1 - d1|, cutoff = the reference's cutoff
 * Pairs are written sorted; *n_pairs is the total (call again with a larger buffer if it exceeds capacity). */
enum { AWSEM_LIST_EXCL = 0, AWSEM_LIST_CONTACT = 1, AWSEM_LIST_BETA = 2, AWSEM_LIST_PAP = 3 };
int awsem_neighbors(awsem_handle h, int replica, int kind, int index, double skin, int32_t* pairs, int capacity,
                    int* n_pairs);

/* simulation.step(n) with LangevinIntegrator(T, friction, dt) scripts/mm_run.py:145,171-175.
 * T_kelvin is per replica (integrator.setTemperature); all terms (every group) drive the dynamics. */
int awsem_step(awsem_handle h, int nsteps, const double* T_kelvin, double dt_ps, double friction_per_ps,
               uint64_t seed);
/* The annealing loop of scripts/mm_run.py:171-175 -- `for i in range(n): integrator.setTemperature(T_i); simulation.step(k)` -- as ONE
 * call: n_stages stages of steps_per_stage steps, T_kelvin[stage][replica].  The trajectory is bit for bit the one the same sequence of
 * awsem_step calls gives (the random numbers are a function of the step counter, the noise amplitude is set at each stage's first step);
 * what goes is the launch boundary between the stages: state write-back and reload, the staging of the tables, and the wait for the
 * slowest replica (1r69, 1024 replicas, stages of 50 steps: +4 % at four stages a launch).  At most 16 stages ride in one launch, longer
 * schedules are split; the cell-list path runs the stages as awsem_step calls. */
int awsem_step_stages(awsem_handle h, int n_stages, int steps_per_stage, const double* T_kelvin, double dt_ps,
                      double friction_per_ps, uint64_t seed);
int64_t awsem_step_count(awsem_handle h);
int awsem_set_step_count(awsem_handle h, int64_t step);

/* simulation.minimizeEnergy(tolerance, maxIterations) scripts/mm_run.py:136,153 (OpenMM LocalEnergyMinimizer).
 * Every replica is relaxed on the device until the RMS force over its massive degrees of freedom is <= tolerance
 * (kJ/mol/nm) or max_iterations force evaluations are spent (0 = no limit); virtual sites follow their parents,
 * velocities are left untouched.  The algorithm is FIRE, not OpenMM's L-BFGS: the reference pins only the contract
 * (a nearby local minimum within the force tolerance).  iterations / rms_force: n_replicas values each, or NULL.
 * Both paths: one workgroup per replica on the replica-resident path, the same iteration as grid-wide kernels on the cell-list path. */
int awsem_minimize(awsem_handle h, double tolerance_kj_mol_nm, int max_iterations, int32_t* iterations,
                   double* rms_force);

/* awsem_step returns as soon as the launch is enqueued (up to 32 launches may be in flight; the temperatures are
 * copied before it returns).  awsem_synchronize waits for them; every other call on the handle synchronises first.
 * awsem_last_kernel_ms: duration of the last awsem_energy* / awsem_forces / awsem_minimize operation, or the MEAN
 * duration of the awsem_step launches since the previous synchronisation, measured with HIP events on the handle's
 * stream (ms); used by bench.py for the roofline line. */
int awsem_last_kernel_ms(awsem_handle h, double* ms);
int awsem_synchronize(awsem_handle h);

/* ---- temperature replica exchange without host hops ------------------------------------------------------------------
 * The reference hands replica exchange to openmmtools' ReplicaExchangeSampler in ONE process (scripts/mm_remd.py:85-149: a
 * temperature ladder, `exchange_frequency` Langevin steps, a sweep of neighbour swaps).  Here the ladder is spread over ranks
 * (one handle per GPU holds `n_replicas` consecutive replicas of the `n_total`); an exchange is: every rank leaves the
 * potential energies of its replicas in device memory, ONE all-gather (RCCL) assembles all n_total of them on every rank,
 * every rank runs the same sweep kernel on them and arrives at the same ladder assignment; temperatures, velocity factors
 * sqrt(T_new/T_old) and the bookkeeping stay on the device.  Nothing crosses PCIe per exchange.
 *   awsem_remd_setup     ladder_T[n_total] (K); gather_index[n_total] = where replica r sits in the gathered energy buffer (NULL:
 *                        r itself); local_start = global index of this handle's replica 0.  Replica r starts at ladder_T[r].
 *   awsem_step_remd      awsem_step with the temperatures the ladder assignment on the device currently gives the local replicas
 *   awsem_remd_energies  potential energy (masked groups) of the local replicas -> d_energy[n_replicas] (a DEVICE pointer; NULL =
 *                        the handle's own n_total buffer at local_start, enough when one rank holds the whole ladder)
 *   awsem_remd_sweep     one sweep on d_all[.] (DEVICE pointer to the gathered energies; NULL = the handle's own buffer): pairs
 *                        (i, i+1) with i of the parity of `iteration`, accepted when delta = (beta_i - beta_i+1)(U_a - U_b) >= 0
 *                        or u < exp(delta), u = Philox4x32-10(counter (i, iteration, 'REMD'), key seed) -- the stream
 *                        openawsem_amd/ensemble.py:counter_uniform reproduces on the host; then rescales the local velocities
 *                        (as a factor per replica that the next step launch multiplies in when it loads them; every other call
 *                        that reads or writes velocities applies it first, so callers always see the rescaled velocities)
 *   awsem_remd_set_mixing  0 (default): the neighbour sweep above.  1: openmmtools' DEFAULT scheme 'swap-all', which the reference gets by
 *                        passing no replica_mixing_scheme (scripts/mm_remd.py:133-140): n_total^3 attempts per iteration on uniformly
 *                        drawn replica pairs (i, j), each accepted when (beta_si - beta_sj)(U_i - U_j) >= 0 or ln u < that; draws from
 *                        Philox4x32-10(counter (attempt, iteration, 'RMSA'), key seed) (ensemble.py:_swap_all_draws on the host).  A
 *                        chain of dependent decisions: ~0.1 us per attempt on one thread (12 replicas: 0.2 ms; 64: 27 ms)
 *   awsem_remd_state     copies out the ladder index of every replica, attempt / acceptance counts per neighbouring pair and
 *                        the local temperatures (any pointer may be NULL); the only call here that waits for the device
 * awsem_remd_energies and awsem_remd_sweep are asynchronous on the handle's stream; a caller that gathers with another
 * runtime's stream (torch.distributed) calls awsem_synchronize before the gather and waits for the gather before the sweep. */
int awsem_remd_setup(awsem_handle h, int n_total, const double* ladder_T, const int32_t* gather_index, int local_start,
                     uint64_t seed);
int awsem_step_remd(awsem_handle h, int nsteps, double dt_ps, double friction_per_ps, uint64_t seed);
int awsem_remd_energies(awsem_handle h, uint32_t group_mask, double* d_energy);
int awsem_remd_sweep(awsem_handle h, const double* d_all_energies, int64_t iteration);
int awsem_remd_set_mixing(awsem_handle h, int scheme);
int awsem_remd_state(awsem_handle h, int32_t* ladder_index, int64_t* attempted, int64_t* accepted, double* T_local);

/* ---- allocation-free evaluation ---------------------------------------------------------------------------
 * NO COUNTERPART IN THE REFERENCE.  A variant of awsem_energy_by_group for callers that cannot allocate device
 * memory: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It runs the same
 * kernels on caller-owned DEVICE buffers and keeps everything else it needs (the per-replica Verlet pair lists of the
 * pair terms and the device status word) in a file-scope `static __device__` array of AWSEM_NOALLOC_SCRATCH_BYTES that
 * every GPU the library runs on holds until the process ends.  The call allocates and frees nothing, copies nothing to
 * or from the host and does not synchronise: it enqueues one memset node and one kernel on `hip_stream` (a hipStream_t
 * of the HIP runtime this library is linked against; NULL = the null stream) and returns.
 *   d_xyz     device pointer, n_replicas x natoms x 3 doubles (nm), the caller's atom order
 *   d_energy  device pointer, n_replicas x 32 doubles: the energy of every force group, as awsem_energy_by_group
 * n_replicas is free per call (it need not equal the handle's): the handle contributes the finalized system only.
 * Limits: replica-resident handles only (AWSEM_ESTATE on a cell-list handle); 256 + n_replicas x (pair-list entries
 * reserved per replica) x 4 bytes must fit the array, else AWSEM_ELIMIT with the numbers in awsem_last_error().
 * All calls on one GPU share the one array (it keeps the previous call's lists): calls that overlap in time on the same
 * GPU -- two streams, two handles, two host threads -- race; serialise them.
 * Errors: argument and size errors are returned before anything is queued.  DEVICE-side conditions are NOT reported on this path:
 * the status word the kernel writes lives in the static array, is cleared by the call's memset node and is read by nobody (reading
 * it would be a copy and a synchronisation); the only condition it can hold for an energy evaluation is a pair-list overflow, which
 * the list capacities (= the number of candidate pairs) rule out.  The caller's current device is left as it was found.
 * Safe under stream capture (hipStreamBeginCapture ... EndCapture, replayed with hipGraphLaunch): tests/test_zz_no_alloc.py. */
#define AWSEM_NOALLOC_SCRATCH_BYTES (32u << 20)
int awsem_energy_by_group_noalloc(awsem_handle h, int n_replicas, const double* d_xyz, double* d_energy,
                                  void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* AWSEM_HIP_H */
