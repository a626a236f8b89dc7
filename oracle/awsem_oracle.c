/*
 * awsem_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C fp64 CPU restatement of the AWSEM force field as the reference (epochbio/openawsem) configures it on
 * top of OpenMM, plus OpenMM's LangevinIntegrator update.  It is the checker for the HIP library
 * (libawsem_hip.so): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * The arithmetic of the reference's hot path runs inside OpenMM (third-party, un-vendored, version unpinned in
 * the reference's requirements.txt:5 / setup.py:52), which is not available here; the formulas below are the
 * algebraic expressions the reference's term factories hand to OpenMM, restated with analytic gradients.
 * Parity is PINNED by the reference's own golden vectors tests/data/{1brs,1mbn,1ubq,2lyz,2lzm}_energies.csv
 * (OpenMM Reference platform, 12 frames x 7 non-zero columns each; see tests/test_oracle_golden.py).
 * Forces, Debye-Hueckel, membrane, standalone burial, z-dependent contact and the Langevin step have no
 * reference fixture ("parity unpinned"): forces are checked by central finite differences of these energies.
 *
 * The exported functions mirror include/awsem_hip.h one to one with the prefix awsem_oracle_.
 * All paths cited are relative to the reference checkout.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NGROUPS 32
#define KB 0.00831446261815324 /* kJ/mol/K, OpenMM's BOLTZ */

typedef struct { int group, n; int *pairs; double *r0, *k; } harm_t;
typedef struct { int group, n; int *atoms; double k, chi0; } chi_t;
typedef struct { int group, n, nwell; int *atoms; double *wp, *bw; double k; } rama_t;
typedef struct { int group, nsites, nexcl; int *atom, *cls, *excl; double k, r0; uint64_t* keys; /* sorted (min << 32 | max) of the excluded pairs */ } excl_t;
typedef struct {
    int group, nsites, env, min_seq_sep, flags, has_burial;
    int *atom, *restype, *resid, *chain;
    double p[32];
    double gamma[2 * 3 * 400];
    double bgamma[120];   /* [20][3]; with flag 8 a second [20][3] block for the membrane */
    double *gamma_idx, *bgamma_idx;   /* flag 16: [3][nsites][nsites] and [nsites][3] by site index */
} contact_t;
/* functional form of a lambda / gamma table (awsem_add_hbond_beta_fn / _pap_fn): info = [nres][4] type, chain, edge, helix */
typedef struct { int kind, nres; int* info; double C[5][3], G[3], T1[20][20], T2[20][20], O[20]; } hbfn_t;
typedef struct {
    int group, two_theta, nacc, ndon, nres;
    int *acc_atoms, *acc_res, *don_atoms, *don_res;
    hbfn_t* fn;
    double *lambda;
    double k, r_ON, s_NO, r_OH, s_HO, cutoff;
} beta_t;
typedef struct {
    int group, nacc, ndon, nres;
    int *acc_atoms, *acc_res, *don_atoms, *don_res;
    hbfn_t* fn;
    double *gamma;
    double k, eta, r0, d_i4, cutoff;
} pap_t;
typedef struct { int group, n; int *atoms; double *coef; double r_ON, s_NO, r_OH, s_HO;
                 int *ca; double *coef_m; double eta, z_m, center; } helical_t;   /* ca != NULL: z_dependent_helical_term */
typedef struct { int group, n, ntab; int *pairs; double *table; double k, dr; } fm_t;
typedef struct { int group, n; int *atoms; double *scale; double k, k_m, z_m, center; int kind; } memb_t;   /* kind 0: slab; 1-3: -k s x|y|z; 4: k|z-c-s| */
typedef struct { int group, n; int *pairs; double *qq; double k, lam; } dh_t;
typedef struct { int group, n; int *pairs; double *r0, *sigma, *kg; int kind, cv; double cv_k, cv_x0; } gauss_t;

enum { T_HARM, T_CHI, T_RAMA, T_EXCL, T_CONTACT, T_BETA, T_PAP, T_HELICAL, T_FM, T_MEMB, T_DH, T_GAUSS };
typedef struct { int kind; void* p; } term_t;

typedef struct awsem_oracle_ctx {
    int natoms, nrep, finalized;
    double* mass;
    int nvs; int* vs_site; int* vs_par; double* vs_w;
    int nterms, cap; term_t* terms;
    double* pos; /* [rep][natoms][3] */
    double* vel;
    int64_t step;
    int nthreads;
    int precision;   /* 0 = fp32, 1 = fp64: only awsem_oracle_neighbors looks at it (distance screen in that precision) */
    int remove_cm;   /* CMMotionRemover present (ForceField.createSystem default, openAWSEM.py:826,878) */
} ctx_t;

static __thread char g_err[512];
static int fail(int code, const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); return code; }
const char* awsem_oracle_last_error(void) { return g_err; }
const char* awsem_oracle_version(void) { return "awsem_oracle 0.1 fp64-cpu"; }

static void* dup_mem(const void* src, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (src && bytes) memcpy(p, src, bytes);
    return p;
}
static int check_atoms(ctx_t* c, const int* idx, int n, int allow_neg) {
    for (int i = 0; i < n; i++) {
        if (idx[i] < 0) { if (!allow_neg || idx[i] != -1) return 0; }
        else if (idx[i] >= c->natoms) return 0;
    }
    return 1;
}
static int add_term(ctx_t* c, int kind, void* p) {
    if (c->nterms == c->cap) { c->cap = c->cap ? 2 * c->cap : 16; c->terms = realloc(c->terms, c->cap * sizeof(term_t)); }
    c->terms[c->nterms].kind = kind; c->terms[c->nterms].p = p; c->nterms++;
    return 0;
}
#define CHECK_BUILD(c, group) \
    if (!(c)) return fail(-1, "null handle"); \
    if ((c)->finalized) return fail(-2, "system already finalized"); \
    if ((group) < 0 || (group) >= NGROUPS) return fail(-1, "force group out of range 0..31");

int awsem_oracle_create(int natoms, int nrep, int device, int precision, ctx_t** out) {
    (void)device;
    if (natoms <= 0 || nrep <= 0 || !out) return fail(-1, "awsem_create: natoms and n_replicas must be positive");
    ctx_t* c = calloc(1, sizeof(ctx_t));
    c->natoms = natoms; c->nrep = nrep;
    c->mass = calloc(natoms, sizeof(double));
    for (int i = 0; i < natoms; i++) c->mass[i] = 1.0;
    c->pos = calloc((size_t)nrep * natoms * 3, sizeof(double));
    c->vel = calloc((size_t)nrep * natoms * 3, sizeof(double));
    c->nthreads = 1;
    c->precision = precision;
    *out = c;
    return 0;
}

static void free_term(term_t* t) {
    void* p = t->p;
    switch (t->kind) {
    case T_HARM: { harm_t* x = p; free(x->pairs); free(x->r0); free(x->k); } break;
    case T_CHI: { chi_t* x = p; free(x->atoms); } break;
    case T_RAMA: { rama_t* x = p; free(x->atoms); free(x->wp); free(x->bw); } break;
    case T_EXCL: { excl_t* x = p; free(x->atom); free(x->cls); free(x->excl); free(x->keys); } break;
    case T_CONTACT: { contact_t* x = p; free(x->atom); free(x->restype); free(x->resid); free(x->chain); free(x->gamma_idx); free(x->bgamma_idx); } break;
    case T_BETA: { beta_t* x = p; if (x->fn) { free(x->fn->info); free(x->fn); } free(x->acc_atoms); free(x->acc_res); free(x->don_atoms); free(x->don_res); free(x->lambda); } break;
    case T_PAP: { pap_t* x = p; if (x->fn) { free(x->fn->info); free(x->fn); } free(x->acc_atoms); free(x->acc_res); free(x->don_atoms); free(x->don_res); free(x->gamma); } break;
    case T_HELICAL: { helical_t* x = p; free(x->atoms); free(x->coef); free(x->ca); free(x->coef_m); } break;
    case T_FM: { fm_t* x = p; free(x->pairs); free(x->table); } break;
    case T_MEMB: { memb_t* x = p; free(x->atoms); free(x->scale); } break;
    case T_DH: { dh_t* x = p; free(x->pairs); free(x->qq); } break;
    case T_GAUSS: { gauss_t* x = p; free(x->pairs); free(x->r0); free(x->sigma); free(x->kg); } break;   /* free(NULL) is fine for the r^2 kind */
    }
    free(p);
}
void awsem_oracle_destroy(ctx_t* c) {
    if (!c) return;
    for (int i = 0; i < c->nterms; i++) free_term(&c->terms[i]);
    free(c->terms); free(c->mass); free(c->vs_site); free(c->vs_par); free(c->vs_w); free(c->pos); free(c->vel);
    free(c);
}
int awsem_oracle_set_threads(ctx_t* c, int n) { if (!c || n < 1) return fail(-1, "bad thread count"); c->nthreads = n; return 0; }

int awsem_oracle_set_cm_motion_removal(ctx_t* c, int enabled) {
    CHECK_BUILD(c, 0);
    c->remove_cm = enabled != 0;
    return 0;
}
int awsem_oracle_set_masses(ctx_t* c, const double* m, int natoms) {
    CHECK_BUILD(c, 0);
    if (!m || natoms != c->natoms) return fail(-1, "set_masses: natoms mismatch");
    for (int i = 0; i < natoms; i++) { if (!(m[i] >= 0)) return fail(-1, "set_masses: negative mass"); c->mass[i] = m[i]; }
    return 0;
}
int awsem_oracle_set_virtual_sites(ctx_t* c, int n, const int* site, const int* par, const double* w) {
    CHECK_BUILD(c, 0);
    if (n < 0 || (n && (!site || !par || !w))) return fail(-1, "set_virtual_sites: bad arguments");
    if (!check_atoms(c, site, n, 0) || !check_atoms(c, par, 3 * n, 0)) return fail(-1, "set_virtual_sites: atom index out of range");
    free(c->vs_site); free(c->vs_par); free(c->vs_w);
    c->nvs = n; c->vs_site = dup_mem(site, n * sizeof(int)); c->vs_par = dup_mem(par, 3 * n * sizeof(int));
    c->vs_w = dup_mem(w, 3 * n * sizeof(double));
    return 0;
}

int awsem_oracle_add_harmonic_bonds(ctx_t* c, int group, int n, const int* pairs, const double* r0, const double* k) {
    CHECK_BUILD(c, group);
    if (n < 0 || (n && (!pairs || !r0 || !k)) || !check_atoms(c, pairs, 2 * n, 0)) return fail(-1, "add_harmonic_bonds: bad arguments");
    harm_t* t = calloc(1, sizeof *t); t->group = group; t->n = n;
    t->pairs = dup_mem(pairs, 2 * n * sizeof(int)); t->r0 = dup_mem(r0, n * sizeof(double)); t->k = dup_mem(k, n * sizeof(double));
    return add_term(c, T_HARM, t);
}
int awsem_oracle_add_chi(ctx_t* c, int group, int n, const int* atoms, double k, double chi0) {
    CHECK_BUILD(c, group);
    if (n < 0 || (n && !atoms) || !check_atoms(c, atoms, 4 * n, 0)) return fail(-1, "add_chi: bad arguments");
    chi_t* t = calloc(1, sizeof *t); t->group = group; t->n = n; t->atoms = dup_mem(atoms, 4 * n * sizeof(int)); t->k = k; t->chi0 = chi0;
    return add_term(c, T_CHI, t);
}
int awsem_oracle_add_rama(ctx_t* c, int group, int n, const int* atoms, int nwell, const double* wp, const double* bw, double k) {
    CHECK_BUILD(c, group);
    if (n < 0 || nwell < 1 || nwell > 8 || !wp || (n && !atoms) || !check_atoms(c, atoms, 5 * n, 0)) return fail(-1, "add_rama: bad arguments");
    rama_t* t = calloc(1, sizeof *t); t->group = group; t->n = n; t->nwell = nwell; t->k = k;
    t->atoms = dup_mem(atoms, 5 * n * sizeof(int)); t->wp = dup_mem(wp, 6 * nwell * sizeof(double));
    t->bw = malloc(((size_t)n * nwell + 1) * sizeof(double));
    for (int i = 0; i < n * nwell; i++) t->bw[i] = bw ? bw[i] : 1.0;
    return add_term(c, T_RAMA, t);
}
/* createExclusionsFromBonds(oa.bonds, 1) (basicTerms.py:189): is the atom pair (a, b) one of the excluded ones?  Binary search in
 * the sorted pair keys (a linear scan per candidate pair is fine for a protein, not for the 20 000-residue system). */
static uint64_t pair_key(int a, int b) { return a < b ? ((uint64_t)a << 32 | (uint32_t)b) : ((uint64_t)b << 32 | (uint32_t)a); }
static int cmp_u64(const void* x, const void* y) { uint64_t a = *(const uint64_t*)x, b = *(const uint64_t*)y; return a < b ? -1 : a > b; }
static int is_excluded(const uint64_t* keys, int n, int a, int b) {
    uint64_t k = pair_key(a, b);
    return bsearch(&k, keys, n, sizeof(uint64_t), cmp_u64) != NULL;
}
int awsem_oracle_add_excl(ctx_t* c, int group, int nsites, const int* atom, const int* cls, int nexcl, const int* excl, double k, double r0) {
    CHECK_BUILD(c, group);
    if (nsites < 0 || nexcl < 0 || (nsites && (!atom || !cls)) || (nexcl && !excl) || !check_atoms(c, atom, nsites, 0) || !check_atoms(c, excl, 2 * nexcl, 0))
        return fail(-1, "add_excl: bad arguments");
    excl_t* t = calloc(1, sizeof *t); t->group = group; t->nsites = nsites; t->nexcl = nexcl; t->k = k; t->r0 = r0;
    t->atom = dup_mem(atom, nsites * sizeof(int)); t->cls = dup_mem(cls, nsites * sizeof(int)); t->excl = dup_mem(excl, 2 * nexcl * sizeof(int));
    t->keys = malloc((nexcl ? nexcl : 1) * sizeof(uint64_t));
    for (int q = 0; q < nexcl; q++) t->keys[q] = pair_key(excl[2 * q], excl[2 * q + 1]);
    qsort(t->keys, nexcl, sizeof(uint64_t), cmp_u64);
    return add_term(c, T_EXCL, t);
}
int awsem_oracle_add_contact(ctx_t* c, int group, int nsites, const int* atom, const int* restype, const int* resid, const int* chain,
                             int env, int min_seq_sep, int flags, const double* dparams, const double* gamma, const double* bgamma) {
    CHECK_BUILD(c, group);
    if (nsites < 0 || !dparams || (nsites && (!atom || !restype || !resid || !chain)) || !check_atoms(c, atom, nsites, 0) || env < 0 || env > 1)
        return fail(-1, "add_contact: bad arguments");
    if (!(flags & 2) && !gamma) return fail(-1, "add_contact: gamma table required");
    for (int i = 0; i < nsites; i++) if (restype[i] < 0 || restype[i] >= 20) return fail(-1, "add_contact: residue type out of range 0..19");
    contact_t* t = calloc(1, sizeof *t); t->group = group; t->nsites = nsites; t->env = env; t->min_seq_sep = min_seq_sep; t->flags = flags;
    t->atom = dup_mem(atom, nsites * sizeof(int)); t->restype = dup_mem(restype, nsites * sizeof(int));
    t->resid = dup_mem(resid, nsites * sizeof(int)); t->chain = dup_mem(chain, nsites * sizeof(int));
    memcpy(t->p, dparams, 20 * sizeof(double));
    t->has_burial = bgamma != NULL;
    if (flags & 16) {   /* index_based_contact_term: tables by site index */
        if (flags & (2 | 8)) return fail(-1, "add_contact: the index-based variant has its own burial table and no burial-only form");
        /* [3][n][n], or [2][3][n][n] (water, membrane) for the z-dependent blend, contactTerms.py:820-831 */
        t->gamma_idx = dup_mem(gamma, ((flags & 1) ? 6 : 3) * (size_t)nsites * nsites * sizeof(double));
        if (bgamma) t->bgamma_idx = dup_mem(bgamma, 3 * (size_t)nsites * sizeof(double));
        return add_term(c, T_CONTACT, t);
    }
    if (gamma) memcpy(t->gamma, gamma, sizeof t->gamma);
    if (bgamma) memcpy(t->bgamma, bgamma, ((flags & 8) ? 120 : 60) * sizeof(double));
    return add_term(c, T_CONTACT, t);
}
/* The lambda / gamma entry of (donor residue i, acceptor residue j) from its defining formulas instead of a dense table:
 *   lambda_1: get_lambda_by_index(i, j, 0)                                  hydrogenBondTerms.py:160-184
 *   lambda_2: get_Lambda_2 (300-328), lambda_3: get_Lambda_3 (331-359); both zero when i or j is a chain edge (513-519, 577-583)
 *   P-AP: get_pap_gamma_APH + get_pap_gamma_AP (216-267) or get_pap_gamma_P (270-297)
 * C holds the class tables with the signs and the 0.5 / 0.25 factors of the reference folded in (see include/awsem_hip.h). */
static double hbfn_value(const hbfn_t* f, int i, int j) {
    const int* I = f->info + 4 * (size_t)i; const int* J = f->info + 4 * (size_t)j;
    if (f->kind <= 3) {
        int d = abs(i - j), c;
        if (d < 4) return 0.0;
        c = d >= 45 ? 2 : (d >= 18 ? 1 : 0);
        if (f->kind == 1) return f->C[0][c];
        if (I[2] || J[2]) return 0.0;
        int ai = I[0], aj = J[0], ain = I[4], aip = I[-4], ajn = J[4], ajp = J[-4];   /* i, j are no chain edges: i+-1, j+-1 exist */
        double L = f->C[0][c];
        if (f->kind == 2) {
            L += f->C[1][c] * f->T1[ai][aj];
            L += f->C[2][c] * (f->T2[ain][ajp] + f->T2[aip][ajn]);
            L += f->C[3][c] * (f->O[ai] + f->O[aj]);
        } else {
            L += f->C[1][c] * f->T1[ain][aj];
            L += f->C[2][c] * f->O[ain];
            L += f->C[1][c] * f->O[aj];
        }
        return L;
    }
    int d = i - j, other = I[1] != J[1];
    double s = (I[3] && J[3]) ? 1.5 : 1.0;
    if (f->kind == 4) return ((d >= 13 && d <= 16 && !other) ? f->G[0] : 0.0) + ((d >= 17 || other) ? s * f->G[1] : 0.0);
    return (d >= 9 || other) ? s * f->G[2] : 0.0;
}
static hbfn_t* make_fn(int kind, int nres, const int* info, const double* coef) {
    hbfn_t* f = calloc(1, sizeof *f);
    f->kind = kind; f->nres = nres; f->info = dup_mem(info, 4 * (size_t)nres * sizeof(int));
    memcpy(f->C, coef, 15 * sizeof(double)); memcpy(f->G, coef + 15, 3 * sizeof(double));
    memcpy(f->T1, coef + 18, 400 * sizeof(double)); memcpy(f->T2, coef + 418, 400 * sizeof(double)); memcpy(f->O, coef + 818, 20 * sizeof(double));
    return f;
}
/* test hook: the whole table a functional form stands for, row = donor residue (checked against the dense builders) */
int awsem_oracle_hb_fn_table(int kind, int nres, const int* info, const double* coef, double* out) {
    if (!info || !coef || !out || kind < 1 || kind > 5 || nres <= 0) return fail(-1, "hb_fn_table: bad arguments");
    hbfn_t* f = make_fn(kind, nres, info, coef);
    for (int i = 0; i < nres; i++) for (int j = 0; j < nres; j++) out[(size_t)i * nres + j] = hbfn_value(f, i, j);
    free(f->info); free(f);
    return 0;
}
static double beta_lambda(const beta_t* t, int q, int p) { return t->fn ? hbfn_value(t->fn, q, p) : t->lambda[(size_t)q * t->nres + p]; }
static double pap_gamma(const pap_t* t, int q, int p) { return t->fn ? hbfn_value(t->fn, q, p) : t->gamma[(size_t)q * t->nres + p]; }
static int add_beta(ctx_t* c, int group, int two_theta, int nacc, const int* acc_atoms, const int* acc_res, int ndon,
                    const int* don_atoms, const int* don_res, int nres, const double* lambda, hbfn_t* fn, double k, double r_ON,
                    double s_NO, double r_OH, double s_HO, double cutoff) {
    CHECK_BUILD(c, group);
    if (nacc < 0 || ndon < 0 || nres <= 0 || (!lambda && !fn) || (nacc && (!acc_atoms || !acc_res)) || (ndon && (!don_atoms || !don_res)))
        return fail(-1, "add_hbond_beta: bad arguments");
    if (!check_atoms(c, acc_atoms, 3 * nacc, 1) || !check_atoms(c, don_atoms, 3 * ndon, 1)) return fail(-1, "add_hbond_beta: atom index out of range");
    for (int i = 0; i < nacc; i++) {
        if (acc_res[i] < 0 || acc_res[i] >= nres || acc_atoms[3 * i] < 0) return fail(-1, "add_hbond_beta: bad acceptor");
        if (two_theta && (acc_atoms[3 * i + 1] < 0 || acc_atoms[3 * i + 2] < 0)) return fail(-1, "add_hbond_beta: acceptor needs a2,a3");
    }
    for (int i = 0; i < ndon; i++) {
        if (don_res[i] < 0 || don_res[i] >= nres || don_atoms[3 * i] < 0 || don_atoms[3 * i + 1] < 0) return fail(-1, "add_hbond_beta: bad donor");
        if (two_theta && don_atoms[3 * i + 2] < 0) return fail(-1, "add_hbond_beta: donor needs d3");
    }
    beta_t* t = calloc(1, sizeof *t); t->group = group; t->two_theta = two_theta; t->nacc = nacc; t->ndon = ndon; t->nres = nres;
    t->acc_atoms = dup_mem(acc_atoms, 3 * nacc * sizeof(int)); t->acc_res = dup_mem(acc_res, nacc * sizeof(int));
    t->don_atoms = dup_mem(don_atoms, 3 * ndon * sizeof(int)); t->don_res = dup_mem(don_res, ndon * sizeof(int));
    t->lambda = lambda ? dup_mem(lambda, (size_t)nres * nres * sizeof(double)) : NULL;
    t->fn = fn;
    t->k = k; t->r_ON = r_ON; t->s_NO = s_NO; t->r_OH = r_OH; t->s_HO = s_HO; t->cutoff = cutoff;
    return add_term(c, T_BETA, t);
}
int awsem_oracle_add_hbond_beta(ctx_t* c, int group, int two_theta, int nacc, const int* acc_atoms, const int* acc_res, int ndon,
                                const int* don_atoms, const int* don_res, int nres, const double* lambda, double k, double r_ON,
                                double s_NO, double r_OH, double s_HO, double cutoff) {
    if (!lambda) return fail(-1, "add_hbond_beta: bad arguments");
    return add_beta(c, group, two_theta, nacc, acc_atoms, acc_res, ndon, don_atoms, don_res, nres, lambda, NULL, k, r_ON, s_NO, r_OH, s_HO, cutoff);
}
static int fn_ok(int kind, int lo, int hi, int nres, const int* info, const double* coef) {
    if (!info || !coef || kind < lo || kind > hi || nres <= 0) return 0;
    for (int i = 0; i < nres; i++) if (info[4 * i] < 0 || info[4 * i] >= 20 || info[4 * i + 1] < 0) return 0;
    return 1;
}
int awsem_oracle_add_hbond_beta_fn(ctx_t* c, int group, int two_theta, int nacc, const int* acc_atoms, const int* acc_res, int ndon,
                                   const int* don_atoms, const int* don_res, int nres, int kind, const int* info, const double* coef,
                                   double k, double r_ON, double s_NO, double r_OH, double s_HO, double cutoff) {
    if (!fn_ok(kind, 1, 3, nres, info, coef)) return fail(-1, "add_hbond_beta_fn: bad arguments");
    hbfn_t* fn = make_fn(kind, nres, info, coef);
    int rc = add_beta(c, group, two_theta, nacc, acc_atoms, acc_res, ndon, don_atoms, don_res, nres, NULL, fn, k, r_ON, s_NO, r_OH, s_HO, cutoff);
    if (rc) { free(fn->info); free(fn); }
    return rc;
}
static int add_pap(ctx_t* c, int group, int nacc, const int* acc_atoms, const int* acc_res, int ndon, const int* don_atoms,
                   const int* don_res, int nres, const double* gamma, hbfn_t* fn, double k, double eta, double r0, double d_i4, double cutoff) {
    CHECK_BUILD(c, group);
    if (nacc < 0 || ndon < 0 || nres <= 0 || (!gamma && !fn) || (nacc && (!acc_atoms || !acc_res)) || (ndon && (!don_atoms || !don_res)))
        return fail(-1, "add_hbond_pap: bad arguments");
    if (!check_atoms(c, acc_atoms, 2 * nacc, 0) || !check_atoms(c, don_atoms, 2 * ndon, 0)) return fail(-1, "add_hbond_pap: atom index out of range");
    for (int i = 0; i < nacc; i++) if (acc_res[i] < 0 || acc_res[i] >= nres) return fail(-1, "add_hbond_pap: bad acceptor residue");
    for (int i = 0; i < ndon; i++) if (don_res[i] < 0 || don_res[i] >= nres) return fail(-1, "add_hbond_pap: bad donor residue");
    pap_t* t = calloc(1, sizeof *t); t->group = group; t->nacc = nacc; t->ndon = ndon; t->nres = nres;
    t->acc_atoms = dup_mem(acc_atoms, 2 * nacc * sizeof(int)); t->acc_res = dup_mem(acc_res, nacc * sizeof(int));
    t->don_atoms = dup_mem(don_atoms, 2 * ndon * sizeof(int)); t->don_res = dup_mem(don_res, ndon * sizeof(int));
    t->gamma = gamma ? dup_mem(gamma, (size_t)nres * nres * sizeof(double)) : NULL;
    t->fn = fn;
    t->k = k; t->eta = eta; t->r0 = r0; t->d_i4 = d_i4; t->cutoff = cutoff;
    return add_term(c, T_PAP, t);
}
int awsem_oracle_add_hbond_pap(ctx_t* c, int group, int nacc, const int* acc_atoms, const int* acc_res, int ndon, const int* don_atoms,
                               const int* don_res, int nres, const double* gamma, double k, double eta, double r0, double d_i4, double cutoff) {
    if (!gamma) return fail(-1, "add_hbond_pap: bad arguments");
    return add_pap(c, group, nacc, acc_atoms, acc_res, ndon, don_atoms, don_res, nres, gamma, NULL, k, eta, r0, d_i4, cutoff);
}
int awsem_oracle_add_hbond_pap_fn(ctx_t* c, int group, int nacc, const int* acc_atoms, const int* acc_res, int ndon, const int* don_atoms,
                                  const int* don_res, int nres, int kind, const int* info, const double* coef, double k, double eta, double r0,
                                  double d_i4, double cutoff) {
    if (!fn_ok(kind, 4, 5, nres, info, coef)) return fail(-1, "add_hbond_pap_fn: bad arguments");
    hbfn_t* fn = make_fn(kind, nres, info, coef);
    int rc = add_pap(c, group, nacc, acc_atoms, acc_res, ndon, don_atoms, don_res, nres, NULL, fn, k, eta, r0, d_i4, cutoff);
    if (rc) { free(fn->info); free(fn); }
    return rc;
}
int awsem_oracle_add_helical(ctx_t* c, int group, int n, const int* atoms, const double* coef, double r_ON, double s_NO, double r_OH, double s_HO) {
    CHECK_BUILD(c, group);
    if (n < 0 || (n && (!atoms || !coef)) || !check_atoms(c, atoms, 3 * n, 0)) return fail(-1, "add_helical: bad arguments");
    helical_t* t = calloc(1, sizeof *t); t->group = group; t->n = n; t->atoms = dup_mem(atoms, 3 * n * sizeof(int));
    t->coef = dup_mem(coef, n * sizeof(double)); t->r_ON = r_ON; t->s_NO = s_NO; t->r_OH = r_OH; t->s_HO = s_HO;
    return add_term(c, T_HELICAL, t);
}
int awsem_oracle_add_helical_z(ctx_t* c, int group, int n, const int* atoms4, const double* coef, const double* coef_m, double r_ON, double s_NO,
                               double r_OH, double s_HO, double eta, double z_m, double center) {
    CHECK_BUILD(c, group);
    if (n < 0 || (n && (!atoms4 || !coef || !coef_m)) || !check_atoms(c, atoms4, 4 * n, 0)) return fail(-1, "add_helical_z: bad arguments");
    helical_t* t = calloc(1, sizeof *t); t->group = group; t->n = n;
    t->atoms = malloc((3 * n + 1) * sizeof(int)); t->ca = malloc((n + 1) * sizeof(int));
    for (int b = 0; b < n; b++) { for (int k = 0; k < 3; k++) t->atoms[3 * b + k] = atoms4[4 * b + k]; t->ca[b] = atoms4[4 * b + 3]; }
    t->coef = dup_mem(coef, n * sizeof(double)); t->coef_m = dup_mem(coef_m, n * sizeof(double));
    t->r_ON = r_ON; t->s_NO = s_NO; t->r_OH = r_OH; t->s_HO = s_HO; t->eta = eta; t->z_m = z_m; t->center = center;
    return add_term(c, T_HELICAL, t);
}
int awsem_oracle_add_fragment_memory(ctx_t* c, int group, int nb, const int* pairs, int ntab, const double* table, double k, double dr) {
    CHECK_BUILD(c, group);
    if (nb < 0 || ntab < 2 || !(dr > 0) || (nb && (!pairs || !table)) || !check_atoms(c, pairs, 2 * nb, 0)) return fail(-1, "add_fragment_memory: bad arguments");
    fm_t* t = calloc(1, sizeof *t); t->group = group; t->n = nb; t->ntab = ntab; t->k = k; t->dr = dr;
    t->pairs = dup_mem(pairs, 2 * nb * sizeof(int)); t->table = dup_mem(table, (size_t)nb * ntab * sizeof(double));
    return add_term(c, T_FM, t);
}
int awsem_oracle_add_external(ctx_t* c, int group, int kind, int n, const int* atoms, const double* param, double k, double center) {
    CHECK_BUILD(c, group);
    if (kind < 1 || kind > 4) return fail(-1, "add_external: unknown kind");
    if (n < 0 || (n && (!atoms || !param)) || !check_atoms(c, atoms, n, 0)) return fail(-1, "add_external: bad arguments");
    memb_t* t = calloc(1, sizeof *t); t->group = group; t->n = n; t->atoms = dup_mem(atoms, n * sizeof(int)); t->scale = dup_mem(param, n * sizeof(double));
    t->k = k; t->center = center; t->kind = kind;
    return add_term(c, T_MEMB, t);
}
int awsem_oracle_add_membrane(ctx_t* c, int group, int n, const int* atoms, const double* scale, double k, double k_m, double z_m, double center) {
    CHECK_BUILD(c, group);
    if (n < 0 || (n && (!atoms || !scale)) || !check_atoms(c, atoms, n, 0)) return fail(-1, "add_membrane: bad arguments");
    memb_t* t = calloc(1, sizeof *t); t->group = group; t->n = n; t->atoms = dup_mem(atoms, n * sizeof(int)); t->scale = dup_mem(scale, n * sizeof(double));
    t->k = k; t->k_m = k_m; t->z_m = z_m; t->center = center;
    return add_term(c, T_MEMB, t);
}
int awsem_oracle_add_debye_huckel(ctx_t* c, int group, int n, const int* pairs, const double* qq, double k, double lam) {
    CHECK_BUILD(c, group);
    if (n < 0 || !(lam > 0) || (n && (!pairs || !qq)) || !check_atoms(c, pairs, 2 * n, 0)) return fail(-1, "add_debye_huckel: bad arguments");
    dh_t* t = calloc(1, sizeof *t); t->group = group; t->n = n; t->pairs = dup_mem(pairs, 2 * n * sizeof(int)); t->qq = dup_mem(qq, n * sizeof(double));
    t->k = k; t->lam = lam;
    return add_term(c, T_DH, t);
}
int awsem_oracle_add_gaussian_bonds(ctx_t* c, int group, int n, const int* pairs, const double* r0, const double* sigma, const double* kg) {
    CHECK_BUILD(c, group);
    if (n < 0 || (n && (!pairs || !r0 || !sigma || !kg)) || !check_atoms(c, pairs, 2 * n, 0)) return fail(-1, "add_gaussian_bonds: bad arguments");
    for (int b = 0; b < n; b++) if (!(sigma[b] > 0)) return fail(-1, "add_gaussian_bonds: sigma must be positive");
    gauss_t* t = calloc(1, sizeof *t); t->group = group; t->n = n; t->pairs = dup_mem(pairs, 2 * n * sizeof(int));
    t->r0 = dup_mem(r0, n * sizeof(double)); t->sigma = dup_mem(sigma, n * sizeof(double)); t->kg = dup_mem(kg, n * sizeof(double));
    return add_term(c, T_GAUSS, t);
}
/* q_value / qc_value / partial_q_value, qbias_term, rg_term, rg_bias_term (biasTerms.py:146-302, 461-528):
 * a CustomBondForce sum S, alone (cv 0) or as the collective variable of a CustomCVForce E = f(S) */
int awsem_oracle_add_bond_sum(ctx_t* c, int group, int bond_kind, int n, const int* pairs, const double* r0, const double* sigma,
                              const double* coef, int cv_kind, double cv_k, double cv_x0) {
    CHECK_BUILD(c, group);
    if (bond_kind < 0 || bond_kind > 2 || cv_kind < 0 || cv_kind > 3) return fail(-1, "add_bond_sum: unknown bond or collective-variable kind");
    if (n < 0 || (n && (!pairs || !coef || (bond_kind == 0 && (!r0 || !sigma)))) || !check_atoms(c, pairs, 2 * n, 0)) return fail(-1, "add_bond_sum: bad arguments");
    for (int b = 0; b < n && bond_kind == 0; b++) if (!(sigma[b] > 0)) return fail(-1, "add_bond_sum: sigma must be positive");
    gauss_t* t = calloc(1, sizeof *t); t->group = group; t->n = n; t->pairs = dup_mem(pairs, 2 * n * sizeof(int));
    if (bond_kind == 0) { t->r0 = dup_mem(r0, n * sizeof(double)); t->sigma = dup_mem(sigma, n * sizeof(double)); }
    t->kg = dup_mem(coef, n * sizeof(double));
    for (int b = 0; b < n && bond_kind == 0; b++) t->kg[b] = -t->kg[b];   /* eval_gauss: g = -kg exp(...) */
    t->kind = bond_kind; t->cv = cv_kind; t->cv_k = cv_k; t->cv_x0 = cv_x0;
    return add_term(c, T_GAUSS, t);
}
int awsem_oracle_finalize(ctx_t* c) { if (!c) return fail(-1, "null handle"); c->finalized = 1; return 0; }

/* ------------------------------------------------------------------ state */
#define CHECK_RUN(c) if (!(c)) return fail(-1, "null handle"); if (!(c)->finalized) return fail(-2, "system not finalized");
int awsem_oracle_set_positions(ctx_t* c, int rep, const double* xyz, int natoms) {
    CHECK_RUN(c);
    if (!xyz || natoms != c->natoms || rep < -1 || rep >= c->nrep) return fail(-1, "set_positions: bad arguments");
    size_t n = (size_t)natoms * 3;
    if (rep >= 0) memcpy(c->pos + rep * n, xyz, n * sizeof(double));
    else for (int r = 0; r < c->nrep; r++) memcpy(c->pos + r * n, xyz, n * sizeof(double));
    return 0;
}
int awsem_oracle_get_positions(ctx_t* c, int rep, double* xyz, int natoms) {
    CHECK_RUN(c);
    if (!xyz || natoms != c->natoms || rep < 0 || rep >= c->nrep) return fail(-1, "get_positions: bad arguments");
    memcpy(xyz, c->pos + (size_t)rep * natoms * 3, (size_t)natoms * 3 * sizeof(double));
    return 0;
}
int awsem_oracle_set_velocities(ctx_t* c, int rep, const double* v, int natoms) {
    CHECK_RUN(c);
    if (!v || natoms != c->natoms || rep < -1 || rep >= c->nrep) return fail(-1, "set_velocities: bad arguments");
    size_t n = (size_t)natoms * 3;
    if (rep >= 0) memcpy(c->vel + rep * n, v, n * sizeof(double));
    else for (int r = 0; r < c->nrep; r++) memcpy(c->vel + r * n, v, n * sizeof(double));
    return 0;
}
int awsem_oracle_get_velocities(ctx_t* c, int rep, double* v, int natoms) {
    CHECK_RUN(c);
    if (!v || natoms != c->natoms || rep < 0 || rep >= c->nrep) return fail(-1, "get_velocities: bad arguments");
    memcpy(v, c->vel + (size_t)rep * natoms * 3, (size_t)natoms * 3 * sizeof(double));
    return 0;
}
/* ThreeParticleAverageSite: x = w1 x1 + w2 x2 + w3 x3 (openAWSEM.py:685-696) */
/* all replicas at once: n_replicas x natoms x 3 each, either may be NULL */
int awsem_oracle_set_state(ctx_t* c, const double* xyz, const double* v) {
    CHECK_RUN(c);
    size_t n = (size_t)c->nrep * c->natoms * 3;
    if (xyz) memcpy(c->pos, xyz, n * sizeof(double));
    if (v) memcpy(c->vel, v, n * sizeof(double));
    return 0;
}
int awsem_oracle_get_state(ctx_t* c, double* xyz, double* v) {
    CHECK_RUN(c);
    size_t n = (size_t)c->nrep * c->natoms * 3;
    if (xyz) memcpy(xyz, c->pos, n * sizeof(double));
    if (v) memcpy(v, c->vel, n * sizeof(double));
    return 0;
}
/* sum over the massive particles of 1/2 m v^2 (kJ/mol), one value per replica */
int awsem_oracle_kinetic_energy(ctx_t* c, double* ke) {
    CHECK_RUN(c);
    if (!ke) return fail(-1, "kinetic_energy: null output");
    for (int r = 0; r < c->nrep; r++) {
        const double* v = c->vel + (size_t)r * c->natoms * 3;
        double s = 0;
        for (int i = 0; i < c->natoms; i++) s += 0.5 * c->mass[i] * (v[3 * i] * v[3 * i] + v[3 * i + 1] * v[3 * i + 1] + v[3 * i + 2] * v[3 * i + 2]);
        ke[r] = s;
    }
    return 0;
}
static void virtual_sites(const ctx_t* c, double* x) {
    for (int i = 0; i < c->nvs; i++) {
        const int* p = c->vs_par + 3 * i; const double* w = c->vs_w + 3 * i; double* s = x + 3 * c->vs_site[i];
        for (int d = 0; d < 3; d++) s[d] = w[0] * x[3 * p[0] + d] + w[1] * x[3 * p[1] + d] + w[2] * x[3 * p[2] + d];
    }
}
int awsem_oracle_compute_virtual_sites(ctx_t* c) {
    CHECK_RUN(c);
    for (int r = 0; r < c->nrep; r++) virtual_sites(c, c->pos + (size_t)r * c->natoms * 3);
    return 0;
}

/* ------------------------------------------------------------------ math helpers */
static inline void sub3(const double* a, const double* b, double* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline double dist(const double* x, int i, int j, double* d) { sub3(x + 3 * i, x + 3 * j, d); return sqrt(dot3(d, d)); }
/* adds the force of a pair potential with derivative dEdr along d = x_i - x_j (|d| = r) */
static inline void pair_force(double* f, int i, int j, const double* d, double r, double dEdr) {
    if (!f) return;
    double s = -dEdr / r;
    for (int k = 0; k < 3; k++) { f[3 * i + k] += s * d[k]; f[3 * j + k] -= s * d[k]; }
}
static inline double sech2(double x) { double t = tanh(x); return 1.0 - t * t; }

/* IUPAC dihedral of p1..p4 and its gradient (OpenMM's dihedral(p1,p2,p3,p4): basicTerms.py:300-301). */
static double dihedral(const double* x, const int* a, double g[4][3]) {
    double F[3], G[3], H[3], A[3], B[3];
    sub3(x + 3 * a[0], x + 3 * a[1], F); sub3(x + 3 * a[1], x + 3 * a[2], G); sub3(x + 3 * a[3], x + 3 * a[2], H);
    cross3(F, G, A); cross3(H, G, B);
    double A2 = dot3(A, A), B2 = dot3(B, B), Gn = sqrt(dot3(G, G));
    double cosphi = dot3(A, B) / sqrt(A2 * B2);
    double C[3]; cross3(B, A, C);
    double sinphi = dot3(C, G) / (sqrt(A2 * B2) * Gn);
    double phi = atan2(sinphi, cosphi);
    (void)cosphi;
    if (g) {
        double FG = dot3(F, G), HG = dot3(H, G);
        for (int k = 0; k < 3; k++) {
            double dA = -Gn / A2 * A[k], dB = Gn / B2 * B[k];
            double t = FG / (A2 * Gn) * A[k] - HG / (B2 * Gn) * B[k];
            g[0][k] = dA; g[3][k] = dB; g[1][k] = -dA + t; g[2][k] = -dB - t;
        }
    }
    return phi;
}

/* ------------------------------------------------------------------ term evaluation */
static double eval_harm(const harm_t* t, const double* x, double* f) {
    double e = 0, d[3];
    for (int b = 0; b < t->n; b++) {
        int i = t->pairs[2 * b], j = t->pairs[2 * b + 1];
        double r = dist(x, i, j, d), dr = r - t->r0[b];
        e += 0.5 * t->k[b] * dr * dr;            /* OpenMM HarmonicBondForce */
        pair_force(f, i, j, d, r, t->k[b] * dr);
    }
    return e;
}
/* basicTerms.py:119-129 */
static double eval_chi(const chi_t* t, const double* x, double* f) {
    double e = 0;
    for (int b = 0; b < t->n; b++) {
        const int* a = t->atoms + 4 * b;
        double u[3], v[3], w[3], uv[3], vw[3], wu[3];
        sub3(x + 3 * a[0], x + 3 * a[1], u); sub3(x + 3 * a[2], x + 3 * a[0], v); sub3(x + 3 * a[0], x + 3 * a[3], w);
        cross3(u, v, uv);
        double u2 = dot3(u, u), v2 = dot3(v, v), w2 = dot3(w, w);
        double norm = 1.0 / sqrt(u2 * v2 * w2);
        double chi = dot3(uv, w) * norm;
        double dchi = chi - t->chi0;
        e += t->k * dchi * dchi;
        if (f) {
            cross3(v, w, vw); cross3(w, u, wu);
            double c = -2.0 * t->k * dchi;
            for (int k = 0; k < 3; k++) {
                double gu = vw[k] * norm - chi * u[k] / u2, gv = wu[k] * norm - chi * v[k] / v2, gw = uv[k] * norm - chi * w[k] / w2;
                f[3 * a[0] + k] += c * (gu - gv + gw); f[3 * a[1] + k] += c * (-gu); f[3 * a[2] + k] += c * gv; f[3 * a[3] + k] += c * (-gw);
            }
        }
    }
    return e;
}
/* basicTerms.py:297-303 (rama), 355-361 (proline), 414-420 (ssweight) */
static double eval_rama(const rama_t* t, const double* x, double* f) {
    double e = 0, g1[4][3], g2[4][3];
    for (int b = 0; b < t->n; b++) {
        const int* a = t->atoms + 5 * b;
        double phi = dihedral(x, a, f ? g1 : NULL), psi = dihedral(x, a + 1, f ? g2 : NULL);
        double dEdphi = 0, dEdpsi = 0;
        for (int w = 0; w < t->nwell; w++) {
            const double* p = t->wp + 6 * w;
            double W = p[0] * t->bw[(size_t)b * t->nwell + w];
            double cp = cos(phi - p[3]) - 1, cs = cos(psi - p[5]) - 1;
            double ex = exp(-p[1] * (p[2] * cp * cp + p[4] * cs * cs));
            e += -t->k * W * ex;
            dEdphi += -t->k * W * ex * 2 * p[1] * p[2] * cp * sin(phi - p[3]);
            dEdpsi += -t->k * W * ex * 2 * p[1] * p[4] * cs * sin(psi - p[5]);
        }
        if (f) for (int m = 0; m < 4; m++) for (int k = 0; k < 3; k++) {
            f[3 * a[m] + k] -= dEdphi * g1[m][k];
            f[3 * a[m + 1] + k] -= dEdpsi * g2[m][k];
        }
    }
    return e;
}
/* basicTerms.py:168-190 */
static double eval_excl(const excl_t* t, const double* x, double* f) {
    double e = 0, d[3];
    for (int i = 0; i < t->nsites; i++) for (int j = i + 1; j < t->nsites; j++) {
        if (t->cls[i] != t->cls[j]) continue;
        int a = t->atom[i], b = t->atom[j];
        if (a == b) continue;
        double r = dist(x, a, b, d);
        if (r >= t->r0) continue;
        if (is_excluded(t->keys, t->nexcl, a, b)) continue;
        e += t->k * (r - t->r0) * (r - t->r0);
        pair_force(f, a, b, d, r, 2 * t->k * (r - t->r0));
    }
    return e;
}
/* theta(r;a,b) = 1/4 (1+tanh(eta(r-a)))(1+tanh(eta(b-r))) and d/dr  (contactTerms.py:255,322-323) */
static inline double theta_sw(double r, double a, double b, double eta, double* dth) {
    double t1 = tanh(eta * (r - a)), t2 = tanh(eta * (b - r));
    if (dth) *dth = 0.25 * eta * ((1 - t1 * t1) * (1 + t2) - (1 + t1) * (1 - t2 * t2));
    return 0.25 * (1 + t1) * (1 + t2);
}
/* contactTerms.py:245-351 (CustomGBForce: rho pass, pair energy, burial single-particle energy) */
static double eval_contact(const contact_t* t, const double* x, double* f) {
    const double* p = t->p;
    const double kc = p[0], rmin = p[1], rmax = p[2], rminII = p[3], rmaxII = p[4], eta = p[5], eta_s = p[6], rho0 = p[7];
    const double cutoff = (t->flags & 4) ? 1e300 : p[8], kappa = p[9], eta_sw = p[16], z_m = p[17], zc = p[18], kb = p[19];
    const int n = t->nsites, zdep = t->flags & 1, burial_only = t->flags & 2;
    double* rho = calloc(n, sizeof(double)); double* dEdrho = calloc(n, sizeof(double));
    double* alpha = calloc(n, sizeof(double)); double* dalpha = calloc(n, sizeof(double));
    double e = 0, d[3];
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) {
        if (abs(t->resid[i] - t->resid[j]) < 2) continue;      /* step(abs(resId1-resId2)-2) */
        double r = dist(x, t->atom[i], t->atom[j], d);
        if (r >= cutoff) continue;
        double th = theta_sw(r, rmin, rmax, eta, NULL);
        rho[i] += th; rho[j] += th;
    }
    if (zdep) for (int i = 0; i < n; i++) {
        double z = x[3 * t->atom[i] + 2] - zc;
        alpha[i] = 0.5 * tanh(eta_sw * (z + z_m)) + 0.5 * tanh(eta_sw * (z_m - z));
        dalpha[i] = 0.5 * eta_sw * (sech2(eta_sw * (z + z_m)) - sech2(eta_sw * (z_m - z)));
    }
    if (!burial_only) for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) {
        int same = t->chain[i] == t->chain[j];
        if (same && abs(t->resid[i] - t->resid[j]) < t->min_seq_sep) continue;   /* res_table */
        double r = dist(x, t->atom[i], t->atom[j], d);
        if (r >= cutoff) continue;
        double dth1, dth2;
        double th1 = theta_sw(r, rmin, rmax, eta, &dth1), th2 = theta_sw(r, rminII, rmaxII, eta, &dth2);
        double ti = tanh(eta_s * (rho[i] - rho0)), tj = tanh(eta_s * (rho[j] - rho0));
        double si = 0.5 * (1 - ti), sj = 0.5 * (1 - tj), sw = si * sj;
        double dsi = -0.5 * eta_s * (1 - ti * ti), dsj = -0.5 * eta_s * (1 - tj * tj);
        int ty = t->restype[i] * 20 + t->restype[j];
        double E = 0, dEdr = 0, dEdsw = 0;
        if ((t->flags & 16) && zdep) {   /* water_part / membrane_part from gamma_ijm(0 / 1, resId1, resId2), contactTerms.py:820-831 */
            const size_t n2 = (size_t)n * n, ij = (size_t)i * n + j;
            const double *g0 = t->gamma_idx, *g1 = t->gamma_idx + 3 * n2;
            double W = -kc * (g0[ij] * th1 + th2 * (sw * g0[2 * n2 + ij] + (1 - sw) * g0[n2 + ij]));
            double M = -kc * (g1[ij] * th1 + th2 * (sw * g1[2 * n2 + ij] + (1 - sw) * g1[n2 + ij]));
            double dW = -kc * (g0[ij] * dth1 + dth2 * (sw * g0[2 * n2 + ij] + (1 - sw) * g0[n2 + ij]));
            double dM = -kc * (g1[ij] * dth1 + dth2 * (sw * g1[2 * n2 + ij] + (1 - sw) * g1[n2 + ij]));
            double aa = alpha[i] * alpha[j];
            E = (1 - aa) * W + aa * M;
            dEdr = (1 - aa) * dW + aa * dM;
            dEdsw = (1 - aa) * (-kc * th2 * (g0[2 * n2 + ij] - g0[n2 + ij])) + aa * (-kc * th2 * (g1[2 * n2 + ij] - g1[n2 + ij]));
            if (f) {
                f[3 * t->atom[i] + 2] -= (M - W) * dalpha[i] * alpha[j];
                f[3 * t->atom[j] + 2] -= (M - W) * alpha[i] * dalpha[j];
            }
        } else if (t->flags & 16) {   /* gamma_ijm(0, resId1, resId2) etc., contactTerms.py:853-860 */
            const size_t n2 = (size_t)n * n, ij = (size_t)i * n + j;
            double gd = t->gamma_idx[ij], gp = t->gamma_idx[n2 + ij], gw = t->gamma_idx[2 * n2 + ij];
            E = -kc * (gd * th1 + th2 * (sw * gw + (1 - sw) * gp));
            dEdr = -kc * (gd * dth1 + dth2 * (sw * gw + (1 - sw) * gp));
            dEdsw = -kc * th2 * (gw - gp);
        } else if (!zdep) {
            const double* g = t->gamma + t->env * 1200;
            double gd = g[ty], gp = g[400 + ty], gw = g[800 + ty];
            E = -kc * (gd * th1 + th2 * (sw * gw + (1 - sw) * gp));
            dEdr = -kc * (gd * dth1 + dth2 * (sw * gw + (1 - sw) * gp));
            dEdsw = -kc * th2 * (gw - gp);
        } else {
            const double *g0 = t->gamma, *g1 = t->gamma + 1200;
            double W = -kc * (g0[ty] * th1 + th2 * (sw * g0[800 + ty] + (1 - sw) * g0[400 + ty]));
            double M = -kc * (g1[ty] * th1 + th2 * (sw * g1[800 + ty] + (1 - sw) * g1[400 + ty]));
            double dW = -kc * (g0[ty] * dth1 + dth2 * (sw * g0[800 + ty] + (1 - sw) * g0[400 + ty]));
            double dM = -kc * (g1[ty] * dth1 + dth2 * (sw * g1[800 + ty] + (1 - sw) * g1[400 + ty]));
            double aa = alpha[i] * alpha[j];
            E = (1 - aa) * W + aa * M;
            dEdr = (1 - aa) * dW + aa * dM;
            dEdsw = (1 - aa) * (-kc * th2 * (g0[800 + ty] - g0[400 + ty])) + aa * (-kc * th2 * (g1[800 + ty] - g1[400 + ty]));
            if (f) {
                f[3 * t->atom[i] + 2] -= (M - W) * dalpha[i] * alpha[j];
                f[3 * t->atom[j] + 2] -= (M - W) * alpha[i] * dalpha[j];
            }
        }
        e += E;
        dEdrho[i] += dEdsw * dsi * sj; dEdrho[j] += dEdsw * si * dsj;
        pair_force(f, t->atom[i], t->atom[j], d, r, dEdr);
    }
    if (t->has_burial) for (int i = 0; i < n; i++) for (int m = 0; m < 3; m++) {   /* contactTerms.py:343-351 */
        double gb = (t->flags & 16) ? t->bgamma_idx[i * 3 + m] : t->bgamma[t->restype[i] * 3 + m], lo = p[10 + m], hi = p[13 + m];
        double ta = tanh(kappa * (rho[i] - lo)), tb = tanh(kappa * (hi - rho[i]));
        if (t->flags & 8) {   /* hybrid_contact_term, contactTerms.py:1283-1289: (1-alpha_i) water wells + alpha_i membrane wells */
            double gm = t->bgamma[60 + t->restype[i] * 3 + m];
            if (f) f[3 * t->atom[i] + 2] -= -0.5 * kb * (gm - gb) * (ta + tb) * dalpha[i];
            gb = (1 - alpha[i]) * gb + alpha[i] * gm;
        }
        e += -0.5 * kb * gb * (ta + tb);
        dEdrho[i] += -0.5 * kb * gb * kappa * ((1 - ta * ta) - (1 - tb * tb));
    }
    if (f) for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) {           /* chain rule through rho */
        if (abs(t->resid[i] - t->resid[j]) < 2) continue;
        double r = dist(x, t->atom[i], t->atom[j], d);
        if (r >= cutoff) continue;
        double dth; theta_sw(r, rmin, rmax, eta, &dth);
        pair_force(f, t->atom[i], t->atom[j], d, r, (dEdrho[i] + dEdrho[j]) * dth);
    }
    free(rho); free(dEdrho); free(alpha); free(dalpha);
    return e;
}
/* theta(x;y,z) of hydrogenBondTerms.py:447 with gradient coefficients wrt the two distances */
static inline double hb_theta(double r1, double r2, double r_ON, double s_NO, double r_OH, double s_HO, double* d1, double* d2) {
    double th = exp(-(r1 - r_ON) * (r1 - r_ON) / (2 * s_NO * s_NO) - (r2 - r_OH) * (r2 - r_OH) / (2 * s_HO * s_HO));
    *d1 = -th * (r1 - r_ON) / (s_NO * s_NO); *d2 = -th * (r2 - r_OH) / (s_HO * s_HO);
    return th;
}
/* hydrogenBondTerms.py:456-473 (beta_1), 522-536 (beta_2), 588-606 (beta_3) */
static double eval_beta(const beta_t* t, const double* x, double* f) {
    double e = 0, dON[3], dOH[3], d2N[3], d2H[3];
    for (int ia = 0; ia < t->nacc; ia++) for (int id = 0; id < t->ndon; id++) {
        const int* A = t->acc_atoms + 3 * ia; const int* D = t->don_atoms + 3 * id;
        double rON = dist(x, A[0], D[0], dON);
        if (rON >= t->cutoff) continue;
        double lam = beta_lambda(t, t->don_res[id], t->acc_res[ia]);
        if (lam == 0.0) continue;
        double rOH = dist(x, A[0], D[1], dOH), g1, g2, h1 = 0, h2 = 0, r2N = 0, r2H = 0;
        double th1 = hb_theta(rON, rOH, t->r_ON, t->s_NO, t->r_OH, t->s_HO, &g1, &g2), th2 = 1.0;
        if (t->two_theta) {
            r2N = dist(x, D[2], A[1], d2N); r2H = dist(x, D[2], A[2], d2H);
            th2 = hb_theta(r2N, r2H, t->r_ON, t->s_NO, t->r_OH, t->s_HO, &h1, &h2);
        }
        double c = -t->k * lam;
        e += c * th1 * th2;
        if (f) {
            pair_force(f, A[0], D[0], dON, rON, c * g1 * th2);
            pair_force(f, A[0], D[1], dOH, rOH, c * g2 * th2);
            if (t->two_theta) {
                pair_force(f, D[2], A[1], d2N, r2N, c * th1 * h1);
                pair_force(f, D[2], A[2], d2H, r2H, c * th1 * h2);
            }
        }
    }
    return e;
}
/* hydrogenBondTerms.py:660-690 (pap_1), 739-758 (pap_2) */
static double eval_pap(const pap_t* t, const double* x, double* f) {
    double e = 0, d1[3], d2[3], d3[3];
    for (int ia = 0; ia < t->nacc; ia++) for (int id = 0; id < t->ndon; id++) {
        const int* A = t->acc_atoms + 2 * ia; const int* D = t->don_atoms + 2 * id;
        double r1 = dist(x, A[0], D[0], d1);
        if (r1 >= t->cutoff) continue;
        double gam = pap_gamma(t, t->don_res[id], t->acc_res[ia]);
        if (gam == 0.0) continue;
        double r2 = dist(x, A[1], D[1], d2), r3 = dist(x, A[0], A[1], d3);
        double t1 = tanh(t->eta * (t->r0 - r1)), t2 = tanh(t->eta * (t->r0 - r2)), t3 = tanh(t->eta * (r3 - t->d_i4));
        double a1 = 0.5 * (1 + t1), a2 = 0.5 * (1 + t2), a3 = 0.5 * (1 + t3);
        double c = -t->k * gam;
        e += c * a1 * a2 * a3;
        if (f) {
            pair_force(f, A[0], D[0], d1, r1, c * (-0.5 * t->eta * (1 - t1 * t1)) * a2 * a3);
            pair_force(f, A[1], D[1], d2, r2, c * a1 * (-0.5 * t->eta * (1 - t2 * t2)) * a3);
            pair_force(f, A[0], A[1], d3, r3, c * a1 * a2 * (0.5 * t->eta * (1 - t3 * t3)));
        }
    }
    return e;
}
/* hydrogenBondTerms.py:818-829; z_dependent_helical_term 869-873: the coefficient (fa_i + fa_ip4)(1 - alpha) +
 * (fa_i_membrane + fa_ip4_membrane) alpha with alpha = 1/2 tanh(eta (z + z_m)) + 1/2 tanh(eta (z_m - z)), z of CA_i */
static double eval_helical(const helical_t* t, const double* x, double* f) {
    double e = 0, d1[3], d2[3];
    for (int b = 0; b < t->n; b++) {
        const int* a = t->atoms + 3 * b;
        double r1 = dist(x, a[0], a[1], d1), r2 = dist(x, a[0], a[2], d2), g1, g2;
        double th = hb_theta(r1, r2, t->r_ON, t->s_NO, t->r_OH, t->s_HO, &g1, &g2);
        double coef = t->coef[b];
        if (t->ca) {
            const double z = x[3 * t->ca[b] + 2] - t->center;
            const double tu = tanh(t->eta * (z + t->z_m)), tv = tanh(t->eta * (t->z_m - z));
            const double alpha = 0.5 * tu + 0.5 * tv, dalpha = 0.5 * t->eta * ((1 - tu * tu) - (1 - tv * tv));
            coef = t->coef[b] * (1 - alpha) + t->coef_m[b] * alpha;
            if (f) f[3 * t->ca[b] + 2] += th * (t->coef_m[b] - t->coef[b]) * dalpha;   /* -dE/dz, E = -coef theta */
        }
        e += -coef * th;
        pair_force(f, a[0], a[1], d1, r1, -coef * g1);
        pair_force(f, a[0], a[2], d2, r2, -coef * g2);
    }
    return e;
}
/* templateTerms.py:273-281 */
static double eval_fm(const fm_t* t, const double* x, double* f) {
    double e = 0, d[3];
    for (int b = 0; b < t->n; b++) {
        int i = t->pairs[2 * b], j = t->pairs[2 * b + 1];
        double r = dist(x, i, j, d);
        double fi = floor(r / t->dr);
        int i1 = fi > t->ntab - 2 ? t->ntab - 2 : (int)fi;
        double v1 = t->table[(size_t)b * t->ntab + i1], v2 = t->table[(size_t)b * t->ntab + i1 + 1];
        double r1 = t->dr * i1, r2 = t->dr * (i1 + 1);
        e += -t->k * ((v2 - v1) * r + v1 * r2 - v2 * r1) / (r2 - r1);
        pair_force(f, i, j, d, r, -t->k * (v2 - v1) / (r2 - r1));
    }
    return e;
}
/* membraneTerms.py:43-52, 144-153 */
static double eval_memb(const memb_t* t, const double* x, double* f) {
    double e = 0;
    for (int b = 0; b < t->n; b++) {
        int a = t->atoms[b];
        if (t->kind == 4) {            /* membraneTerms.py:202: k |z - center - z_m,i| */
            double u = x[3 * a + 2] - t->center - t->scale[b];
            e += t->k * fabs(u);
            if (f) f[3 * a + 2] -= t->k * (u > 0 ? 1.0 : (u < 0 ? -1.0 : 0.0));
            continue;
        }
        if (t->kind != 0) {            /* biasTerms.py:595: (-k) * x */
            int ax = t->kind - 1;
            e += -t->k * t->scale[b] * x[3 * a + ax];
            if (f) f[3 * a + ax] += t->k * t->scale[b];
            continue;
        }
        double z = x[3 * a + 2] - t->center;
        double u = t->k_m * (z + t->z_m), v = t->k_m * (t->z_m - z);
        e += t->k * t->scale[b] * (0.5 * tanh(u) + 0.5 * tanh(v));
        if (f) f[3 * a + 2] -= t->k * t->scale[b] * 0.5 * t->k_m * (sech2(u) - sech2(v));
    }
    return e;
}
/* debyeHuckelTerms.py:33 */
static double eval_dh(const dh_t* t, const double* x, double* f) {
    double e = 0, d[3];
    for (int b = 0; b < t->n; b++) {
        int i = t->pairs[2 * b], j = t->pairs[2 * b + 1];
        double r = dist(x, i, j, d);
        double E = t->k * t->qq[b] / r * exp(-r / t->lam);
        e += E;
        pair_force(f, i, j, d, r, -E * (1.0 / r + 1.0 / t->lam));
    }
    return e;
}

/* er_term templateTerms.py:643-693 and additive_amhgo_term templateTerms.py:600-640 (CustomBondForce):
 * E = -k gamma_ij exp(-(r - r_ijN)^2 / (2 sigma_ij^2)) */
static double bond_value(const gauss_t* t, int b, double r, double* dgdr) {
    if (t->kind != 0) { *dgdr = 2 * t->kg[b] * r; return t->kg[b] * r * r; }   /* kind 2: r is the length of the xy projection */
    double dr = r - t->r0[b], s2 = t->sigma[b] * t->sigma[b];
    double g = -t->kg[b] * exp(-dr * dr / (2 * s2));
    *dgdr = -g * dr / s2;
    return g;
}
/* separation of a bond; for the cylindrical kind (biasTerms.py:560) its projection on the xy plane */
static double bond_dist(const gauss_t* t, const double* x, int i, int j, double* d) {
    double r = dist(x, i, j, d);
    if (t->kind == 2) { d[2] = 0; r = sqrt(d[0] * d[0] + d[1] * d[1]); }
    return r;
}
static double eval_gauss(const gauss_t* t, const double* x, double* f) {
    double S = 0, d[3], dg;
    for (int b = 0; b < t->n; b++) S += bond_value(t, b, bond_dist(t, x, t->pairs[2 * b], t->pairs[2 * b + 1], d), &dg);
    double E = S, fp = 1;   /* E = f(S), fp = f'(S) */
    if (t->cv == 1) { E = 0.5 * t->cv_k * (S - t->cv_x0) * (S - t->cv_x0); fp = t->cv_k * (S - t->cv_x0); }
    else if (t->cv == 2 || t->cv == 3) {
        double q = S > 0 ? sqrt(S) : 0, iq = q > 0 ? 1 / q : 0;
        if (t->cv == 2) { E = t->cv_k * q; fp = 0.5 * t->cv_k * iq; }
        else { E = t->cv_k * (q - t->cv_x0) * (q - t->cv_x0); fp = t->cv_k * (q - t->cv_x0) * iq; }
    }
    for (int b = 0; b < t->n; b++) {
        int i = t->pairs[2 * b], j = t->pairs[2 * b + 1];
        double r = bond_dist(t, x, i, j, d);
        bond_value(t, b, r, &dg);
        if (r > 0) pair_force(f, i, j, d, r, fp * dg);
    }
    return E;
}

/* energies per group (eg[32], may be NULL) and forces f (natoms*3, may be NULL; accumulated into) */
static void evaluate(const ctx_t* c, const double* x, uint32_t mask, double* eg, double* f) {
    for (int i = 0; i < c->nterms; i++) {
        const term_t* t = &c->terms[i];
        int group = *(const int*)t->p;   /* first member of every term struct */
        if (!((mask >> group) & 1u)) continue;
        double e = 0;
        switch (t->kind) {
        case T_HARM: e = eval_harm(t->p, x, f); break;
        case T_CHI: e = eval_chi(t->p, x, f); break;
        case T_RAMA: e = eval_rama(t->p, x, f); break;
        case T_EXCL: e = eval_excl(t->p, x, f); break;
        case T_CONTACT: e = eval_contact(t->p, x, f); break;
        case T_BETA: e = eval_beta(t->p, x, f); break;
        case T_PAP: e = eval_pap(t->p, x, f); break;
        case T_HELICAL: e = eval_helical(t->p, x, f); break;
        case T_FM: e = eval_fm(t->p, x, f); break;
        case T_MEMB: e = eval_memb(t->p, x, f); break;
        case T_DH: e = eval_dh(t->p, x, f); break;
        case T_GAUSS: e = eval_gauss(t->p, x, f); break;
        }
        if (eg) eg[group] += e;
    }
}
/* fold forces on virtual sites onto their parents (OpenMM ThreeParticleAverageSite force distribution) */
static void redistribute(const ctx_t* c, double* f) {
    for (int i = 0; i < c->nvs; i++) {
        int s = c->vs_site[i]; const int* p = c->vs_par + 3 * i; const double* w = c->vs_w + 3 * i;
        for (int k = 0; k < 3; k++) {
            double fs = f[3 * s + k];
            f[3 * p[0] + k] += w[0] * fs; f[3 * p[1] + k] += w[1] * fs; f[3 * p[2] + k] += w[2] * fs;
            f[3 * s + k] = 0;
        }
    }
}

int awsem_oracle_energy_by_group(ctx_t* c, double* e) {
    CHECK_RUN(c);
    if (!e) return fail(-1, "energy_by_group: null output");
    memset(e, 0, (size_t)c->nrep * NGROUPS * sizeof(double));
#pragma omp parallel for num_threads(c->nthreads) schedule(dynamic)
    for (int r = 0; r < c->nrep; r++) evaluate(c, c->pos + (size_t)r * c->natoms * 3, 0xffffffffu, e + (size_t)r * NGROUPS, NULL);
    return 0;
}
int awsem_oracle_energy(ctx_t* c, uint32_t mask, double* e) {
    CHECK_RUN(c);
    if (!e) return fail(-1, "energy: null output");
#pragma omp parallel for num_threads(c->nthreads) schedule(dynamic)
    for (int r = 0; r < c->nrep; r++) {
        double eg[NGROUPS] = {0};
        evaluate(c, c->pos + (size_t)r * c->natoms * 3, mask, eg, NULL);
        double s = 0; for (int g = 0; g < NGROUPS; g++) s += eg[g];
        e[r] = s;
    }
    return 0;
}
int awsem_oracle_forces(ctx_t* c, uint32_t mask, int redist, double* f) {
    CHECK_RUN(c);
    if (!f) return fail(-1, "forces: null output");
    size_t n = (size_t)c->natoms * 3;
    memset(f, 0, c->nrep * n * sizeof(double));
#pragma omp parallel for num_threads(c->nthreads) schedule(dynamic)
    for (int r = 0; r < c->nrep; r++) {
        evaluate(c, c->pos + r * n, mask, NULL, f + r * n);
        if (redist) redistribute(c, f + r * n);
    }
    return 0;
}

/* ------------------------------------------------------------------ Philox4x32-10 + Gaussian stream
 * Counter-based RNG shared bit-for-bit with the HIP engine so the two paths can be compared step-locked:
 * counter = (atom, step_lo, step_hi, replica), key = (seed_lo, seed_hi ^ stream); the four 32-bit outputs give
 * uniforms u = ((x >> 8) + 0.5) / 2^24 and two Box-Muller pairs; the first three normals are used (x, y, z). */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int i = 0; i < 10; i++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
static void gauss3(uint64_t seed, uint32_t stream, int rep, int64_t step, int atom, double g[3]) {
    uint32_t c[4] = {(uint32_t)atom, (uint32_t)(uint64_t)step, (uint32_t)((uint64_t)step >> 32), (uint32_t)rep};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ stream);
    double u0 = ((c[0] >> 8) + 0.5) / 16777216.0, u1 = ((c[1] >> 8) + 0.5) / 16777216.0;
    double u2 = ((c[2] >> 8) + 0.5) / 16777216.0, u3 = ((c[3] >> 8) + 0.5) / 16777216.0;
    double r0 = sqrt(-2.0 * log(u0)), r1 = sqrt(-2.0 * log(u2));
    g[0] = r0 * cos(2 * M_PI * u1); g[1] = r0 * sin(2 * M_PI * u1); g[2] = r1 * cos(2 * M_PI * u3);
}
void awsem_oracle_gauss3(uint64_t seed, uint32_t stream, int rep, int64_t step, int atom, double* g) { gauss3(seed, stream, rep, step, atom, g); }

/* context.setVelocitiesToTemperature (scripts/mm_run.py:151): v = sqrt(kT/m) N(0,1) for massive particles. */
int awsem_oracle_set_velocities_to_temperature(ctx_t* c, const double* T, uint64_t seed) {
    CHECK_RUN(c);
    if (!T) return fail(-1, "set_velocities_to_temperature: null temperatures");
    for (int r = 0; r < c->nrep; r++) for (int a = 0; a < c->natoms; a++) {
        double* v = c->vel + ((size_t)r * c->natoms + a) * 3, g[3];
        if (c->mass[a] == 0) { v[0] = v[1] = v[2] = 0; continue; }
        gauss3(seed, 0x56454c30u /* "VEL0" */, r, 0, a, g);
        double s = sqrt(KB * T[r] / c->mass[a]);
        v[0] = s * g[0]; v[1] = s * g[1]; v[2] = s * g[2];
    }
    return 0;
}
int awsem_oracle_scale_velocities(ctx_t* c, const double* scale) {
    CHECK_RUN(c);
    if (!scale) return fail(-1, "scale_velocities: null factors");
    for (int r = 0; r < c->nrep; r++) for (int i = 0; i < c->natoms * 3; i++) c->vel[(size_t)r * c->natoms * 3 + i] *= scale[r];
    return 0;
}
/* OpenMM LangevinIntegrator (ReferenceStochasticDynamics), as configured at scripts/mm_run.py:145:
 *   v <- a v + (1-a)/gamma F/m + sqrt(kT(1-a^2)/m) N(0,1),  a = exp(-gamma dt);   x <- x + v dt; virtual sites re-derived.
 * No constraints exist in the AWSEM system, so OpenMM's post-constraint v = (x'-x)/dt is the identity. */
int awsem_oracle_step(ctx_t* c, int nsteps, const double* T, double dt, double friction, uint64_t seed) {
    CHECK_RUN(c);
    if (nsteps < 0 || !T || !(dt > 0) || friction < 0) return fail(-1, "step: bad arguments");
    const int n = c->natoms;
    const double a = exp(-friction * dt), fs = friction == 0 ? dt : (1 - a) / friction;
    int64_t step0 = c->step;
#pragma omp parallel for num_threads(c->nthreads) schedule(dynamic)
    for (int r = 0; r < c->nrep; r++) {
        double* x = c->pos + (size_t)r * n * 3; double* v = c->vel + (size_t)r * n * 3;
        double* f = malloc((size_t)n * 3 * sizeof(double));
        const double ns = sqrt(KB * T[r] * (1 - a * a));
        for (int s = 0; s < nsteps; s++) {
            if (c->remove_cm) {   /* OpenMM ReferenceRemoveCMMotionKernel, run by updateContextState at the top of the step:
                                     v_i -= (sum m v) / (sum m) for every particle with a mass */
                double p[3] = {0, 0, 0}, mt = 0;
                for (int i = 0; i < n; i++) { mt += c->mass[i]; for (int k = 0; k < 3; k++) p[k] += c->mass[i] * v[3 * i + k]; }
                for (int i = 0; i < n; i++) if (c->mass[i] != 0) for (int k = 0; k < 3; k++) v[3 * i + k] -= p[k] / mt;
            }
            virtual_sites(c, x);
            memset(f, 0, (size_t)n * 3 * sizeof(double));
            evaluate(c, x, 0xffffffffu, NULL, f);
            redistribute(c, f);
            for (int i = 0; i < n; i++) {
                if (c->mass[i] == 0) continue;
                double g[3], im = 1.0 / c->mass[i], sq = sqrt(im);
                gauss3(seed, 0x4c414e47u /* "LANG" */, r, step0 + s, i, g);
                for (int k = 0; k < 3; k++) {
                    v[3 * i + k] = a * v[3 * i + k] + fs * im * f[3 * i + k] + ns * sq * g[k];
                    x[3 * i + k] += dt * v[3 * i + k];
                }
            }
        }
        virtual_sites(c, x);
        free(f);
    }
    c->step += nsteps;
    return 0;
}
/* simulation.minimizeEnergy() (scripts/mm_run.py:136,153).  OpenMM's LocalEnergyMinimizer is L-BFGS; nothing in the
 * reference pins the minimiser beyond "a local minimum within the force tolerance", so this is the engine's own
 * algorithm restated in fp64 (parity unpinned): FIRE (Bitzek et al. 2006) on the massive particles with unit masses,
 *   P = F.v;  v <- (1-alpha) v + alpha |v| F/|F|;  P >= 0 for more than 5 steps: dt <- min(1.1 dt, dt_max), alpha *= 0.99;
 *   P < 0: v <- 0, dt <- dt/2, alpha <- 0.1;   v <- v + dt F;  x <- x + clamp(dt v, |.| <= 0.02 nm);  virtual sites re-derived,
 * until the RMS force over the massive degrees of freedom is <= tolerance (kJ/mol/nm) or max_iterations is reached. */
int awsem_oracle_minimize(ctx_t* c, double tolerance, int max_iterations, int* iterations, double* rms_force) {
    CHECK_RUN(c);
    if (!(tolerance > 0) || max_iterations < 0) return fail(-1, "minimize: bad arguments");
    if (max_iterations == 0) max_iterations = 100000;
    const int n = c->natoms;
    int nm = 0;
    for (int i = 0; i < n; i++) if (c->mass[i] != 0) nm++;
#pragma omp parallel for num_threads(c->nthreads) schedule(dynamic)
    for (int r = 0; r < c->nrep; r++) {
        double* x = c->pos + (size_t)r * n * 3;
        double* f = malloc((size_t)n * 3 * sizeof(double));
        double* v = calloc((size_t)n * 3, sizeof(double));
        double dt = 0.001, alpha = 0.1, rms = 0;
        const double dt_max = 0.01, dmax = 0.02;
        int npos = 0, it;
        for (it = 0; it < max_iterations; it++) {
            virtual_sites(c, x);
            memset(f, 0, (size_t)n * 3 * sizeof(double));
            evaluate(c, x, 0xffffffffu, NULL, f);
            redistribute(c, f);
            double P = 0, FF = 0, VV = 0;
            for (int i = 0; i < n; i++) {
                if (c->mass[i] == 0) continue;
                for (int k = 0; k < 3; k++) { P += f[3 * i + k] * v[3 * i + k]; FF += f[3 * i + k] * f[3 * i + k]; VV += v[3 * i + k] * v[3 * i + k]; }
            }
            rms = sqrt(FF / (3.0 * (nm > 0 ? nm : 1)));
            if (rms <= tolerance) break;
            double keep = 0, mix = 0;
            if (P >= 0) {   /* not uphill (the first iteration has v = 0) */
                keep = 1.0 - alpha; mix = alpha * sqrt(VV / FF);
                if (++npos > 5) { dt = dt * 1.1 < dt_max ? dt * 1.1 : dt_max; alpha *= 0.99; }
            } else { npos = 0; dt *= 0.5; alpha = 0.1; }
            for (int i = 0; i < n; i++) {
                if (c->mass[i] == 0) continue;
                double d[3], d2 = 0;
                for (int k = 0; k < 3; k++) {
                    const double vn = keep * v[3 * i + k] + mix * f[3 * i + k] + dt * f[3 * i + k];
                    v[3 * i + k] = vn; d[k] = dt * vn; d2 += d[k] * d[k];
                }
                const double sc = d2 > dmax * dmax ? dmax / sqrt(d2) : 1.0;
                for (int k = 0; k < 3; k++) x[3 * i + k] += sc * d[k];
            }
        }
        virtual_sites(c, x);
        if (iterations) iterations[r] = it;
        if (rms_force) rms_force[r] = rms;
        free(f); free(v);
    }
    return 0;
}
int64_t awsem_oracle_step_count(ctx_t* c) { return c ? c->step : -1; }
int awsem_oracle_set_step_count(ctx_t* c, int64_t s) { if (!c) return fail(-1, "null handle"); c->step = s; return 0; }

/* ------------------------------------------------------------------ brute-force pair sets (include/awsem_hip.h: awsem_neighbors)
 * O(N^2) scan with the distance screen evaluated exactly as the engine's list builder does it: coordinates rounded
 * to the handle's precision, r^2 = fma(dz,dz,fma(dy,dy,dx*dx)), compared with (real)((cutoff+skin)^2). */
static int below(const ctx_t* c, const double* x, int a, int b, double rc) {
    if (c->precision == 0) {
        float dx = (float)x[3 * a] - (float)x[3 * b], dy = (float)x[3 * a + 1] - (float)x[3 * b + 1], dz = (float)x[3 * a + 2] - (float)x[3 * b + 2];
        return fmaf(dz, dz, fmaf(dy, dy, dx * dx)) < (float)(rc * rc);
    }
    double dx = x[3 * a] - x[3 * b], dy = x[3 * a + 1] - x[3 * b + 1], dz = x[3 * a + 2] - x[3 * b + 2];
    return fma(dz, dz, fma(dy, dy, dx * dx)) < rc * rc;
}
static int cmp_pair(const void* a, const void* b) {
    const int *p = a, *q = b;
    return p[0] != q[0] ? (p[0] < q[0] ? -1 : 1) : (p[1] < q[1] ? -1 : (p[1] > q[1]));
}
int awsem_oracle_neighbors(ctx_t* c, int rep, int kind, int index, double skin, int* pairs, int capacity, int* n_pairs) {
    CHECK_RUN(c);
    if (rep < 0 || rep >= c->nrep || capacity < 0 || (capacity && !pairs) || !n_pairs || !(skin >= 0)) return fail(-1, "neighbors: bad arguments");
    static const int want[4] = {T_EXCL, T_CONTACT, T_BETA, T_PAP};
    if (kind < 0 || kind > 3) return fail(-1, "neighbors: bad kind");
    void* term = NULL;
    for (int t = 0, k = 0; t < c->nterms; t++) if (c->terms[t].kind == want[kind]) { if (k++ == index) { term = c->terms[t].p; break; } }
    if (!term) return fail(-1, "neighbors: no such term");
    const double* x = c->pos + (size_t)rep * c->natoms * 3;
    int n = 0, cap = 1024;
    int* out = malloc(sizeof(int) * 2 * cap);
    /* the scans run on c->nthreads threads (set_threads); hits are rare next to tests, so they are appended under a lock and the
     * result is sorted afterwards */
#define PUSH(p, q) do { _Pragma("omp critical(awsem_push)") { if (n == cap) { cap *= 2; out = realloc(out, sizeof(int) * 2 * cap); } out[2 * n] = (p); out[2 * n + 1] = (q); n++; } } while (0)
#define SCAN _Pragma("omp parallel for schedule(dynamic, 8) num_threads(c->nthreads)")
    if (kind == 0) {
        const excl_t* t = term;
        SCAN
        for (int i = 0; i < t->nsites; i++) for (int j = i + 1; j < t->nsites; j++) {
            if (t->cls[i] != t->cls[j]) continue;
            int a = t->atom[i], b = t->atom[j];
            if (a == b || !below(c, x, a, b, t->r0 + skin)) continue;
            if (!is_excluded(t->keys, t->nexcl, a, b)) PUSH(a < b ? a : b, a < b ? b : a);
        }
    } else if (kind == 1) {
        const contact_t* t = term;
        const double cutoff = (t->flags & 4) ? 1e15 : t->p[8];
        SCAN
        for (int i = 0; i < t->nsites; i++) for (int j = i + 1; j < t->nsites; j++)
        {
            int dres = abs(t->resid[i] - t->resid[j]);
            int pair_e = !(t->flags & 2) && (t->chain[i] != t->chain[j] || dres >= t->min_seq_sep);
            if ((dres >= 2 || pair_e) && below(c, x, t->atom[i], t->atom[j], cutoff + skin)) PUSH(i, j);
        }
    } else if (kind == 2) {
        const beta_t* t = term;
        double base = t->cutoff;
        if (c->precision == 0) {
            /* the engine's fp32 screening exponent (awsem_hip.hip, beta block): X with |c|max e^-X sqrt(2X) / sigma_min = 2^-16 over all
             * beta terms, 30 when a term carries the functional form for more than 4096 residues */
            double X = 30.0, cmax = 0, smin = 1e300;
            int dense = 1;
            for (int u = 0; u < c->nterms; u++) if (c->terms[u].kind == T_BETA) {
                const beta_t* b = c->terms[u].p;
                if (b->fn && b->nres > 4096) { dense = 0; break; }
                double m = 0;
                for (int q = 0; q < b->nres; q++) for (int p = 0; p < b->nres; p++) { double v = fabs(beta_lambda(b, q, p)); if (v > m) m = v; }
                if (fabs(b->k) * m > cmax) cmax = fabs(b->k) * m;
                if (b->s_NO < smin) smin = b->s_NO;
                if (b->s_HO < smin) smin = b->s_HO;
            }
            if (dense && cmax > 0 && !(getenv("AWSEM_BETA_SCREEN") && atoi(getenv("AWSEM_BETA_SCREEN")) == 30)) {
                X = 20.0;
                for (int it = 0; it < 8; it++) X = log(cmax * 65536.0 * sqrt(2.0 * X) / smin);
                X = X > 30.0 ? 30.0 : (X < 8.0 ? 8.0 : X);
            }
            double rs = t->r_ON + sqrt(2.0 * X) * t->s_NO, s2 = t->cutoff * t->cutoff < rs * rs ? t->cutoff * t->cutoff : rs * rs;
            base = sqrt((double)(float)s2);
        } else base = sqrt(t->cutoff * t->cutoff);
        SCAN
        for (int a = 0; a < t->nacc; a++) for (int q = 0; q < t->ndon; q++)
            if (beta_lambda(t, t->don_res[q], t->acc_res[a]) != 0.0 && below(c, x, t->acc_atoms[3 * a], t->don_atoms[3 * q], base + skin)) PUSH(a, q);
    } else {
        const pap_t* t = term;
        double base = t->cutoff;
        if (c->precision == 0) {
            /* the engine's fp32 screen (awsem_hip.hip, pap block): switch arguments below A = log(2^-16 / (|c|max eta2)) give forces that
             * round to zero, so the list closes where the first switch reaches A */
            double A = -30.0, cmax = 0;
            int dense = 1;
            for (int u = 0; u < c->nterms; u++) if (c->terms[u].kind == T_PAP) {
                const pap_t* b = c->terms[u].p;
                if (b->fn && b->nres > 4096) { dense = 0; break; }
                double m = 0;
                for (int q = 0; q < b->nres; q++) for (int p = 0; p < b->nres; p++) { double v = fabs(pap_gamma(b, q, p)); if (v > m) m = v; }
                if (fabs(b->k) * m * 2.0 * b->eta > cmax) cmax = fabs(b->k) * m * 2.0 * b->eta;
            }
            if (dense && cmax > 0 && !(getenv("AWSEM_PAP_SCREEN") && atoi(getenv("AWSEM_PAP_SCREEN")) == 30)) {
                A = log(1.0 / (65536.0 * cmax));
                A = A < -30.0 ? -30.0 : (A > -8.0 ? -8.0 : A);
            }
            double rs = t->r0 - A / (2.0 * t->eta), s2 = t->cutoff * t->cutoff < rs * rs ? t->cutoff * t->cutoff : rs * rs;
            base = sqrt((double)(float)s2);
        }
        SCAN
        for (int a = 0; a < t->nacc; a++) for (int q = 0; q < t->ndon; q++)
            if (pap_gamma(t, t->don_res[q], t->acc_res[a]) != 0.0 && below(c, x, t->acc_atoms[2 * a], t->don_atoms[2 * q], base + skin)) PUSH(a, q);
    }
#undef PUSH
#undef SCAN
    qsort(out, n, 2 * sizeof(int), cmp_pair);
    for (int k = 0; k < n && k < capacity; k++) { pairs[2 * k] = out[2 * k]; pairs[2 * k + 1] = out[2 * k + 1]; }
    free(out);
    *n_pairs = n;
    return 0;
}
int awsem_oracle_last_kernel_ms(ctx_t* c, double* ms) { (void)c; if (ms) *ms = 0; return 0; }
int awsem_oracle_synchronize(ctx_t* c) { (void)c; return 0; }
