// awsem_hip.hip -- host side of libawsem_hip.so: the C ABI of include/awsem_hip.h on top of the gfx950 kernels in
// awsem_kernels.h.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (see __graft_entry__.build()).
//
// A handle collects term descriptors on the host (add_* calls), converts and uploads them once (finalize), and
// then owns: one DevSys<real> (shared, read-only), SoA replica state pos/vel [R][3][natoms], and a HIP stream.
// Precision is a template parameter of the whole engine: Engine<float> is the production path, Engine<double> the
// check mode running the identical kernels in fp64.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include <atomic>
#include <mutex>
#include <limits>
#include <map>

#include "../../include/awsem_hip.h"
#include "awsem_kernels.h"
#include "awsem_cell.h"

// ---- awsem_energy_by_group_noalloc: the one piece of device memory that is not behind a handle -----------------------
// Pair lists and status word of the allocation-free entry point.  In host code the NAME of a __device__ array is only a
// stand-in for the symbol: its device address is looked up once per device (hipGetSymbolAddress, with that device
// current -- every GPU holds its own copy at its own address) when a handle is finalized there, kept in
// g_noalloc_ptr[device], and only that pointer is ever handed to a kernel or a memset.
static __device__ __align__(256) unsigned char scratch[AWSEM_NOALLOC_SCRATCH_BYTES];
#define AWSEM_MAX_DEVICES 64
static std::atomic<void*> g_noalloc_ptr[AWSEM_MAX_DEVICES] = {};   // two threads finalizing handles on one GPU look up the same address: either store is right

namespace {

thread_local char g_err[1024] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

struct HipError { hipError_t e; const char* what; };
struct StateError { const char* what; };
#define HIP_TRY(call)                                               \
    do {                                                            \
        hipError_t _e = (call);                                     \
        if (_e != hipSuccess) throw HipError{_e, #call};            \
    } while (0)

// ---- host-side (precision independent) term records -------------------------------------------------------
struct HHarm { int group; std::vector<int> pairs; std::vector<double> r0, k; };
struct HChi { int group; std::vector<int> atoms; double k, chi0; };
struct HRama { int group, nwell; std::vector<int> atoms; std::vector<double> wp, bw; double k; };
struct HExcl { int group; std::vector<int> atom, cls, excl; double k, r0; };
struct HContact {
    int group, env, min_seq_sep, flags; bool has_burial;
    std::vector<int> atom, restype, resid, chain;
    double p[AWSEM_CP_COUNT];
    std::vector<double> gamma, bgamma;
};
// functional form of a lambda / gamma table (awsem_add_hbond_beta_fn / _pap_fn): kind 0 = none (dense table)
struct HFn { int kind = 0; std::vector<int> info; std::vector<double> coef; };
struct HBeta { int group, two_theta, nres; std::vector<int> acc_atoms, acc_res, don_atoms, don_res; std::vector<double> lambda; double k, r_ON, s_NO, r_OH, s_HO, cutoff; HFn fn; };
struct HPap { int group, nres; std::vector<int> acc_atoms, acc_res, don_atoms, don_res; std::vector<double> gamma; double k, eta, r0, d_i4, cutoff; HFn fn; };
struct HHelical { int group; std::vector<int> atoms; std::vector<double> coef; double r_ON, s_NO, r_OH, s_HO;   // atoms: (O, N, H, CA or -1) per bond
                  std::vector<double> coef_m; double eta = 0, z_m = 0, center = 0; };   // z-dependent variant: coef_m non-empty
struct HFm { int group, ntab; std::vector<int> pairs; std::vector<double> table; double k, dr; };
struct HMemb { int group; std::vector<int> atoms; std::vector<double> scale; double k, k_m, z_m, center; int kind = 0; };
struct HDh { int group; std::vector<int> pairs; std::vector<double> qq; double k, lam; };
struct HGauss { int group; std::vector<int> pairs; std::vector<double> r0, sigma, kg; int kind = 0, cv = 0; double cv_k = 0, cv_x0 = 0; };

struct HostSystem {
    int natoms = 0;
    bool remove_cm = false;
    std::vector<double> mass;
    std::vector<int> vs_site, vs_par;
    std::vector<double> vs_w;
    std::vector<HHarm> harm; std::vector<HChi> chi; std::vector<HRama> rama; std::vector<HExcl> excl;
    std::vector<HContact> contact; std::vector<HBeta> beta; std::vector<HPap> pap; std::vector<HHelical> helical;
    std::vector<HFm> fm; std::vector<HMemb> memb; std::vector<HDh> dh; std::vector<HGauss> gauss;
};

// ---- precision-specific engine ---------------------------------------------------------------------------
struct EngineBase {
    virtual ~EngineBase() {}
    virtual void finalize(const HostSystem& H) = 0;
    virtual void set_positions(int rep, const double* xyz) = 0;
    virtual void get_positions(int rep, double* xyz) = 0;
    virtual void set_velocities(int rep, const double* v) = 0;
    virtual void get_velocities(int rep, double* v) = 0;
    virtual void compute_virtual_sites() = 0;
    virtual void maxwell(const double* T, uint64_t seed) = 0;
    virtual void scale_velocities(const double* scale) = 0;
    virtual void energy_by_group(unsigned mask, double* e) = 0;
    virtual void forces(unsigned mask, int redistribute, double* f) = 0;
    virtual void step(int nsteps, const double* T, double dt, double friction, uint64_t seed, int64_t step0) = 0;
    virtual void step_stages(int nstages, int steps_per_stage, const double* T, double dt, double friction, uint64_t seed, int64_t step0) = 0;
    virtual void synchronize() = 0;
    virtual void kinetic_energy(double* ke) = 0;
    virtual void set_state_all(const double* xyz, const double* v) = 0;
    virtual void get_state_all(double* xyz, double* v) = 0;
    virtual void snapshot_begin(int slot, unsigned mask, int energy_replica, int replica, int with_state) = 0;
    virtual void snapshot_end(int slot, double* e, double* ke, double* xyz_replica, double* xyz_all, double* v_all) = 0;
    virtual void minimize(double tolerance, int max_iterations, int32_t* iterations, double* rms) = 0;
    virtual int neighbors(int rep, int kind, int index, double skin, int32_t* pairs, int capacity) = 0;
    virtual void energy_noalloc(int nrep, const double* d_xyz, double* d_energy, hipStream_t stream) = 0;
    virtual void remd_setup(int n_total, const double* ladder, const int32_t* where, int local_start, uint64_t seed) = 0;
    virtual void step_remd(int nsteps, double dt, double friction, uint64_t seed, int64_t step0) = 0;
    virtual void remd_energies(unsigned mask, double* d_energy) = 0;
    virtual void remd_sweep(const double* d_all, int64_t iteration) = 0;
    virtual void remd_set_mixing(int scheme) = 0;
    virtual void remd_state(int32_t* slot, int64_t* attempted, int64_t* accepted, double* T_local) = 0;
    double last_ms = 0.0;
};

template <class real> class Engine : public EngineBase {
public:
    Engine(int natoms, int nrep, int device) : natoms_(natoms), nrep_(nrep), device_(device) {
        HIP_TRY(hipSetDevice(device_));
        HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&ev0_));
        HIP_TRY(hipEventCreate(&ev1_));
    }
    ~Engine() override {
        (void)hipSetDevice(device_);
        if (getenv("AWSEM_PROF_DUMP") && dsys_.prof) {   // phase cycles of -DAWSEM_PROF experiment builds
            long long p[16] = {0};
            (void)hipDeviceSynchronize();
            if (hipMemcpy(p, dsys_.prof, sizeof p, hipMemcpyDeviceToHost) == hipSuccess) {
                // (the rebuild is split by builder: build_lists_prof in awsem_kernels.h)
                static const char* names[10] = {"vsites", "rebuild: contact", "rebuild: pap + rest", "rebuild: excl", "density+fm", "single+harm+rama",
                                                "pairs+beta", "chain+chi+dh+excl+pap", "rebuild: beta", "integrate"};
                long long tot = 0;
                for (int k = 0; k < 10; k++) tot += p[k];
                if (tot > 0) for (int k = 0; k < 10; k++) fprintf(stderr, "[awsem prof] %-24s %14lld cycles  %5.1f %%\n", names[k], p[k], 100.0 * p[k] / tot);
                fprintf(stderr, "[awsem prof] list rebuilds: %lld (all replicas, all launches)\n", p[10]);
                long long wv[16 * 20] = {0};   // per wave: own work and barrier wait of every phase
                if (tot > 0 && hipMemcpy(wv, dsys_.prof + 16 + 3 * (size_t)nrep_, sizeof wv, hipMemcpyDeviceToHost) == hipSuccess)
                    for (int w = 0; w < 16; w++) {
                        long long any = 0; for (int k = 0; k < 20; k++) any += wv[w * 20 + k];
                        if (!any) continue;
                        fprintf(stderr, "[awsem prof]   wave %2d work%%/wait%%:", w);
                        for (int k : {0, 2, 4, 5, 6, 7, 9}) fprintf(stderr, "  %s %4.1f/%4.1f", names[k], 100.0 * wv[w * 20 + k] / tot, 100.0 * wv[w * 20 + 10 + k] / tot);
                        fprintf(stderr, "\n");
                    }
            }
            // residency of the workgroups of the LAST step launch (wall clock, 100 MHz): how much of the kernel's span a replica is resident for,
            // and how the end times spread inside a CU and between CUs (third word: XCC_ID << 32 | HW_ID)
            std::vector<long long> w(3 * (size_t)nrep_);
            if (hipMemcpy(w.data(), dsys_.prof + 16, w.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess && w[1] > 0) {
                long long s0 = w[0], s1 = w[0], e0 = w[1], e1 = w[1];
                double dsum = 0;
                std::vector<long long> dur(nrep_);
                std::map<long long, std::vector<long long>> by_cu;
                for (int r = 0; r < nrep_; r++) {
                    s0 = std::min(s0, w[3 * r]); s1 = std::max(s1, w[3 * r]); e0 = std::min(e0, w[3 * r + 1]); e1 = std::max(e1, w[3 * r + 1]);
                    dur[r] = w[3 * r + 1] - w[3 * r]; dsum += (double)dur[r];
                    by_cu[((w[3 * r + 2] >> 32) << 8) | ((w[3 * r + 2] >> 8) & 0xff)].push_back(dur[r]);
                }
                std::sort(dur.begin(), dur.end());
                fprintf(stderr, "[awsem prof] last launch: span %.1f us, first..last start %.1f us, first..last end %.1f us; workgroup residency mean %.1f us "
                        "(%.1f %% of the span), min %.1f, median %.1f, p90 %.1f, max %.1f\n", (e1 - s0) * 0.01, (s1 - s0) * 0.01, (e1 - e0) * 0.01, dsum / nrep_ * 0.01,
                        100.0 * dsum / nrep_ / (double)(e1 - s0), dur[0] * 0.01, dur[nrep_ / 2] * 0.01, dur[(size_t)nrep_ * 9 / 10] * 0.01, dur[nrep_ - 1] * 0.01);
                if (FILE* f = strlen(getenv("AWSEM_PROF_DUMP")) > 1 ? fopen(getenv("AWSEM_PROF_DUMP"), "w") : nullptr) {   // raw stamps for offline analysis
                    std::vector<long long> lc((size_t)nrep_ * AWSEM_MAX_LISTS, 0);   // list lengths at the end of the launch (-DAWSEM_WALL builds)
                    (void)hipMemcpy(lc.data(), dsys_.prof + 16 + 7 * (size_t)nrep_ + 16 * 20, lc.size() * sizeof(long long), hipMemcpyDeviceToHost);
                    for (int r = 0; r < nrep_; r++) {
                        fprintf(f, "%d %lld %lld %llx", r, w[3 * r] - s0, w[3 * r + 1] - s0, (unsigned long long)w[3 * r + 2]);
                        for (int l = 0; l < dsys_.n_lists; l++) fprintf(f, " %lld", lc[(size_t)r * AWSEM_MAX_LISTS + l]);
                        fprintf(f, "\n");
                    }
                    fclose(f);
                }
                std::vector<double> cu_max, cu_spread, cu_n;
                for (auto& kv : by_cu) {
                    auto& v = kv.second;
                    std::sort(v.begin(), v.end());
                    cu_max.push_back(v.back() * 0.01); cu_spread.push_back((v.back() - v.front()) * 0.01); cu_n.push_back((double)v.size());
                }
                auto stat = [](std::vector<double> v, const char* what) {
                    std::sort(v.begin(), v.end());
                    fprintf(stderr, "[awsem prof]   per CU %-28s min %.1f  median %.1f  max %.1f\n", what, v.front(), v[v.size() / 2], v.back());
                };
                fprintf(stderr, "[awsem prof]   %zu CUs seen\n", by_cu.size());
                stat(cu_n, "workgroups"); stat(cu_max, "last end (us)"); stat(cu_spread, "first..last end (us)");
                int shown = 0;
                for (auto& kv : by_cu) if (shown++ < 6) { fprintf(stderr, "[awsem prof]   cu %llx:", (unsigned long long)kv.first); for (long long d : kv.second) fprintf(stderr, " %.1f", d * 0.01); fprintf(stderr, "\n"); }
            }
        }
        for (auto& G : cell_graph_) if (G.exec) (void)hipGraphExecDestroy(G.exec);
        for (void* p : allocs_) (void)hipFree(p);
        if (h_lmean_) (void)hipHostFree(h_lmean_);
        if (lmean_ev_) (void)hipEventDestroy(lmean_ev_);
        for (int k = 0; k < NRING; k++) { if (ring0_[k]) (void)hipEventDestroy(ring0_[k]); if (ring1_[k]) (void)hipEventDestroy(ring1_[k]); if (ringT_[k]) (void)hipHostFree(ringT_[k]); if (ringS_[k]) (void)hipHostFree(ringS_[k]); }
        if (ev0_) (void)hipEventDestroy(ev0_);
        for (auto& sn : snap_) {
            if (sn.ev) (void)hipEventDestroy(sn.ev);
            for (void* q : {(void*)sn.e, (void*)sn.ke, (void*)sn.x1, (void*)sn.xs, (void*)sn.vs}) if (q) (void)hipHostFree(q);
        }
        if (ev1_) (void)hipEventDestroy(ev1_);
        if (stream_) (void)hipStreamDestroy(stream_);
    }

    void finalize(const HostSystem& H) override;

    void set_positions(int rep, const double* xyz) override { set_state(pos_, rep, xyz); cell_stale_ = true; lists_valid_ = false; }
    void get_positions(int rep, double* xyz) override { get_state(pos_, rep, xyz); }
    void set_velocities(int rep, const double* v) override { set_state(vel(), rep, v); }
    void get_velocities(int rep, double* v) override { get_state(vel(), rep, v); }

    void compute_virtual_sites() override {
        HIP_TRY(hipSetDevice(device_));
        synchronize();
        cell_stale_ = true; lists_valid_ = false;
        if (cell_mode_) hipLaunchKernelGGL(cell_vsite_kernel<real>, cell_grid(dsys_.nvs), dim3(256), 0, stream_, dsys_, pos_);
        else hipLaunchKernelGGL(awsem_vsite_kernel<real>, dim3(nrep_), dim3(64), 0, stream_, dsys_, pos_);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream_));
    }
    void maxwell(const double* T, uint64_t seed) override {
        HIP_TRY(hipSetDevice(device_));
        synchronize();
        upload_T(T);
        hipLaunchKernelGGL(awsem_maxwell_kernel<real>, dim3(nrep_), dim3(256), 0, stream_, dsys_, vel(), dT_, (unsigned long long)seed);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream_));
    }
    void scale_velocities(const double* scale) override {
        HIP_TRY(hipSetDevice(device_));
        synchronize();
        HIP_TRY(hipMemcpyAsync(dtmp_, scale, sizeof(double) * nrep_, hipMemcpyHostToDevice, stream_));   // not dT_: on a replica-exchange handle that holds the ladder's temperatures
        HIP_TRY(hipStreamSynchronize(stream_));   // scale is a caller-owned pageable buffer
        hipLaunchKernelGGL(awsem_scale_velocities_kernel<real>, dim3(nrep_), dim3(256), 0, stream_, vel(), dtmp_, 3 * natoms_);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream_));
    }
    void energy_by_group(unsigned mask, double* e) override {
        HIP_TRY(hipSetDevice(device_));
        if (cell_mode_) { cell_evaluate(true, mask, 0, e, nullptr); return; }
        const bool reuse = query_reuse();
        HIP_TRY(hipEventRecord(ev0_, stream_));
        launch_eval<true>(dim3(nrep_), stream_, reuse ? dsys_ : eval_sys(), pos_, mask, 0, dE_, (double*)nullptr, (const double*)nullptr, (int)reuse, 0, (double*)nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev1_, stream_));
        HIP_TRY(hipMemcpyAsync(e, dE_, sizeof(double) * nrep_ * AWSEM_NGROUPS, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        record_ms();
        check_status();
    }
    void forces(unsigned mask, int redistribute, double* f) override {
        HIP_TRY(hipSetDevice(device_));
        if (cell_mode_) { cell_evaluate(false, mask, redistribute, nullptr, f); return; }
        const bool reuse = query_reuse();
        HIP_TRY(hipEventRecord(ev0_, stream_));
        launch_eval<false>(dim3(nrep_), stream_, reuse ? dsys_ : eval_sys(), pos_, mask, redistribute, (double*)nullptr, dF_, (const double*)nullptr, (int)reuse, 0, (double*)nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev1_, stream_));
        HIP_TRY(hipMemcpyAsync(f, dF_, sizeof(double) * 3 * natoms_ * nrep_, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        record_ms();
        check_status();
    }
    // ---- replica exchange on the device (include/awsem_hip.h: awsem_remd_*)
    void remd_setup(int n_total, const double* ladder, const int32_t* where, int local_start, uint64_t seed) override {
        HIP_TRY(hipSetDevice(device_));
        synchronize();
        if (n_total < nrep_ || local_start < 0 || local_start + nrep_ > n_total) throw std::string("remd_setup: the local replicas must be a block of the ladder");
        if (n_total > 8192) throw std::string("remd_setup: at most 8192 replicas in one ladder");
        remd_n_ = n_total; remd_start_ = local_start; remd_seed_ = seed;
        std::vector<int> slot(n_total), wh(n_total);
        double tmax = 0;
        for (int r = 0; r < n_total; r++) {
            slot[r] = r; wh[r] = where ? where[r] : r; tmax = std::max(tmax, ladder[r]);
            if (wh[r] < 0) throw std::string("remd_setup: negative gather index");
            if (!(ladder[r] > 0)) throw std::string("remd_setup: temperatures must be positive");
        }
        remd_tmax_ = tmax;
        // a second set-up on the same handle (a new ladder, a restart) replaces the first one's buffers
        (void)vel();   // (a rescale still pending from the ladder this one replaces)
        synchronize();
        dfree(remd_ladder_); dfree(remd_slot_); dfree(remd_where_); dfree(remd_counts_); dfree(remd_U_); dfree(remd_scale_); dfree(remd_T_);
        remd_ladder_ = upload(std::vector<double>(ladder, ladder + n_total));
        remd_slot_ = const_cast<int*>(upload(slot));
        remd_where_ = upload(wh);
        remd_counts_ = dalloc<unsigned long long>(2 * (size_t)std::max(1, n_total - 1));
        HIP_TRY(hipMemset(remd_counts_, 0, 2 * (size_t)std::max(1, n_total - 1) * sizeof(unsigned long long)));
        remd_U_ = dalloc<double>(n_total);
        HIP_TRY(hipMemset(remd_U_, 0, n_total * sizeof(double)));
        remd_scale_ = const_cast<double*>(upload(std::vector<double>(nrep_, 1.0)));
        // The temperatures the replicas run at.  A buffer of its own, written by this call and by the sweep kernel only: awsem_step(T) and
        // awsem_set_velocities_to_temperature(T) on the same handle go through dT_ and cannot disturb the ladder's assignment.
        remd_T_ = dalloc<double>(nrep_);
        HIP_TRY(hipMemcpy(remd_T_, ladder + local_start, sizeof(double) * nrep_, hipMemcpyHostToDevice));   // replica r starts at ladder[r]
    }
    // The velocities with every rescale applied.  All of the engine reads and writes them through here except the step launch, which
    // applies a pending rescale itself.  `wait`: for the readers that copy on the null stream (stream_ does not order with it); the ones
    // that queue behind it on stream_ pass false.
    real* vel(bool wait = true) {
        if (vel_scale_pending_) {
            vel_scale_pending_ = false;
            HIP_TRY(hipSetDevice(device_));
            hipLaunchKernelGGL(awsem_take_scale_kernel<real>, dim3(nrep_), dim3(256), 0, stream_, vel_raw_, remd_scale_, 3 * natoms_);
            HIP_TRY(hipGetLastError());
            if (wait) HIP_TRY(hipStreamSynchronize(stream_));
        }
        return vel_raw_;
    }
    void need_remd() const { if (!remd_n_) throw StateError{"replica exchange: call awsem_remd_setup first"}; }
    void step_remd(int nsteps, double dt, double friction, uint64_t seed, int64_t step0) override {
        need_remd();
        step_impl(nsteps, nullptr, dt, friction, seed, step0);
    }
    void remd_energies(unsigned mask, double* d_energy) override {
        need_remd();
        HIP_TRY(hipSetDevice(device_));
        double* out = d_energy ? d_energy : remd_U_ + remd_start_;
        if (cell_mode_) {   // the cell-list evaluation is a host-driven sequence with its own retries: one upload of n_replicas doubles
            std::vector<double> eg((size_t)nrep_ * AWSEM_NGROUPS), e(nrep_, 0.0);
            cell_evaluate(true, mask, 0, eg.data(), nullptr);
            for (int r = 0; r < nrep_; r++) for (int g = 0; g < AWSEM_NGROUPS; g++) if ((mask >> g) & 1u) e[r] += eg[(size_t)r * AWSEM_NGROUPS + g];
            HIP_TRY(hipMemcpy(out, e.data(), sizeof(double) * nrep_, hipMemcpyHostToDevice));
            return;
        }
        const bool reuse = query_reuse();
        // the sum over the masked groups comes out of the energy kernel itself (one launch fewer per exchange)
        launch_eval<true>(dim3(nrep_), stream_, reuse ? dsys_ : eval_sys(), pos_, mask, 0, dE_, (double*)nullptr, (const double*)nullptr, (int)reuse, 0, out);
        HIP_TRY(hipGetLastError());
    }
    void remd_sweep(const double* d_all, int64_t iteration) override {
        need_remd();
        HIP_TRY(hipSetDevice(device_));
        const int N = remd_n_;
        if (remd_mixing_ == 1) {
            const size_t lds = (2 * (size_t)N + AWSEM_MIX_BATCH) * sizeof(double) + (4 * (size_t)N + 2 * AWSEM_MIX_BATCH) * sizeof(int);
            hipLaunchKernelGGL(awsem_remd_mix_all_kernel, dim3(1), dim3(256), lds, stream_, d_all ? d_all : (const double*)remd_U_, remd_where_,
                               remd_ladder_, remd_slot_, remd_counts_, remd_counts_ + std::max(1, N - 1), N, (long long)iteration, (unsigned long long)remd_seed_,
                               remd_start_, nrep_, remd_T_, remd_scale_);
        } else
        hipLaunchKernelGGL(awsem_remd_sweep_kernel, dim3(1), dim3(256), 2 * (size_t)N * sizeof(int), stream_, d_all ? d_all : (const double*)remd_U_, remd_where_,
                           remd_ladder_, remd_slot_, remd_counts_, remd_counts_ + std::max(1, N - 1), N, (long long)iteration, (unsigned long long)remd_seed_,
                           remd_start_, nrep_, remd_T_, remd_scale_);
        // The velocity rescale is not a launch of its own: the factors stay in remd_scale_ (a later sweep multiplies into them) until the
        // next step launch takes them in with its velocity load, or until somebody else asks for the velocities (vel()).
        vel_scale_pending_ = true;
        HIP_TRY(hipGetLastError());
    }
    void remd_set_mixing(int scheme) override {
        if (scheme != 0 && scheme != 1) throw std::string("remd_set_mixing: scheme must be 0 (neighbour swaps) or 1 (swap-all)");
        if (scheme == 1 && remd_n_ > 1024) throw std::string("remd_set_mixing: swap-all resolves N^3 attempts per iteration in one chain; at most 1024 replicas");
        remd_mixing_ = scheme;
    }
    void remd_state(int32_t* slot, int64_t* attempted, int64_t* accepted, double* T_local) override {
        need_remd();
        HIP_TRY(hipSetDevice(device_));
        synchronize();
        const size_t np = (size_t)std::max(1, remd_n_ - 1);
        if (slot) HIP_TRY(hipMemcpy(slot, remd_slot_, sizeof(int) * remd_n_, hipMemcpyDeviceToHost));
        if (attempted && remd_n_ > 1) HIP_TRY(hipMemcpy(attempted, remd_counts_, sizeof(long long) * (remd_n_ - 1), hipMemcpyDeviceToHost));
        if (accepted && remd_n_ > 1) HIP_TRY(hipMemcpy(accepted, remd_counts_ + np, sizeof(long long) * (remd_n_ - 1), hipMemcpyDeviceToHost));
        if (T_local) HIP_TRY(hipMemcpy(T_local, remd_T_, sizeof(double) * nrep_, hipMemcpyDeviceToHost));
    }
    void step(int nsteps, const double* T, double dt, double friction, uint64_t seed, int64_t step0) override { step_impl(nsteps, T, dt, friction, seed, step0); }
    // nstages temperature stages of steps_per_stage steps in ONE launch (T: [stage][replica]): the trajectory of nstages step() calls,
    // without the launch boundaries between them (state write-back and reload, table staging, and the wait for the slowest replica)
    void step_stages(int nstages, int steps_per_stage, const double* T, double dt, double friction, uint64_t seed, int64_t step0) override {
        if (cell_mode_ || !step_ok_ || nstages == 1) {   // the cell-list path (and the error path) stage by stage
            for (int g = 0; g < nstages; g++) step_impl(steps_per_stage, T + (size_t)g * nrep_, dt, friction, seed, step0 + (int64_t)g * steps_per_stage);
            return;
        }
        for (int g = 0; g < nstages; g += AWSEM_MAX_STAGES) {
            const int n = std::min(AWSEM_MAX_STAGES, nstages - g);
            step_impl(n * steps_per_stage, T + (size_t)g * nrep_, dt, friction, seed, step0 + (int64_t)g * steps_per_stage, n);
        }
    }
    // T == nullptr: the temperatures already on the device (replica exchange) drive the launch
    void step_impl(int nsteps, const double* T, double dt, double friction, uint64_t seed, int64_t step0, int nstages = 1) {
        if (cell_mode_) { (void)vel(false); cell_step(nsteps, T, dt, friction, seed, step0); return; }   // (no rescale left to settle inside a graph capture)
        if (!step_ok_) {
            char msg[256];
            snprintf(msg, sizeof msg, "dynamics of %d atoms needs %zu bytes of LDS per replica in this precision; the replica-resident path is limited to %zu",
                     natoms_, lds_step_, lds_max_);
            throw std::string(msg);
        }
        HIP_TRY(hipSetDevice(device_));
        if (nstages > 1) {   // the skin follows the hottest stage
            std::vector<double> hot(T, T + nrep_);
            for (int g = 1; g < nstages; g++) for (int r = 0; r < nrep_; r++) hot[r] = std::max(hot[r], T[(size_t)g * nrep_ + r]);
            apply_skin(hot.data());
        } else apply_skin(T);
        begin_async(T, nstages);
        if ((dsys_.prio & 0xff) == 3 && place_every_ > 0 && launch_ > 0 && launch_ % place_every_ == 0 && nrep_ <= 8192)   // deal the replicas to the CUs anew
            hipLaunchKernelGGL(awsem_place_kernel, dim3(1), dim3(1024), (size_t)nrep_ * sizeof(int), stream_, dsys_.prof, nrep_, dsys_.pace_group);
        const bool reuse = list_reuse_ && lists_valid_ && skin_ <= lists_skin_;
        dsys_.vel_scale = vel_scale_pending_ ? remd_scale_ : nullptr;   // the last sweep's rescale rides in this launch's velocity load
        vel_scale_pending_ = false;
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(nrep_), dim3(block_), lds_step_, stream_, dsys_, pos_, vel_raw_, step_T_, nsteps, dt, friction,
                               (unsigned long long)seed, (long long)step0, launch_++, (int)reuse, nsteps / nstages);
        };
        if (step_lean_ == 1) launch(awsem_step_kernel<real, 1>);
        else if (step_lean_ == 2) launch(awsem_step_kernel<real, 2>);
        else if (step_lean_ == 3) launch(awsem_step_kernel<real, 3>);
        else launch(awsem_step_kernel<real, 0>);
        dsys_.vel_scale = nullptr;
        // the skin the lists counted after this launch were built under: lists carried in from the previous launch carry ITS skin
        const double built_skin = reuse ? std::max(skin_, lists_skin_) : skin_;
        lists_valid_ = nsteps > 0;
        lists_skin_ = skin_;
        if (skin_adapt_ && !skin_fixed_ && h_lmean_ && nsteps > 0 && launch_ % 8 == 1) {   // (launch_ was incremented above)
            lmean_skin_req_ = built_skin;
            lmean_due_ = launch_ + AWSEM_LMEAN_LAG;   // the launch whose skin takes the answer
            if (!lmean_ev_) HIP_TRY(hipEventCreateWithFlags(&lmean_ev_, hipEventDisableTiming));
            hipLaunchKernelGGL(awsem_list_mean_kernel, dim3(1), dim3(256), 0, stream_, (const long long*)dsys_.prof, nrep_, (double)++lmean_tag_, (volatile double*)h_lmean_);
            HIP_TRY(hipEventRecord(lmean_ev_, stream_));
        }
        HIP_TRY(hipGetLastError());
        end_async();   // asynchronous: resolved by synchronize() / last_kernel_ms
    }
    void minimize(double tolerance, int max_iterations, int32_t* iterations, double* rms) override {
        if (cell_mode_) { cell_minimize(tolerance, max_iterations > 0 ? max_iterations : 100000, iterations, rms); return; }
        if (!step_ok_) throw std::string("minimize: the replica does not fit the LDS of one CU together with its velocities in this precision");
        HIP_TRY(hipSetDevice(device_));
        if (!dmin_it_) { dmin_it_ = dalloc<int>(nrep_); dmin_rms_ = dalloc<double>(nrep_); }
        HIP_TRY(hipEventRecord(ev0_, stream_));
        lists_valid_ = false;
        hipLaunchKernelGGL(awsem_minimize_kernel<real>, dim3(nrep_), dim3(block_), lds_step_, stream_, dsys_, pos_, tolerance,
                           max_iterations > 0 ? max_iterations : 100000, dmin_it_, dmin_rms_);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev1_, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        record_ms();
        check_status();
        if (iterations) HIP_TRY(hipMemcpy(iterations, dmin_it_, sizeof(int) * nrep_, hipMemcpyDeviceToHost));
        if (rms) HIP_TRY(hipMemcpy(rms, dmin_rms_, sizeof(double) * nrep_, hipMemcpyDeviceToHost));
    }
    void kinetic_energy(double* ke) override {
        HIP_TRY(hipSetDevice(device_));
        synchronize();
        hipLaunchKernelGGL(awsem_kinetic_kernel<real>, dim3(nrep_), dim3(256), 0, stream_, dsys_, vel(), dtmp_);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ke, dtmp_, sizeof(double) * nrep_, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
    }
    // whole ensemble at once: one copy of the SoA device array, transposed on the host
    void set_state_all(const double* xyz, const double* v) override {
        HIP_TRY(hipSetDevice(device_));
        synchronize();
        const size_t n3 = (size_t)3 * natoms_;
        std::vector<real> t(n3 * nrep_);
        for (int pass = 0; pass < 2; pass++) {
            const double* src = pass ? v : xyz;
            if (!src) continue;
            for (int r = 0; r < nrep_; r++) for (int i = 0; i < natoms_; i++) for (int k = 0; k < 3; k++)
                t[r * n3 + (size_t)k * natoms_ + in_(i)] = (real)src[r * n3 + 3 * (size_t)i + k];
            HIP_TRY(hipMemcpy(pass ? vel() : pos_, t.data(), t.size() * sizeof(real), hipMemcpyHostToDevice));
            if (!pass) cell_stale_ = true; lists_valid_ = false;
        }
    }
    void get_state_all(double* xyz, double* v) override {
        HIP_TRY(hipSetDevice(device_));
        synchronize();
        const size_t n3 = (size_t)3 * natoms_;
        std::vector<real> t(n3 * nrep_);
        for (int pass = 0; pass < 2; pass++) {
            double* dst = pass ? v : xyz;
            if (!dst) continue;
            HIP_TRY(hipMemcpy(t.data(), pass ? vel() : pos_, t.size() * sizeof(real), hipMemcpyDeviceToHost));
            for (int r = 0; r < nrep_; r++) for (int i = 0; i < natoms_; i++) for (int k = 0; k < 3; k++)
                dst[r * n3 + 3 * (size_t)i + k] = (double)t[r * n3 + (size_t)k * natoms_ + in_(i)];
        }
    }
    // ---- asynchronous snapshots for reporters (include/awsem_hip.h: awsem_snapshot_begin / _end)
    // begin: behind whatever the stream holds, the masked potential energies and the kinetic energies of all replicas, the positions of
    // one replica and (with_state) the whole ensemble state are evaluated / copied into pinned host memory of slot 0 or 1; nothing waits.
    // end: waits for that slot's event only -- launches queued after `begin` keep the GPU busy meanwhile -- and converts exactly as the
    // synchronous queries do (awsem_energy, awsem_kinetic_energy, awsem_get_positions, awsem_get_state): the same numbers.
    struct Snap { hipEvent_t ev = nullptr; double* e = nullptr; double* ke = nullptr; real* x1 = nullptr; real* xs = nullptr; real* vs = nullptr;
                  unsigned mask = 0; int replica = -1, energy_replica = -1; bool with_state = false, busy = false; };
    Snap snap_[2];
    void snapshot_begin(int slot, unsigned mask, int energy_replica, int replica, int with_state) override {
        if (slot < 0 || slot > 1) throw std::string("snapshot_begin: slot must be 0 or 1");
        if (replica >= nrep_ || energy_replica >= nrep_) throw std::string("snapshot_begin: replica out of range");
        Snap& sn = snap_[slot];
        if (sn.busy) throw StateError{"snapshot_begin: the slot holds a snapshot that has not been collected (awsem_snapshot_end)"};
        HIP_TRY(hipSetDevice(device_));
        const size_t n3 = (size_t)3 * natoms_;
        if (!sn.ev) {
            HIP_TRY(hipEventCreateWithFlags(&sn.ev, hipEventDisableTiming));
            HIP_TRY(hipHostMalloc((void**)&sn.e, sizeof(double) * nrep_ * AWSEM_NGROUPS, hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void**)&sn.ke, sizeof(double) * nrep_, hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void**)&sn.x1, sizeof(real) * n3, hipHostMallocDefault));
        }
        if (with_state && !sn.xs) {
            HIP_TRY(hipHostMalloc((void**)&sn.xs, sizeof(real) * n3 * nrep_, hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void**)&sn.vs, sizeof(real) * n3 * nrep_, hipHostMallocDefault));
        }
        sn.mask = mask; sn.replica = replica; sn.with_state = with_state != 0;
        if (cell_mode_) {   // the cell-list evaluation is a host-driven sequence with its own retries: taken at once
            cell_evaluate(true, mask, 0, sn.e, nullptr);
        } else if (mask) {
            const bool reuse = query_reuse();
            // one replica's energies (what StateDataReporter prints) are one workgroup; the other rows of the slot are left as NaN
            const int first = energy_replica >= 0 ? energy_replica : 0, count = energy_replica >= 0 ? 1 : nrep_;
            launch_eval<true>(dim3(count), stream_, reuse ? dsys_ : eval_sys(), pos_, mask, 0, dE_, (double*)nullptr, (const double*)nullptr, (int)reuse, first, (double*)nullptr);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(sn.e + (size_t)first * AWSEM_NGROUPS, dE_ + (size_t)first * AWSEM_NGROUPS, sizeof(double) * count * AWSEM_NGROUPS, hipMemcpyDeviceToHost, stream_));
        }
        sn.energy_replica = cell_mode_ ? -1 : energy_replica;
        hipLaunchKernelGGL(awsem_kinetic_kernel<real>, dim3(nrep_), dim3(256), 0, stream_, dsys_, vel(false), dtmp_);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(sn.ke, dtmp_, sizeof(double) * nrep_, hipMemcpyDeviceToHost, stream_));
        if (replica >= 0) HIP_TRY(hipMemcpyAsync(sn.x1, pos_ + (size_t)replica * n3, sizeof(real) * n3, hipMemcpyDeviceToHost, stream_));
        if (with_state) {
            HIP_TRY(hipMemcpyAsync(sn.xs, pos_, sizeof(real) * n3 * nrep_, hipMemcpyDeviceToHost, stream_));
            HIP_TRY(hipMemcpyAsync(sn.vs, vel(), sizeof(real) * n3 * nrep_, hipMemcpyDeviceToHost, stream_));
        }
        HIP_TRY(hipEventRecord(sn.ev, stream_));
        sn.busy = true;
    }
    void snapshot_end(int slot, double* e, double* ke, double* xyz_replica, double* xyz_all, double* v_all) override {
        if (slot < 0 || slot > 1) throw std::string("snapshot_end: slot must be 0 or 1");
        Snap& sn = snap_[slot];
        if (!sn.busy) throw StateError{"snapshot_end: no snapshot was begun in this slot"};
        if ((xyz_all || v_all) && !sn.with_state) throw StateError{"snapshot_end: the snapshot was begun without the ensemble state"};
        if (xyz_replica && sn.replica < 0) throw StateError{"snapshot_end: the snapshot was begun without a replica's positions"};
        {   // not while another thread captures a graph on this handle's stream (cell_step): hipEventSynchronize on an event of a stream
            // that is being captured invalidates the capture, whichever thread calls it and although the event was recorded before
            std::lock_guard<std::mutex> hold(capture_mu_);
            HIP_TRY(hipSetDevice(device_));
            HIP_TRY(hipEventSynchronize(sn.ev));
        }
        sn.busy = false;
        const size_t n3 = (size_t)3 * natoms_;
        if (e) for (int r = 0; r < nrep_; r++) {
            if (sn.energy_replica >= 0 && r != sn.energy_replica) { e[r] = std::numeric_limits<double>::quiet_NaN(); continue; }
            double s = 0;
            if (sn.mask) for (int g = 0; g < AWSEM_NGROUPS; g++) if ((sn.mask >> g) & 1u) s += sn.e[(size_t)r * AWSEM_NGROUPS + g];
            e[r] = s;
        }
        if (ke) memcpy(ke, sn.ke, sizeof(double) * nrep_);
        if (xyz_replica) for (int i = 0; i < natoms_; i++) for (int k = 0; k < 3; k++) xyz_replica[3 * i + k] = (double)sn.x1[(size_t)k * natoms_ + in_(i)];
        for (int pass = 0; pass < 2; pass++) {
            double* dst = pass ? v_all : xyz_all;
            const real* t = pass ? sn.vs : sn.xs;
            if (!dst) continue;
            for (int r = 0; r < nrep_; r++) for (int i = 0; i < natoms_; i++) for (int k = 0; k < 3; k++)
                dst[r * n3 + 3 * (size_t)i + k] = (double)t[r * n3 + (size_t)k * natoms_ + in_(i)];
        }
    }
    void synchronize() override {
        HIP_TRY(hipSetDevice(device_));
        HIP_TRY(hipStreamSynchronize(stream_));
        if (timing_pending_) {
            record_ring_ms();
            if (cell_mode_ && !cell_pending_.empty()) cell_replay_if_overflowed();
            cell_pending_.clear();
            check_status();
        }
    }
    int neighbors(int rep, int kind, int index, double skin, int32_t* pairs, int capacity) override;
    // No hipMalloc / hipFree / host copy / synchronisation in here: one memset node and one kernel on the caller's stream.
    void energy_noalloc(int nrep, const double* d_xyz, double* d_energy, hipStream_t stream) override {
        if (cell_mode_) throw StateError{"energy_by_group_noalloc: this handle runs on the cell-list path (the system does not fit the LDS of one CU); use awsem_energy_by_group"};
        unsigned char* base = device_ < AWSEM_MAX_DEVICES ? (unsigned char*)g_noalloc_ptr[device_].load(std::memory_order_acquire) : nullptr;
        if (!base) throw StateError{"energy_by_group_noalloc: no scratch address for this device (finalize did not run here)"};
        const unsigned long long need = 256ull + (unsigned long long)nrep * (unsigned long long)dsys_.list_stride * 4ull;
        if (need > AWSEM_NOALLOC_SCRATCH_BYTES) {
            char msg[320];
            snprintf(msg, sizeof msg, "energy_by_group_noalloc: %d replicas x %lld pair-list entries reserved per replica need %llu bytes of scratch; the "
                     "static array holds %u (at most %llu replicas of this system per call)", nrep, dsys_.list_stride, need,
                     (unsigned)AWSEM_NOALLOC_SCRATCH_BYTES, (unsigned long long)((AWSEM_NOALLOC_SCRATCH_BYTES - 256ull) / ((unsigned long long)dsys_.list_stride * 4ull)));
            throw std::string(msg);
        }
        int caller_device = device_;                 // the caller's current device is left as it was found (no allocation in either call)
        HIP_TRY(hipGetDevice(&caller_device));
        if (caller_device != device_) HIP_TRY(hipSetDevice(device_));
        DevSys<real> S = dsys_;                      // the handle's descriptors, with the lists and the status word moved into the static array
        S.status = (int*)base;
        S.lists = (unsigned*)(base + 256);
        HIP_TRY(hipMemsetAsync(base, 0, 256, stream));   // the status word must start at zero; the lists are written before they are read
        launch_eval<true>(dim3(nrep), stream, S, (const real*)nullptr, 0xffffffffu, 0,
                           d_energy, (double*)nullptr, d_xyz, 0, 0, (double*)nullptr);
        const hipError_t launched = hipGetLastError();
        if (caller_device != device_) (void)hipSetDevice(caller_device);
        HIP_TRY(launched);
    }

private:
    template <class T> T* dalloc(size_t n) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
        allocs_.push_back(p);
        return (T*)p;
    }
    void dfree(const void* p) {   // for buffers that are re-allocated during a run (neighbour rows); the stream must be idle
        if (!p) return;
        auto it = std::find(allocs_.begin(), allocs_.end(), (void*)p);
        if (it != allocs_.end()) { allocs_.erase(it); (void)hipFree((void*)p); }
    }
    template <class T> const T* upload(const std::vector<T>& v) {
        T* p = dalloc<T>(v.size());
        if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return p;
    }
    // functional lambda / gamma table: per-residue words and the small coefficient tables, in one device struct
    const HbFn<real>* upload_fn(const HFn& f, int nres, double scale) {
        std::vector<int> rinfo(nres);
        for (int i = 0; i < nres; i++) {
            const int* q = &f.info[4 * (size_t)i];
            const bool edge = q[2] != 0;
            const int prev = (i > 0 && !edge) ? f.info[4 * (size_t)(i - 1)] : 0, next = (i + 1 < nres && !edge) ? f.info[4 * (size_t)(i + 1)] : 0;
            rinfo[i] = q[0] | (prev << 5) | (next << 10) | (edge ? 1 << 15 : 0) | (q[3] ? 1 << 16 : 0) | (q[1] << 17);
        }
        std::vector<HbFn<real>> one(1);
        HbFn<real>& d = one[0];
        memset(&d, 0, sizeof d);
        d.kind = f.kind; d.rinfo = upload(rinfo); d.scale = (real)scale;
        const double* c = f.coef.data();
        for (int a = 0; a < 5; a++) for (int b = 0; b < 3; b++) d.cc[a][b] = (real)c[3 * a + b];
        for (int a = 0; a < 3; a++) d.g[a] = (real)c[15 + a];
        for (int a = 0; a < 400; a++) { d.t1[a] = (real)c[18 + a]; d.t2[a] = (real)c[418 + a]; }
        for (int a = 0; a < 20; a++) d.ob[a] = (real)c[818 + a];
        return upload(one);
    }
    const real* upload_real(const std::vector<double>& v, double scale = 1.0) {
        std::vector<real> t(v.size());
        for (size_t i = 0; i < v.size(); i++) t[i] = (real)(v[i] * scale);
        return upload(t);
    }
    // start / end of an asynchronous launch: temperatures through a pinned ring buffer, events from the ring
    void begin_async(const double* T, int nstages = 1) {
        if (npend_ == NRING) synchronize();
        const int k = npend_;
        if (!ring0_[k]) { HIP_TRY(hipEventCreate(&ring0_[k])); HIP_TRY(hipEventCreate(&ring1_[k])); HIP_TRY(hipHostMalloc((void**)&ringT_[k], sizeof(double) * nrep_, hipHostMallocDefault)); }
        if (nstages > 1) {   // (replica-resident path only.)  Several stages: the kernel reads a stage's temperature in the middle of its run, so
            // the block goes to device memory -- a read of host memory there would stall the workgroup for microseconds, stage after stage
            if (!ringS_[k]) HIP_TRY(hipHostMalloc((void**)&ringS_[k], sizeof(double) * nrep_ * AWSEM_MAX_STAGES, hipHostMallocDefault));
            if (!dT_stages_) dT_stages_ = dalloc<double>((size_t)nrep_ * AWSEM_MAX_STAGES);
            memcpy(ringS_[k], T, sizeof(double) * nrep_ * nstages);
            HIP_TRY(hipMemcpyAsync(dT_stages_, ringS_[k], sizeof(double) * nrep_ * nstages, hipMemcpyHostToDevice, stream_));
            step_T_ = dT_stages_;
            HIP_TRY(hipEventRecord(ring0_[k], stream_));
            return;
        }
        // null: the temperatures are on the device already (replica exchange).  Otherwise the kernel reads them from this launch's
        // slot of pinned host memory itself (one double per workgroup): no copy between two launches on the stream
        step_T_ = dT_;
        if (!T) {   // replica exchange: the ladder's own buffer (the cell-list path bakes dT_ into its captured graphs: one small copy on the stream)
            if (cell_mode_) HIP_TRY(hipMemcpyAsync(dT_, remd_T_, sizeof(double) * nrep_, hipMemcpyDeviceToDevice, stream_));
            else step_T_ = remd_T_;
        }
        if (T) {
            memcpy(ringT_[k], T, sizeof(double) * nrep_);
            if (cell_mode_ || !zero_copy_T_) HIP_TRY(hipMemcpyAsync(dT_, ringT_[k], sizeof(double) * nrep_, hipMemcpyHostToDevice, stream_));
            else step_T_ = ringT_[k];
        }
        HIP_TRY(hipEventRecord(ring0_[k], stream_));
    }
    const double* step_T_ = nullptr;
    bool zero_copy_T_ = !(getenv("AWSEM_ZERO_COPY_T") && atoi(getenv("AWSEM_ZERO_COPY_T")) == 0);
    void end_async() {
        HIP_TRY(hipEventRecord(ring1_[npend_], stream_));
        npend_++;
        timing_pending_ = true;
    }
    void upload_T(const double* T) {
        HIP_TRY(hipMemcpyAsync(dT_, T, sizeof(double) * nrep_, hipMemcpyHostToDevice, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));   // T is a caller-owned pageable buffer
    }
    void check_status() {
        int st = 0;
        HIP_TRY(hipMemcpy(&st, dstatus_, sizeof st, hipMemcpyDeviceToHost));
        if (st) {
            HIP_TRY(hipMemset(dstatus_, 0, sizeof st));
            if (st & 2) {
                cell_stale_ = true; lists_valid_ = false;
                throw std::string("neighbour list overflow on the cell-list path outside a replayable launch sequence: raise AWSEM_CELL_NEIGHBORS (partners reserved per contact site, default 96, grown to twice the fullest row when dynamics start)");
            }
            if (st & 8) {
                cell_stale_ = true; lists_valid_ = false;
                throw std::string("cell-list path: a grid barrier of the list rebuild timed out (its workgroups were not all resident: is the GPU shared with other work?); "
                                  "AWSEM_CELL_FUSED=0 selects the multi-kernel rebuild");
            }
            if (st & 4) throw std::string("a force exceeded the range of the fp32 fixed-point accumulators (2.6e5 kJ/mol/nm): overlapping atoms or an exploding "
                                          "trajectory -- minimise the structure first (awsem_minimize), reduce the time step, or use the fp64 mode");
            throw std::string("pair list overflow on the device (internal error: list capacity below the candidate pair count)");
        }
    }
    void record_ring_ms() {   // mean over the asynchronous launches since the last synchronisation (the stream is idle)
        float ms = 0;
        double sum = 0;
        for (int k = 0; k < npend_; k++) { HIP_TRY(hipEventElapsedTime(&ms, ring0_[k], ring1_[k])); sum += ms; }
        if (npend_ > 0) last_ms = sum / npend_;
        npend_ = 0;
        timing_pending_ = false;
    }
    void record_ms() {        // synchronous operations: ev0_ / ev1_ bracket the operation that just completed
        float ms = 0;
        npend_ = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev0_, ev1_));
        last_ms = ms;
        timing_pending_ = false;
    }
    // AoS double [natoms][3] on the host  <->  SoA real [3][natoms] per replica on the device
    void set_state(real* dst, int rep, const double* src) {
        HIP_TRY(hipSetDevice(device_));
        std::vector<real> t((size_t)3 * natoms_);
        for (int i = 0; i < natoms_; i++) for (int k = 0; k < 3; k++) t[(size_t)k * natoms_ + in_(i)] = (real)src[3 * i + k];
        synchronize();   // pending launches are resolved (status checked, a cell-path replay done) before the state changes under them
        int r0 = rep < 0 ? 0 : rep, r1 = rep < 0 ? nrep_ : rep + 1;
        for (int r = r0; r < r1; r++)
            HIP_TRY(hipMemcpy(dst + (size_t)r * 3 * natoms_, t.data(), t.size() * sizeof(real), hipMemcpyHostToDevice));
    }
    void get_state(const real* src, int rep, double* dst) {
        HIP_TRY(hipSetDevice(device_));
        synchronize();
        std::vector<real> t((size_t)3 * natoms_);
        HIP_TRY(hipMemcpy(t.data(), src + (size_t)rep * 3 * natoms_, t.size() * sizeof(real), hipMemcpyDeviceToHost));
        for (int i = 0; i < natoms_; i++) for (int k = 0; k < 3; k++) dst[3 * i + k] = (double)t[(size_t)k * natoms_ + in_(i)];
    }

    int natoms_, nrep_, device_;
    // replica exchange state (awsem_remd_setup); all of it lives on the device
    int remd_n_ = 0, remd_start_ = 0;
    uint64_t remd_seed_ = 0;
    int remd_mixing_ = 0;   // 0: neighbour swaps of alternating parity; 1: openmmtools' swap-all (awsem_remd_set_mixing)
    double remd_tmax_ = 0;
    const double* remd_ladder_ = nullptr;
    int* remd_slot_ = nullptr;
    const int* remd_where_ = nullptr;
    unsigned long long* remd_counts_ = nullptr;   // attempted[N-1], accepted[N-1]
    double *remd_U_ = nullptr, *remd_scale_ = nullptr, *remd_T_ = nullptr;
    int block_ = 256;
    size_t lds_eval_ = 0, lds_step_ = 0;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    bool timing_pending_ = false;
    // asynchronous launches (awsem_step): a ring of event pairs and pinned temperature buffers, so that the host can
    // enqueue launch after launch without waiting; drained by synchronize()
    static constexpr int NRING = 32;
    hipEvent_t ring0_[NRING] = {}, ring1_[NRING] = {};
    double* ringT_[NRING] = {};
    double* ringS_[NRING] = {};     // staged launches: [stage][replica] blocks (allocated at the first such launch)
    double* dT_stages_ = nullptr;
    int npend_ = 0;
    std::vector<void*> allocs_;
    DevSys<real> dsys_;
    real *pos_ = nullptr, *vel_raw_ = nullptr;   // vel_raw_: through vel(), which settles a pending rescale first
    bool vel_scale_pending_ = false;
    double *dT_ = nullptr, *dE_ = nullptr, *dF_ = nullptr;
    double* dtmp_ = nullptr;   // per-replica doubles for everything that is not a temperature (kinetic energies, velocity factors)
    const HostSystem* host_ = nullptr;   // the system as the kernels see it: the handle's, or hint_ when the atoms are renumbered
    HostSystem hint_;                    // the handle's system in the internal atom order (replica-resident path)
    std::vector<int> to_int_, to_orig_;  // caller's atom index <-> internal (empty: identical)
    int in_(int i) const { return to_int_.empty() ? i : to_int_[i]; }
    int* dstatus_ = nullptr;
    int* dmin_it_ = nullptr;
    double* dmin_rms_ = nullptr;
    unsigned* dcounts_ = nullptr;
    std::vector<GaussD<real>> hgauss_;   // host copy of the bond-sum descriptors (DevSys::gauss points to device memory)
    struct ListParam { double cutoff, margin; };
    std::vector<ListParam> list_params_;   // per pair list, in the order contact, excl, beta, pap
    bool skin_fixed_ = false;
    // The best skin depends on how fast the particles move and how many pairs a shell of that width adds: measured optima
    // for 1r69 are 0.22-0.25 nm at 800 K (extended chains) and 0.10 nm at 300 K (folded: 33.3 M replica-steps/s against
    // 30.0 M with 0.2).  The lists are rebuilt at the start of every launch anyway, so the skin simply follows the hottest
    // replica of the launch; AWSEM_SKIN pins it.
    void apply_skin(const double* T) {
        if (cell_mode_ || skin_fixed_ || list_params_.empty()) return;
        double tmax = T ? 0 : remd_tmax_;   // null: a replica-exchange launch, any replica may hold the hottest rung
        for (int r = 0; T && r < nrep_; r++) tmax = std::max(tmax, T[r]);
        static const double sa = getenv("AWSEM_SKIN_A") ? atof(getenv("AWSEM_SKIN_A")) : -0.0125, sb = getenv("AWSEM_SKIN_B") ? atof(getenv("AWSEM_SKIN_B")) : 0.000375;   // tuning knobs
        double skin = std::min(0.25, std::max(0.07, sa + sb * tmax));
        // Once the lists' lengths are known (every 8th launch asks for their mean, awsem_list_mean_kernel; the answer is taken up
        // exactly AWSEM_LMEAN_LAG launches later -- waiting for it if need be, which by then costs nothing: the launches in between are
        // still queued -- so the skin schedule, and with it a run's timing, does not depend on how far the host runs ahead of the GPU)
        // the skin follows what the optimum depends on: rebuilds cost ~ N^2 sqrt(T) / skin per step, walks
        // ~ L (r + skin)^3, so skin_opt ~ N T^(1/4) / sqrt(L).  Calibrated on 1r69 (63 sites) in eight states -- the default schedule and
        // one five times slower, 650 ... 300 K: the constant came out within 10 % in all of them -- where a rule in T alone is right
        // for one schedule only (equilibrated chains at 500 K want 0.13 nm, the default schedule's 0.19).
        if (skin_adapt_ && h_lmean_ && lmean_due_ >= 0 && launch_ >= lmean_due_) {
            HIP_TRY(hipEventSynchronize(lmean_ev_));
            lmean_due_ = -1;
            const double m = 1.3, r_eff = 0.8;   // the lists' mean cutoff and the mean factor on the skin (virtual-site margins)
            const double f = (r_eff + 0.1 * m) / (r_eff + lmean_skin_req_ * m);
            lhat_ = h_lmean_[0] * f * f * f;      // listed pairs per replica, referred to a skin of 0.1 nm
        }
        if (skin_adapt_ && lhat_ > 0) {
            static const double c = getenv("AWSEM_SKIN_C") ? atof(getenv("AWSEM_SKIN_C")) : 0.91 / 63.0;
            const double want = std::min(0.25, std::max(0.07, c * skin_sites_ * std::sqrt(std::sqrt(std::max(tmax, 1.0))) / std::sqrt(lhat_)));
            skin = want > skin_ && want < skin_ + 0.01 ? skin_ : want;   // a growing skin costs a rebuild of everything: only for a real change
        }
        if (std::fabs(skin - skin_) < 2e-3) return;
        if (getenv("AWSEM_INFO")) fprintf(stderr, "[awsem] skin %.3f -> %.3f nm (T %.0f K, %.0f listed pairs per replica at 0.1 nm)\n", skin_, skin, tmax, lhat_);
        skin_ = skin;
        DevSys<real>& S = dsys_;
        size_t k = 0;
        auto rc2 = [&]() { const ListParam& p = list_params_[k++]; const double r = p.cutoff + p.margin * skin_; return (real)(r * r); };
        for (int t = 0; t < S.n_contact; t++) S.contact[t].list_rc2 = rc2();
        for (int t = 0; t < S.n_excl; t++) S.excl[t].list_rc2 = rc2();
        for (int t = 0; t < S.n_beta; t++) S.beta[t].list_rc2 = rc2();
        for (int t = 0; t < S.n_pap; t++) S.pap[t].list_rc2 = rc2();
        S.skin_half2 = (real)(0.25 * skin_ * skin_ * 0.999);
    }
    bool skin_adapt_ = !(getenv("AWSEM_SKIN_ADAPT") && atoi(getenv("AWSEM_SKIN_ADAPT")) == 0);
    double* h_lmean_ = nullptr;   // pinned: {mean listed pairs per replica, tag of the request}
    long long lmean_tag_ = 0;
    double lmean_skin_req_ = 0, lhat_ = 0, skin_sites_ = 63;
    static constexpr int AWSEM_LMEAN_LAG = 4;
    int lmean_due_ = -1;          // launch number that consumes the pending answer (-1: none pending)
    hipEvent_t lmean_ev_ = nullptr;
    double skin_ = 0.2;    // nm; AWSEM_SKIN overrides (0.15-0.3 measured: 0.2 and above rebuild rarely enough on the hot, extended stage of an annealing run)
    size_t lds_list_ = 0, lds_max_ = 0, n_cu_ = 256;
    int launch_ = 0;        // step launches so far (parity of the pacing records)
    // A step launch goes on with the pair lists its predecessor left (DevSys::list_ref / list_counts) unless something has touched the
    // positions since (every place that sets cell_stale_), or the skin has grown: lists built with a larger skin stay valid under a
    // smaller one (the rebuild test only gets stricter), not the other way round.  Energy / force / neighbour queries build their lists
    // in a buffer of their own (lists_eval_), so a reporter between two launches costs the dynamics nothing.  AWSEM_LIST_REUSE=0: off.
    bool lists_valid_ = false, list_reuse_ = true, query_reuse_ = true;
    int step_lean_ = 0;   // which instantiation of the step kernel runs (evaluate_terms: LEAN)
    // every launch of the evaluation kernel: the same instantiation as the step kernel's (one code path per handle, whoever asks)
    template <bool ENERGY, class... A> void launch_eval(dim3 grid, hipStream_t st, const DevSys<real>& S, A... args) {
        switch (step_lean_) {
        case 1: hipLaunchKernelGGL((awsem_eval_kernel<ENERGY, real, 1>), grid, dim3(block_), lds_eval_, st, S, args...); break;
        case 2: hipLaunchKernelGGL((awsem_eval_kernel<ENERGY, real, 2>), grid, dim3(block_), lds_eval_, st, S, args...); break;
        case 3: hipLaunchKernelGGL((awsem_eval_kernel<ENERGY, real, 3>), grid, dim3(block_), lds_eval_, st, S, args...); break;
        default: hipLaunchKernelGGL((awsem_eval_kernel<ENERGY, real, 0>), grid, dim3(block_), lds_eval_, st, S, args...); break;
        }
    }
    std::mutex capture_mu_;   // graph capture on stream_ (cell_step) against event waits from another thread (snapshot_end)
    double lists_skin_ = 0;
    unsigned* lists_eval_ = nullptr;
    // Energy / force queries walk the pair lists the last step launch left in device memory when those are still the lists of the stored
    // positions (nothing set from outside since, no minimisation, same skin): awsem_eval_kernel checks the displacement per replica.
    bool query_reuse() const { return list_reuse_ && query_reuse_ && lists_valid_ && !cell_mode_; }
    DevSys<real> eval_sys() const { DevSys<real> S = dsys_; if (lists_eval_) S.lists = lists_eval_; return S; }
    int place_every_ = 16;  // launches between two placements of the replicas on the CUs (0: never), AWSEM_PLACE_EVERY
    bool step_ok_ = false;

    // ---- cell-list path (awsem_cell.h): systems that do not fit one CU's LDS
    void cell_setup(DevSys<real>& S, const HostSystem& H);
    void cell_alloc_lists();
    void cell_pre(bool step);
    void cell_refresh(const CellSys<real>& C);
    void cell_forces_pass(bool energy, unsigned mask, const CellSys<real>* Cp = nullptr, bool window = true, int rows_cur = -1, int rows_force = 0);
    void cell_evaluate(bool energy, unsigned mask, int redistribute, double* e, double* f);
    void cell_step(int nsteps, const double* T, double dt, double friction, uint64_t seed, int64_t step0);
    // A neighbour row that overflows in the middle of a trajectory (the chains became denser than the rows were sized for) drops
    // pairs; the launches since the last synchronisation are then REPLAYED from a snapshot of (x, v) with longer rows.  The
    // random numbers are a function of (seed, replica, step, atom), so the replay is the trajectory larger rows would have given.
    struct PendingStep { int nsteps; std::vector<double> T; double dt, friction; uint64_t seed; int64_t step0; };
    std::vector<PendingStep> cell_pending_;
    real *cell_snap_pos_ = nullptr, *cell_snap_vel_ = nullptr;
    bool cell_replaying_ = false;
    void cell_replay_if_overflowed();
    void cell_one_step(double dt, double friction, uint64_t seed, int offset);
    void cell_minimize(double tolerance, int max_iterations, int32_t* iterations, double* rms);
    struct CellGraph { hipGraphExec_t exec = nullptr; double dt = 0, friction = 0; uint64_t seed = 0; long long gen = -1; };
    long long cell_rows_gen_ = 0;          // bumped whenever the neighbour rows are re-allocated (captured graphs hold their addresses)
    HbRows cell_hb_beta_[AWSEM_MAX_BETA] = {}, cell_hb_pap_[AWSEM_MAX_PAP] = {};   // donor rows of the hydrogen-bond terms
    real cell_hb_rc2_beta_[AWSEM_MAX_BETA] = {}, cell_hb_rc2_pap_[AWSEM_MAX_PAP] = {};
    HbTerms<real> cell_hb_ = {};           // device-resident descriptor / rows / radius tables of all hydrogen-bond terms
    int cell_hb_maxacc_ = 0, cell_hb_maxblocks_ = 1;
    // before the hydrogen-bond rows are built or walked: the coordinates as packed records (every time), the donors' bounding spheres
    // (on a rebuild step) -- cell_hb_prepare_kernel
    void cell_hb_prepare(const CellSys<real>& C, int cur, int force) {
        const long long want = std::max((long long)cell_hb_maxblocks_ * AWSEM_HB_BLOCK, (long long)natoms_ / 4);
        const dim3 g((unsigned)std::max(1LL, std::min((want + 255) / 256, 4096LL)), nrep_, cell_hb_.n_beta + cell_hb_.n_pap + 1);
        hipLaunchKernelGGL(cell_hb_prepare_kernel<real>, g, dim3(256), 0, stream_, cell_hb_, C, pos_, cur, force);
    }
    void cell_hb_sync();                   // re-uploads the rows and radius tables after a change on the host
    int dsys_beta_n_ = 0, dsys_pap_n_ = 0, cell_nacc_beta_[AWSEM_MAX_BETA] = {}, cell_nacc_pap_[AWSEM_MAX_PAP] = {},   // known before dsys_ is set
        cell_ndon_beta_[AWSEM_MAX_BETA] = {}, cell_ndon_pap_[AWSEM_MAX_PAP] = {};
    CellGraph cell_graph_[2];
    long long* cell_step_base_ = nullptr;
    int cell_neighbors(int rep, int kind, int index, double skin, int32_t* pairs, int capacity);
    dim3 cell_grid(long long n) const { long long b = (n + 255) / 256; return dim3((unsigned)std::max(1LL, std::min(b, 2048LL)), nrep_); }
    bool cell_mode_ = false, cell_stale_ = true, cell_fused_ = true;
    std::vector<int> vs_dev_site_;   // DevSys::vs[i].x on the host
    // the sort's workgroups wait for one another: 2 x 32 x replicas of them (256 threads, 2 KB of LDS) must be resident at once -- a quarter of what the device holds
    bool cell_sort_ok() const { return cell_fused_ && (long long)2 * AWSEM_SORT_G * nrep_ <= 2LL * (long long)n_cu_; }
    CellSys<real> cell_;
    const CellJob* cell_jobs_ = nullptr;
    int cell_njobs_ = 0, cell_per_site_ = 96, cell_cur_ = 0;
    double cell_rc_contact_ = 0, cell_rc_excl_ = 0;
    std::vector<int> excl_sorted_atom_, excl_sorted_cls_;   // class-sorted site table of excl term 0
    std::vector<int> cell_wstart_, cell_nlocal_;            // per local bonded term (harm, chi, rama, helical, fm, memb order): window starts, local item count
    const int* cell_wstart_dev_ = nullptr;
    int cell_nwin_ = 0, cell_nslots_ = 0;
    int cell_ncvjobs_ = 0;
    const CellJob *cell_cv_sum_jobs_ = nullptr, *cell_cv_apply_jobs_ = nullptr;
};

// Order static items (bonds, dihedrals, virtual sites ...) so that the 64 items one wave handles in one
// instruction touch pairwise different atoms wherever the topology allows it: the fixed-point force accumulation is
// an LDS atomic, and lanes of one atomic instruction that hit the same address serialise (measured ~4 clk per extra
// lane; a factory's natural order -- all bonds of atom i next to each other -- gives 10-fold conflicts).
// Returns a permutation: position k of the new order holds item perm[k].  Aligned groups of 64 are filled greedily;
// a group that cannot be filled with multiplicity m is topped up with multiplicity m+1.
static std::vector<int> spread_order(int n, int width, const int* atoms, int natoms) {
    std::vector<int> perm;
    perm.reserve(n);
    std::vector<int> remaining(n), next;
    for (int i = 0; i < n; i++) remaining[i] = i;
    std::vector<int> used(natoms, 0);
    while (!remaining.empty()) {
        int in_group = 0;
        std::vector<int> touched;
        for (int m = 1; in_group < 64 && !remaining.empty(); m++) {
            next.clear();
            for (int it : remaining) {
                bool ok = in_group < 64;
                for (int k = 0; ok && k < width; k++) { int a = atoms[(size_t)it * width + k]; ok = a < 0 || used[a] < m; }
                if (ok) {
                    perm.push_back(it); in_group++;
                    for (int k = 0; k < width; k++) { int a = atoms[(size_t)it * width + k]; if (a >= 0) { used[a]++; touched.push_back(a); } }
                } else next.push_back(it);
            }
            remaining.swap(next);
        }
        for (int a : touched) used[a] = 0;
    }
    return perm;
}
template <class T> static std::vector<T> permuted(const std::vector<T>& v, const std::vector<int>& perm, int width = 1) {
    std::vector<T> out(v.size());
    for (size_t k = 0; k < perm.size(); k++) for (int w = 0; w < width; w++) out[k * width + w] = v[(size_t)perm[k] * width + w];
    return out;
}

// Largest band of diagonals q - p around 0 on which the nres x nres table t[q][p] is identically zero:
// t[q][p] == 0 for all lo < q - p < hi.
static void zero_band(const std::vector<double>& t, int nres, int* lo, int* hi) {
    auto diag_zero = [&](int dq) {
        for (int q = std::max(0, dq); q < std::min(nres, nres + dq); q++)
            if (t[(size_t)q * nres + (q - dq)] != 0.0) return false;
        return true;
    };
    if (!diag_zero(0)) { *lo = 0; *hi = 0; return; }   // empty band: the test (dq <= 0 || dq >= 0) is always true
    int h = 1, l = -1;
    while (h < nres && diag_zero(h)) h++;
    while (l > -nres && diag_zero(l)) l--;
    *lo = l; *hi = h;
}

// The same band for a table given in functional form (HFn): beta kinds vanish for |q - p| < 4; the P-AP gammas of a single
// chain vanish below q - p = 13 (pap_1) / 9 (pap_2); with several chains a pair of neighbouring residues can straddle a chain
// boundary, so only the diagonal itself is known to be zero.
static void fn_band(const HFn& f, int nres, int* lo, int* hi) {
    if (f.kind <= 3) { *lo = -4; *hi = 4; return; }
    bool one_chain = true;
    for (int i = 1; i < nres; i++) one_chain = one_chain && f.info[4 * i + 1] == f.info[1];
    if (one_chain) { *lo = -nres; *hi = f.kind == 4 ? 13 : 9; }
    else { *lo = -1; *hi = 1; }
}
// one bit per table entry: set where t[q][p] != 0 (the list builders test it instead of loading the table)
static std::vector<unsigned> nonzero_mask(const std::vector<double>& t, int nres) {
    std::vector<unsigned> m(((size_t)nres * nres + 31) / 32, 0u);
    for (size_t i = 0; i < (size_t)nres * nres; i++) if (t[i] != 0.0) m[i >> 5] |= 1u << (i & 31);
    return m;
}

// Internal atom order of the replica-resident path.  A factory numbers atoms residue by residue (N, H, CA, C, O, CB: a
// stride of ~6), so the 64 lanes of an instruction that touches "the CB of 64 consecutive residues" hit only a quarter of
// the LDS banks with the 16-byte position records (and the pair lists are built exactly so that their lanes do walk
// consecutive sites).  Internally the atoms are therefore grouped by ROLE -- excluded-volume class, and for virtual sites
// the weight triple that places them -- and kept in the caller's order within a role: CA block, CB block, O block, N, H,
// C blocks.  Consecutive residues are then consecutive LDS records.  The caller never sees this order: positions,
// velocities, forces and neighbour queries are mapped at the API boundary, and the random streams are keyed by the
// caller's atom index.
static std::vector<int> role_order(const HostSystem& H) {
    const int n = H.natoms;
    std::vector<long long> key(n, 0);
    if (!H.excl.empty()) for (size_t i = 0; i < H.excl[0].atom.size(); i++) key[H.excl[0].atom[i]] = 1 + H.excl[0].cls[i];
    std::vector<std::array<long long, 3>> seen;   // distinct weight triples, in order of first appearance
    for (size_t v = 0; v < H.vs_site.size(); v++) {
        std::array<long long, 3> w = {llround(H.vs_w[3 * v] * 1e6), llround(H.vs_w[3 * v + 1] * 1e6), llround(H.vs_w[3 * v + 2] * 1e6)};
        size_t id = 0;
        while (id < seen.size() && seen[id] != w) id++;
        if (id == seen.size()) seen.push_back(w);
        key[H.vs_site[v]] = 1000 + (long long)id;
    }
    std::vector<int> order(n), to(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    for (int k = 0; k < n; k++) to[order[k]] = k;
    return to;
}
static HostSystem renumbered(const HostSystem& H, const std::vector<int>& to) {
    HostSystem R = H;
    auto m = [&](std::vector<int>& v) { for (int& a : v) if (a >= 0) a = to[a]; };
    for (int i = 0; i < H.natoms; i++) R.mass[to[i]] = H.mass[i];
    m(R.vs_site); m(R.vs_par);
    for (auto& t : R.harm) m(t.pairs);
    for (auto& t : R.chi) m(t.atoms);
    for (auto& t : R.rama) m(t.atoms);
    for (auto& t : R.excl) { m(t.atom); m(t.excl); }
    for (auto& t : R.contact) m(t.atom);
    for (auto& t : R.beta) { m(t.acc_atoms); m(t.don_atoms); }
    for (auto& t : R.pap) { m(t.acc_atoms); m(t.don_atoms); }
    for (auto& t : R.helical) m(t.atoms);
    for (auto& t : R.fm) m(t.pairs);
    for (auto& t : R.memb) m(t.atoms);
    for (auto& t : R.dh) m(t.pairs);
    for (auto& t : R.gauss) m(t.pairs);
    return R;
}

template <class real> void Engine<real>::finalize(const HostSystem& H0) {
    HIP_TRY(hipSetDevice(device_));
    if (device_ < AWSEM_MAX_DEVICES && !g_noalloc_ptr[device_].load(std::memory_order_acquire)) {   // set-up: this device's copy of the static scratch array
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_noalloc_ptr[device_].store(p, std::memory_order_release);
    }
    host_ = &H0;
    DevSys<real> S;
    memset(&S, 0, sizeof S);
    S.natoms = natoms_;
    {   // which path: replica-resident (one replica fits the LDS of a CU) or cell lists
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device_));
        size_t lds_max = prop.sharedMemPerBlock ? prop.sharedMemPerBlock : 65536;
        if (prop.maxSharedMemoryPerMultiProcessor > lds_max) lds_max = prop.maxSharedMemoryPerMultiProcessor;
        lds_max_ = lds_max;
        n_cu_ = std::max(1, prop.multiProcessorCount);
        int max_sites = 0;
        for (auto& h : H0.contact) max_sites = std::max(max_sites, (int)h.atom.size());
        const char* force = getenv("AWSEM_FORCE_CELL");
        bool zdep = false;
        for (auto& h : H0.contact) zdep = zdep || (h.flags & AWSEM_CONTACT_ZDEP);
        // sized by what DYNAMICS needs (velocities of the massive particles and the list reference positions on top of the
        // evaluation's image): a system that fits for energies only would otherwise be unable to step or minimise
        int nmassive = 0, ncv = 0;
        for (double m : H0.mass) nmassive += m > 0 ? 1 : 0;
        for (auto& h : H0.gauss) ncv += h.cv ? 1 : 0;
        cell_mode_ = lds_bytes<real>(LdsShape{natoms_, max_sites, (int)H0.contact.size(), nmassive, zdep ? 2400 : 1200, ncv}, true) > lds_max || (force && atoi(force) != 0);
    }
    to_int_.clear(); to_orig_.clear();
    if (!cell_mode_ && !(getenv("AWSEM_NO_REORDER") && atoi(getenv("AWSEM_NO_REORDER")) != 0)) {   // the cell path's windows want the chain order
        to_int_ = role_order(H0);
        bool identity = true;
        for (int i = 0; i < natoms_; i++) identity = identity && to_int_[i] == i;
        if (identity) to_int_.clear();
        else {
            to_orig_.assign(natoms_, 0);
            for (int i = 0; i < natoms_; i++) to_orig_[to_int_[i]] = i;
            hint_ = renumbered(H0, to_int_);
            host_ = &hint_;
            S.orig = upload(to_orig_);
        }
    }
    const HostSystem& H = *host_;
    // Item order of the static lists.  Replica-resident path: conflict-aware (spread_order).  Cell-list path: by window of
    // AWSEM_WIN atoms (the window kernel accumulates a window's items in LDS); items spanning more than a window's LDS
    // span go last and take the direct path.
    auto item_order = [&](int n, int width, const int* atoms) {
        if (!cell_mode_) return spread_order(n, width, atoms, natoms_);
        const int nwin = (natoms_ + AWSEM_WIN - 1) / AWSEM_WIN;
        std::vector<int> win(n), perm(n);
        for (int i = 0; i < n; i++) {
            int lo = natoms_, hi = -1;
            for (int k = 0; k < width; k++) { const int a = atoms[(size_t)i * width + k]; if (a >= 0) { lo = std::min(lo, a); hi = std::max(hi, a); } }
            const int w = lo / AWSEM_WIN;
            win[i] = (hi >= 0 && hi < w * AWSEM_WIN + AWSEM_WIN_SPAN) ? w : nwin;   // nwin = not local
            perm[i] = i;
        }
        std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return win[x] < win[y]; });
        std::vector<int> start(nwin + 2, 0);
        for (int i = 0; i < n; i++) start[win[i] + 1]++;
        for (int w = 0; w <= nwin; w++) start[w + 1] += start[w];
        cell_wstart_.insert(cell_wstart_.end(), start.begin(), start.begin() + nwin + 1);   // start[nwin] = number of local items
        cell_nlocal_.push_back(start[nwin]);
        return perm;
    };
    cell_wstart_.clear(); cell_nlocal_.clear();
    // particles
    {
        std::vector<real> im(natoms_), sim(natoms_);
        for (int i = 0; i < natoms_; i++) {
            im[i] = H.mass[i] > 0 ? (real)(1.0 / H.mass[i]) : (real)0;
            sim[i] = H.mass[i] > 0 ? (real)std::sqrt(1.0 / H.mass[i]) : (real)0;
        }
        S.inv_mass = upload(im);
        S.sqrt_inv_mass = upload(sim);
        double mtot = 0;
        for (int i = 0; i < natoms_; i++) mtot += H.mass[i];
        S.remove_cm = H.remove_cm && mtot > 0 ? 1 : 0;
        S.inv_total_mass = mtot > 0 ? (real)(1.0 / mtot) : (real)0;
        S.prio = getenv("AWSEM_PRIO") ? atoi(getenv("AWSEM_PRIO")) : 3;   // 0 switches the rotation off, 1 is the plain rotation (A/B runs)
        if (S.prio == 3) S.prio |= ((int)lround(16.0 * (getenv("AWSEM_PACE_GAIN") ? atof(getenv("AWSEM_PACE_GAIN")) : 10.0)) & 0xffff) << 8
                                 | ((getenv("AWSEM_PACE_BOOST") ? atoi(getenv("AWSEM_PACE_BOOST")) : 4) & 15) << 24;
        S.pace_group = (int)n_cu_;
        std::vector<MassiveItem<real>> mv;
        for (int i = 0; i < natoms_; i++) if (H.mass[i] > 0) mv.push_back(MassiveItem<real>{i, im[i], sim[i], to_orig_.empty() ? i : to_orig_[i]});
        S.massive = upload(mv);
        S.nmassive = (int)mv.size();
        S.nvs = (int)H.vs_site.size();
        {   // gather table: which virtual sites hand their force to each massive particle, and with which weight
            std::vector<GatherItem<real>> gt(mv.size());
            std::vector<int> slot_of(natoms_, -1), fill(mv.size(), 0), is_vs(natoms_, 0);
            for (size_t k = 0; k < mv.size(); k++) { slot_of[mv[k].atom] = (int)k; for (int q = 0; q < AWSEM_MAX_GATHER; q++) { gt[k].vs[q] = -1; gt[k].w[q] = (real)0; } }
            for (int i = 0; i < S.nvs; i++) {
                is_vs[H.vs_site[i]] = 1;
                for (int c = 0; c < 3; c++) {
                    const int par = H.vs_par[3 * i + c];
                    if (H.mass[par] <= 0) throw std::string("virtual sites: a parent particle must have a mass (sites of sites are not supported)");
                    const int k = slot_of[par];
                    if (fill[k] >= AWSEM_MAX_GATHER) throw std::string("virtual sites: more than 8 sites depend on one particle");
                    gt[k].vs[fill[k]] = H.vs_site[i]; gt[k].w[fill[k]] = (real)H.vs_w[3 * i + c]; fill[k]++;
                }
            }
            S.gather = upload(gt);
            std::vector<int> fixed;
            for (int i = 0; i < natoms_; i++) if (H.mass[i] <= 0 && !is_vs[i]) fixed.push_back(i);
            S.nfixed = (int)fixed.size();
            S.fixed = upload(fixed);
        }
        const std::vector<int> perm = spread_order(S.nvs, 3, H.vs_par.data(), natoms_);   // force redistribution adds to the parents
        std::vector<Int4> vs(S.nvs);
        for (int i = 0; i < S.nvs; i++) { int o = perm[i]; vs[i] = Int4{H.vs_site[o], H.vs_par[3 * o], H.vs_par[3 * o + 1], H.vs_par[3 * o + 2]}; }
        S.vs = upload(vs);
        vs_dev_site_.resize(S.nvs);
        for (int i = 0; i < S.nvs; i++) vs_dev_site_[i] = vs[i].x;
        S.vs_w = upload_real(permuted(H.vs_w, perm, 3));
    }
    {   // harmonic
        std::vector<HarmD<real>> D;
        // Replica-resident path: the harmonic terms of one force group (con_term and chain_term of the default set-up) walk as ONE item
        // list -- an item carries its own r0 and k, and every pass a step makes costs its set-up whether or not it finds work
        // (AWSEM_HARM_MERGE=0: term by term).  The cell-list path keeps them apart (its window tables are per term).
        const bool merge = !cell_mode_ && !(getenv("AWSEM_HARM_MERGE") && atoi(getenv("AWSEM_HARM_MERGE")) == 0);
        std::vector<char> done(H.harm.size(), 0);
        for (size_t t = 0; t < H.harm.size(); t++) {
            if (done[t]) continue;
            std::vector<int> pairs; std::vector<double> r0, k;
            for (size_t u = t; u < H.harm.size(); u++) {
                if (done[u] || (u != t && !(merge && H.harm[u].group == H.harm[t].group))) continue;
                done[u] = 1;
                const auto& h = H.harm[u];
                pairs.insert(pairs.end(), h.pairs.begin(), h.pairs.end()); r0.insert(r0.end(), h.r0.begin(), h.r0.end()); k.insert(k.end(), h.k.begin(), h.k.end());
            }
            std::vector<HarmItem<real>> it(r0.size());
            const std::vector<int> perm = item_order((int)it.size(), 2, pairs.data());
            for (size_t b = 0; b < it.size(); b++) { int o = perm[b]; it[b] = HarmItem<real>{pairs[2 * o], pairs[2 * o + 1], (real)r0[o], (real)k[o]}; }
            D.push_back(HarmD<real>{H.harm[t].group, (int)it.size(), upload(it)});
        }
        if (D.size() > AWSEM_MAX_HARM) throw std::string("too many harm terms: at most " + std::to_string(AWSEM_MAX_HARM) + " per system");
        S.n_harm = (int)D.size(); for (size_t t = 0; t < D.size(); t++) S.harm[t] = D[t];
    }
    {   // chi
        std::vector<ChiD<real>> D;
        for (auto& h : H.chi) {
            std::vector<Int4> a(h.atoms.size() / 4);
            const std::vector<int> perm = item_order((int)a.size(), 4, h.atoms.data());
            for (size_t b = 0; b < a.size(); b++) { int o = perm[b]; a[b] = Int4{h.atoms[4 * o], h.atoms[4 * o + 1], h.atoms[4 * o + 2], h.atoms[4 * o + 3]}; }
            D.push_back(ChiD<real>{h.group, (int)a.size(), upload(a), (real)h.k, (real)h.chi0});
        }
        if (D.size() > AWSEM_MAX_CHI) throw std::string("too many chi terms: at most " + std::to_string(AWSEM_MAX_CHI) + " per system");
        S.n_chi = (int)D.size(); for (size_t t = 0; t < D.size(); t++) S.chi[t] = D[t];
    }
    {   // rama
        std::vector<RamaD<real>> D;
        for (auto& h : H.rama) {
            RamaD<real> d;
            memset(&d, 0, sizeof d);
            d.group = h.group; d.n = (int)(h.atoms.size() / 5); d.nwell = h.nwell;
            std::vector<Int5> a(d.n);
            const std::vector<int> perm = item_order(d.n, 5, h.atoms.data());
            for (int b = 0; b < d.n; b++) for (int k = 0; k < 5; k++) a[b].a[k] = h.atoms[5 * perm[b] + k];
            d.atoms = upload(a);
            std::vector<real> bw((size_t)d.n * h.nwell);
            for (int b = 0; b < d.n; b++) for (int w = 0; w < h.nwell; w++)
                bw[(size_t)b * h.nwell + w] = (real)(-h.k * h.wp[6 * w] * h.bw[(size_t)perm[b] * h.nwell + w]);
            d.bw = upload(bw);
            for (int w = 0; w < h.nwell; w++) {
                const double* p = &h.wp[6 * w];
                d.sigma[w] = (real)p[1]; d.omega_phi[w] = (real)p[2]; d.omega_psi[w] = (real)p[4];
                d.cphi0[w] = (real)std::cos(p[3]); d.sphi0[w] = (real)std::sin(p[3]);
                d.cpsi0[w] = (real)std::cos(p[5]); d.spsi0[w] = (real)std::sin(p[5]);
            }
            D.push_back(d);
        }
        if (D.size() > AWSEM_MAX_RAMA) throw std::string("too many rama terms: at most " + std::to_string(AWSEM_MAX_RAMA) + " per system");
        S.n_rama = (int)D.size(); for (size_t t = 0; t < D.size(); t++) S.rama[t] = D[t];
        // replica-resident path: all terms of one group walk as one item list (AWSEM_RAMA_MERGE=0: term by term, for A/B runs)
        bool merge = !cell_mode_ && D.size() >= 2 && !(getenv("AWSEM_RAMA_MERGE") && atoi(getenv("AWSEM_RAMA_MERGE")) == 0);
        for (auto& d : D) if (d.group != D[0].group) merge = false;
        if (merge) {
            RamaMergedD<real>& m = S.ramam;
            m.group = D[0].group; m.nset = (int)D.size(); m.nwell = 0; m.n = 0;
            for (auto& h : H.rama) { m.nwell = std::max(m.nwell, h.nwell); m.n += (int)(h.atoms.size() / 5); }
            std::vector<int> atoms; std::vector<int> set_of; std::vector<double> w;   // concatenated in the caller's order
            for (size_t t = 0; t < H.rama.size(); t++) {
                const auto& h = H.rama[t];
                const int n = (int)(h.atoms.size() / 5);
                atoms.insert(atoms.end(), h.atoms.begin(), h.atoms.end());
                for (int b = 0; b < n; b++) {
                    set_of.push_back((int)t);
                    for (int q = 0; q < m.nwell; q++) w.push_back(q < h.nwell ? -h.k * h.wp[6 * q] * h.bw[(size_t)b * h.nwell + q] : 0.0);
                }
            }
            const std::vector<int> perm = item_order(m.n, 5, atoms.data());
            std::vector<RamaItem> it(m.n);
            std::vector<real> bw((size_t)m.n * m.nwell);
            for (int b = 0; b < m.n; b++) {
                const int o = perm[b];
                for (int k = 0; k < 5; k++) it[b].a[k] = atoms[5 * o + k];
                it[b].set = set_of[o];
                for (int q = 0; q < m.nwell; q++) bw[(size_t)b * m.nwell + q] = (real)w[(size_t)o * m.nwell + q];
            }
            std::vector<real> par((size_t)m.nset * m.nwell * 8, (real)0);
            for (int t = 0; t < m.nset; t++) for (int q = 0; q < D[t].nwell; q++) {
                real* p = &par[((size_t)t * m.nwell + q) * 8];
                p[0] = D[t].sigma[q]; p[1] = D[t].omega_phi[q]; p[2] = D[t].omega_psi[q];
                p[3] = D[t].cphi0[q]; p[4] = D[t].sphi0[q]; p[5] = D[t].cpsi0[q]; p[6] = D[t].spsi0[q];
            }
            m.items = upload(it); m.bw = upload(bw); m.params = upload(par);
        }
    }
    {   // excl: sort sites by class, per-site partner lists
        std::vector<ExclD<real>> D;
        for (auto& h : H.excl) {
            ExclD<real> d;
            memset(&d, 0, sizeof d);
            d.group = h.group; d.k = (real)h.k; d.r0 = (real)h.r0;
            int ncls = 0;
            for (int c : h.cls) ncls = std::max(ncls, c + 1);
            d.nclass = ncls; d.nsites = (int)h.atom.size();
            std::vector<int> atom;
            for (int c = 0; c < ncls; c++) {
                d.class_start[c] = (int)atom.size();
                for (size_t i = 0; i < h.atom.size(); i++) if (h.cls[i] == c) atom.push_back(h.atom[i]);
            }
            d.class_start[ncls] = (int)atom.size();
            std::vector<Int4> partners(atom.size(), Int4{-1, -1, -1, -1});
            std::vector<int> count(atom.size(), 0);
            std::vector<std::vector<int>> slots_of(natoms_);   // an atom may be listed as a site more than once
            for (size_t s = 0; s < atom.size(); s++) slots_of[atom[s]].push_back((int)s);
            auto add_partner = [&](int a, int b) {
                for (int s : slots_of[a]) {
                    int* q = &partners[s].x;
                    if (count[s] >= 4) throw std::string("excl: more than 4 excluded partners for one atom");
                    q[count[s]++] = b;
                }
            };
            for (size_t q = 0; q + 1 < h.excl.size(); q += 2) { add_partner(h.excl[q], h.excl[q + 1]); add_partner(h.excl[q + 1], h.excl[q]); }
            d.atom = upload(atom); d.partners = upload(partners);
            if (D.empty()) {
                excl_sorted_atom_ = atom;
                excl_sorted_cls_.assign(atom.size(), 0);
                for (int c = 0; c < ncls; c++) for (int q = d.class_start[c]; q < d.class_start[c + 1]; q++) excl_sorted_cls_[q] = c;
            }
            D.push_back(d);
        }
        if (D.size() > AWSEM_MAX_EXCL) throw std::string("too many excl terms: at most " + std::to_string(AWSEM_MAX_EXCL) + " per system");
        S.n_excl = (int)D.size(); for (size_t t = 0; t < D.size(); t++) S.excl[t] = D[t];
    }
    {   // contact
        std::vector<ContactD<real>> D;
        for (auto& h : H.contact) {
            ContactD<real> d;
            memset(&d, 0, sizeof d);
            d.group = h.group; d.nsites = (int)h.atom.size(); d.env = h.env; d.min_seq_sep = h.min_seq_sep; d.flags = h.flags;
            d.has_burial = h.has_burial;
            d.atom = upload(h.atom);
            {
                std::vector<int> info(h.atom.size());
                for (size_t i = 0; i < info.size(); i++) {
                    if (h.resid[i] < 0 || h.resid[i] > 0xffff || h.chain[i] < 0 || h.chain[i] > 0x7ff)
                        throw std::string("contact: residue index must be < 65536 and chain index < 2048");
                    info[i] = h.resid[i] | (h.chain[i] << 16) | (h.restype[i] << 27);
                }
                d.info = upload(info);
            }
            d.gamma = upload_real(h.gamma); d.bgamma = upload_real(h.bgamma);
            const double* p = h.p;
            d.k = (real)p[AWSEM_CP_K]; d.rmin = (real)p[AWSEM_CP_RMIN]; d.rmax = (real)p[AWSEM_CP_RMAX];
            d.rminII = (real)p[AWSEM_CP_RMINII]; d.rmaxII = (real)p[AWSEM_CP_RMAXII];
            d.eta2 = (real)(2 * p[AWSEM_CP_ETA]); d.eta_s2 = (real)(2 * p[AWSEM_CP_ETA_SIGMA]); d.rho0 = (real)p[AWSEM_CP_RHO0];
            double cut = (h.flags & AWSEM_CONTACT_NOCUTOFF) ? 1e15 : p[AWSEM_CP_CUTOFF];
            d.cutoff2 = (real)(cut * cut);
            d.kappa2 = (real)(2 * p[AWSEM_CP_BURIAL_KAPPA]);
            for (int m = 0; m < 3; m++) { d.rho_lo[m] = (real)p[AWSEM_CP_BURIAL_RHO_MIN0 + m]; d.rho_hi[m] = (real)p[AWSEM_CP_BURIAL_RHO_MAX0 + m]; }
            d.eta_sw2 = (real)(2 * p[AWSEM_CP_ETA_SWITCH]); d.z_m = (real)p[AWSEM_CP_Z_M]; d.zc = (real)p[AWSEM_CP_MEMBRANE_CENTER];
            d.kb = (real)p[AWSEM_CP_K_BURIAL];
            S.max_contact_sites = std::max(S.max_contact_sites, d.nsites);
            if (h.flags & AWSEM_CONTACT_ZDEP) S.any_zdep = 1;
            D.push_back(d);
        }
        if (D.size() > AWSEM_MAX_CONTACT) throw std::string("too many contact terms: at most " + std::to_string(AWSEM_MAX_CONTACT) + " per system");
        S.n_contact = (int)D.size(); for (size_t t = 0; t < D.size(); t++) S.contact[t] = D[t];
    }
    // largest |entry| of a lambda / gamma table, dense or in its functional form (which reproduces the dense table to the last bit);
    // -1 for a functional table of more than 4096 residues (the caller then keeps the loose screen)
    auto table_max = [&](const auto& fn, int nres, const std::vector<double>& dense_table) -> double {
        double m = 0;
        if (!fn.kind) { for (double v : dense_table) m = std::max(m, std::fabs(v)); return m; }
        if (nres > 4096) return -1.0;
        std::vector<int> rinfo(nres);
        for (int i = 0; i < nres; i++) {
            const int* q = &fn.info[4 * (size_t)i];
            const bool edge = q[2] != 0;
            const int prev = (i > 0 && !edge) ? fn.info[4 * (size_t)(i - 1)] : 0, next = (i + 1 < nres && !edge) ? fn.info[4 * (size_t)(i + 1)] : 0;
            rinfo[i] = q[0] | (prev << 5) | (next << 10) | (edge ? 1 << 15 : 0) | (q[3] ? 1 << 16 : 0) | (q[1] << 17);
        }
        std::vector<HbFn<double>> one(1);
        HbFn<double>& f = one[0];
        memset(&f, 0, sizeof f);
        f.kind = fn.kind; f.rinfo = rinfo.data(); f.scale = 1.0;
        const double* c = fn.coef.data();
        for (int a = 0; a < 5; a++) for (int b = 0; b < 3; b++) f.cc[a][b] = c[3 * a + b];
        for (int a = 0; a < 3; a++) f.g[a] = c[15 + a];
        for (int a = 0; a < 400; a++) { f.t1[a] = c[18 + a]; f.t2[a] = c[418 + a]; }
        for (int a = 0; a < 20; a++) f.ob[a] = c[818 + a];
        for (int i = 0; i < nres; i++) for (int j = 0; j < nres; j++) m = std::max(m, std::fabs(hb_fn_value(f, i, j)));
        return m;
    };
    {   // beta
        std::vector<BetaD<real>> D;
        // fp32: the exponent X beyond which a pair is not evaluated (and the radius r_ON + sqrt(2X) sigma beyond which it is not listed).
        // Chosen so that what is dropped is EXACTLY zero in the fp32 force accumulators: a force component of such a pair is at most
        // |c| e^-X sqrt(2X) / sigma (c = -k lambda; x e^{-x^2 / 2 sigma^2} / sigma^2 under the constraint that the exponent is >= X),
        // and Q19.12 rounds anything below 2^-13 kJ/mol/nm to zero -- X solves |c|max e^-X sqrt(2X) / sigma_min = 2^-16, a factor 8 inside
        // that.  Round 1 used X = 30 throughout; with the reference's tables X is about 18 and the lists are a third shorter.  The oracle's
        // fp32 list mirror (awsem_oracle_neighbors) computes the same number.  Functional tables of more than 4096 residues: X = 30 as before.
        double beta_X = 30.0;
        {
            double cmax = 0, smin = 1e300;
            bool dense = !H.beta.empty();
            for (auto& h : H.beta) {
                const double m = table_max(h.fn, h.nres, h.lambda);
                if (m < 0) { dense = false; break; }
                cmax = std::max(cmax, std::fabs(h.k) * m); smin = std::min(smin, std::min(h.s_NO, h.s_HO));
            }
            if (dense && cmax > 0 && !(getenv("AWSEM_BETA_SCREEN") && atoi(getenv("AWSEM_BETA_SCREEN")) == 30)) {
                double X = 20.0;
                for (int it = 0; it < 8; it++) X = std::log(cmax * 65536.0 * std::sqrt(2.0 * X) / smin);
                beta_X = std::min(30.0, std::max(8.0, X));
            }
        }
        for (auto& h : H.beta) {
            BetaD<real> d;
            memset(&d, 0, sizeof d);
            d.group = h.group; d.two_theta = h.two_theta; d.nacc = (int)h.acc_res.size(); d.ndon = (int)h.don_res.size(); d.nres = h.nres;
            std::vector<Int4> acc(d.nacc), don(d.ndon);
            for (int i = 0; i < d.nacc; i++) acc[i] = Int4{h.acc_atoms[3 * i], h.acc_atoms[3 * i + 1], h.acc_atoms[3 * i + 2], h.acc_res[i]};
            for (int i = 0; i < d.ndon; i++) don[i] = Int4{h.don_atoms[3 * i], h.don_atoms[3 * i + 1], h.don_atoms[3 * i + 2], h.don_res[i]};
            d.acc = upload(acc); d.don = upload(don);
            if (h.fn.kind) {
                d.fn = upload_fn(h.fn, h.nres, -h.k);
                fn_band(h.fn, h.nres, &d.band_lo, &d.band_hi);
            } else {
                d.lambda = upload_real(h.lambda, -h.k);
                d.nzmask = upload(nonzero_mask(h.lambda, h.nres));
                d.nz_lds = ((size_t)d.nacc + d.ndon + ((size_t)h.nres * h.nres + 31) / 32) * 4 <= (size_t)natoms_ * (sizeof(real) == 4 ? 16 : 24);
                zero_band(h.lambda, h.nres, &d.band_lo, &d.band_hi);
            }
            d.r_ON = (real)h.r_ON; d.r_OH = (real)h.r_OH; d.inv_sNO2 = (real)(1.0 / (h.s_NO * h.s_NO)); d.inv_sHO2 = (real)(1.0 / (h.s_HO * h.s_HO));
            d.cutoff2 = (real)(h.cutoff * h.cutoff);
            if (sizeof(real) == 4) {   // fp32 pre-screen: Gaussian factors below e^-X are dropped (X: beta_screen_exponent above)
                double rs = h.r_ON + std::sqrt(2.0 * beta_X) * h.s_NO;
                d.screen2 = (real)std::min(h.cutoff * h.cutoff, rs * rs);
                d.max_expo = (real)beta_X;
            } else {
                d.screen2 = d.cutoff2;
                d.max_expo = (real)1e300;
            }
            D.push_back(d);
        }
        if (D.size() > AWSEM_MAX_BETA) throw std::string("too many beta terms: at most " + std::to_string(AWSEM_MAX_BETA) + " per system");
        S.n_beta = (int)D.size(); for (size_t t = 0; t < D.size(); t++) S.beta[t] = D[t];
    }
    // The same argument for P-AP: its three forces are c eta2 s_k (1 - s_k) s_l s_m (unit vector), every switch s = sigmoid(arg) < e^arg,
    // so once ANY of the three arguments is below A = log(2^-16 / (|c|max eta2)) all three forces round to zero in Q19.12.  Round 1 used
    // A = -30 (which is also where the reference's 1.0 nm cutoff sits: 140 (0.8 - 1.0) = -28); with the reference's gammas A is about
    // -16.6 and the first switch alone closes the list at r0 - A / eta2 = 0.92 nm.  Mirrored in the oracle's fp32 list scan.
    double pap_A = -30.0;
    {
        double cmax = 0;
        bool dense = !H.pap.empty();
        for (auto& h : H.pap) {
            const double m = table_max(h.fn, h.nres, h.gamma);
            if (m < 0) { dense = false; break; }
            cmax = std::max(cmax, std::fabs(h.k) * m * 2.0 * h.eta);
        }
        if (dense && cmax > 0 && !(getenv("AWSEM_PAP_SCREEN") && atoi(getenv("AWSEM_PAP_SCREEN")) == 30))
            pap_A = std::min(-8.0, std::max(-30.0, std::log(1.0 / (65536.0 * cmax))));
    }
    {   // pap
        std::vector<PapD<real>> D;
        for (auto& h : H.pap) {
            PapD<real> d;
            memset(&d, 0, sizeof d);
            d.group = h.group; d.nacc = (int)h.acc_res.size(); d.ndon = (int)h.don_res.size(); d.nres = h.nres;
            std::vector<Int4> acc(d.nacc), don(d.ndon);
            for (int i = 0; i < d.nacc; i++) acc[i] = Int4{h.acc_atoms[2 * i], h.acc_atoms[2 * i + 1], h.acc_res[i], 0};
            for (int i = 0; i < d.ndon; i++) don[i] = Int4{h.don_atoms[2 * i], h.don_atoms[2 * i + 1], h.don_res[i], 0};
            d.acc = upload(acc); d.don = upload(don);
            if (h.fn.kind) {
                d.fn = upload_fn(h.fn, h.nres, -h.k);
                fn_band(h.fn, h.nres, &d.band_lo, &d.band_hi);
            } else {
                d.gamma = upload_real(h.gamma, -h.k);
                d.nzmask = upload(nonzero_mask(h.gamma, h.nres));
                d.nz_lds = ((size_t)d.nacc + d.ndon + ((size_t)h.nres * h.nres + 31) / 32) * 4 <= (size_t)natoms_ * (sizeof(real) == 4 ? 16 : 24);
                zero_band(h.gamma, h.nres, &d.band_lo, &d.band_hi);
            }
            d.eta2 = (real)(2 * h.eta); d.r0 = (real)h.r0; d.d_i4 = (real)h.d_i4; d.cutoff2 = (real)(h.cutoff * h.cutoff);
            d.min_arg = sizeof(real) == 4 ? (real)pap_A : (real)-1e300;
            if (sizeof(real) == 4) {   // fp32: the first switch's argument is below pap_A beyond this distance
                const double rs = h.r0 - pap_A / (2.0 * h.eta);
                d.cutoff2 = (real)std::min(h.cutoff * h.cutoff, rs * rs);
            }
            D.push_back(d);
        }
        if (D.size() > AWSEM_MAX_PAP) throw std::string("too many pap terms: at most " + std::to_string(AWSEM_MAX_PAP) + " per system");
        S.n_pap = (int)D.size(); for (size_t t = 0; t < D.size(); t++) S.pap[t] = D[t];
    }
    {   // helical
        std::vector<HelicalD<real>> D;
        for (auto& h : H.helical) {
            std::vector<HelicalItem<real>> it(h.coef.size());
            const std::vector<int> perm = item_order((int)it.size(), 4, h.atoms.data());
            for (size_t b = 0; b < it.size(); b++) { int o = perm[b]; it[b] = HelicalItem<real>{h.atoms[4 * o], h.atoms[4 * o + 1], h.atoms[4 * o + 2], (real)h.coef[o]}; }
            HelicalD<real> d{h.group, (int)it.size(), upload(it), (real)h.r_ON, (real)h.r_OH, (real)(1.0 / (h.s_NO * h.s_NO)), (real)(1.0 / (h.s_HO * h.s_HO)),
                             nullptr, (real)(2 * h.eta), (real)h.z_m, (real)h.center};
            if (!h.coef_m.empty()) {
                std::vector<HelicalZItem<real>> zi(it.size());
                for (size_t b = 0; b < it.size(); b++) { int o = perm[b]; zi[b] = HelicalZItem<real>{h.atoms[4 * o + 3], (real)(h.coef_m[o] - h.coef[o])}; }
                d.zitems = upload(zi);
            }
            D.push_back(d);
        }
        if (D.size() > AWSEM_MAX_HELICAL) throw std::string("too many helical terms: at most " + std::to_string(AWSEM_MAX_HELICAL) + " per system");
        S.n_helical = (int)D.size(); for (size_t t = 0; t < D.size(); t++) S.helical[t] = D[t];
    }
    {   // fragment memory
        std::vector<FmD<real>> D;
        for (auto& h : H.fm) {
            const int nb = (int)(h.pairs.size() / 2);
            std::vector<int> perm;
            std::vector<FmBundle> bundles;
            std::vector<unsigned> partners;
            int nb_pad = 0;
            if (!cell_mode_ && natoms_ <= 65535 && !(getenv("AWSEM_FM_BUNDLES") && atoi(getenv("AWSEM_FM_BUNDLES")) == 0)) {
                // bundles: the bonds of an atom towards higher-numbered partners, cut into chunks of AWSEM_FM_BUNDLE; all first
                // chunks come before all second chunks, so that neighbouring lanes own neighbouring atoms
                std::vector<std::vector<std::pair<int, int>>> of(natoms_);   // owner -> (partner, bond)
                for (int b = 0; b < nb; b++) {
                    const int i = h.pairs[2 * b], j = h.pairs[2 * b + 1];
                    of[std::min(i, j)].push_back({std::max(i, j), b});
                }
                size_t maxdeg = 0;
                for (auto& v : of) { std::sort(v.begin(), v.end()); maxdeg = std::max(maxdeg, v.size()); }
                struct Chunk { int owner, lo, hi; };
                std::vector<Chunk> chunks;
                for (size_t c = 0; c * AWSEM_FM_BUNDLE < maxdeg; c++)
                    for (int a = 0; a < natoms_; a++)
                        if (of[a].size() > c * AWSEM_FM_BUNDLE) chunks.push_back(Chunk{a, (int)(c * AWSEM_FM_BUNDLE), (int)std::min(of[a].size(), (c + 1) * AWSEM_FM_BUNDLE)});
                nb_pad = ((int)chunks.size() + 63) & ~63;
                partners.assign((size_t)AWSEM_FM_BUNDLE * std::max(nb_pad, 1), 0u);
                for (size_t k = 0; k < chunks.size(); k++) {
                    const Chunk& c = chunks[k];
                    bundles.push_back(FmBundle{(unsigned)c.owner | ((unsigned)(c.hi - c.lo) << 16), (unsigned)perm.size()});
                    for (int u = 0; u < AWSEM_FM_BUNDLE; u++) partners[(size_t)u * nb_pad + k] = (unsigned)of[c.owner][c.lo + (c.lo + u < c.hi ? u : 0)].first;
                    for (int q = c.lo; q < c.hi; q++) perm.push_back(of[c.owner][q].second);
                }
            } else perm = item_order(nb, 2, h.pairs.data());
            std::vector<real2_t<real>> t2((size_t)nb * (h.ntab - 1));
            for (int b = 0; b < nb; b++) for (int i = 0; i + 1 < h.ntab; i++)
                t2[(size_t)b * (h.ntab - 1) + i] = real2_t<real>{(real)h.table[(size_t)perm[b] * h.ntab + i], (real)h.table[(size_t)perm[b] * h.ntab + i + 1]};
            const std::vector<int> pp = permuted(h.pairs, perm, 2);
            std::vector<unsigned> packed(nb, 0u);
            if (natoms_ <= 65535) for (int b = 0; b < nb; b++) packed[b] = (unsigned)pp[2 * b] | ((unsigned)pp[2 * b + 1] << 16);
            if ((size_t)nb * (h.ntab - 1) >= (1ull << 31)) throw std::string("fragment memory: table too large (nbonds x ntab must stay below 2^31)");
            // Cropped rows for the fp32 force walks of the bundled form (FmD::crop).  A bin is ACTIVE when k |slope|, computed in fp32 as the
            // kernel does, reaches 2^-15: below that every force component s * d * 4096 with |d / r| <= 1 rounds to exactly 0 in a Q19.12
            // accumulator (a factor 4 inside the rounding threshold 2^-13).  A bond keeps the nodes of its first..last active bin; all rows
            // have the length of the longest window, a row that would pass the table's end starts earlier.
            std::vector<float> crop;
            int crop_len = 0;
            if (sizeof(real) == 4 && !bundles.empty()) {
                const float kf = (float)h.k, inv_dr = (float)(1.0 / h.dr), thr = 1.0f / 32768.0f;
                const bool on = !(getenv("AWSEM_FM_CROP") && atoi(getenv("AWSEM_FM_CROP")) == 0);
                std::vector<int> first(nb, 0), last(nb, -1);
                crop_len = 2;
                for (int b = 0; b < nb && on; b++) {
                    const double* row = &h.table[(size_t)perm[b] * h.ntab];
                    int f = -1, l = -1;
                    for (int i = 0; i + 1 < h.ntab; i++) {
                        const float slope = ((float)row[i + 1] - (float)row[i]) * inv_dr;
                        if (std::fabs(kf * slope) >= thr) { if (f < 0) f = i; l = i; }
                    }
                    if (f >= 0) { first[b] = f; last[b] = l; crop_len = std::max(crop_len, l - f + 2); }
                }
                if (!on || crop_len > (h.ntab * 3) / 4) { crop_len = h.ntab; std::fill(first.begin(), first.end(), 0); }   // not worth cropping
                crop.resize((size_t)nb * crop_len);
                std::vector<unsigned> lo_of(nb);
                for (int b = 0; b < nb; b++) {
                    const int lo = std::min(first[b], h.ntab - crop_len);
                    lo_of[b] = (unsigned)lo;
                    for (int i = 0; i < crop_len; i++) crop[(size_t)b * crop_len + i] = (float)h.table[(size_t)perm[b] * h.ntab + lo + i];
                }
                // the window start rides in the high half of the bond's partner word (slots past a bundle's count repeat its first bond)
                for (size_t k = 0; k < bundles.size(); k++) {
                    const unsigned cnt = bundles[k].head >> 16, fb = bundles[k].first;
                    for (unsigned u = 0; u < AWSEM_FM_BUNDLE; u++) partners[(size_t)u * nb_pad + k] |= lo_of[fb + (u < cnt ? u : 0u)] << 16;
                }
            }
            if (getenv("AWSEM_INFO") && crop_len) fprintf(stderr, "[awsem] fragment memory: %d bonds, rows of %d nodes cropped to %d for the fp32 force walk (%.1f MB -> %.1f MB)\n", nb, h.ntab, crop_len, nb * (h.ntab - 1) * 8e-6, nb * crop_len * 4e-6);
            D.push_back(FmD<real>{h.group, nb, h.ntab, crop_len, upload(packed), upload(pp), upload(t2), (real)h.k, (real)h.dr, (real)(1.0 / h.dr),
                                  bundles.empty() ? nullptr : upload(bundles), partners.empty() ? nullptr : upload(partners), (int)bundles.size(), nb_pad,
                                  crop.empty() ? nullptr : upload(crop)});
        }
        if (D.size() > AWSEM_MAX_FM) throw std::string("too many fm terms: at most " + std::to_string(AWSEM_MAX_FM) + " per system");
        S.n_fm = (int)D.size(); for (size_t t = 0; t < D.size(); t++) S.fm[t] = D[t];
    }
    {   // membrane
        std::vector<MembD<real>> D;
        for (auto& h : H.memb) {
            std::vector<MembItem<real>> it(h.atoms.size());
            std::vector<int> perm(it.size());
            for (size_t b = 0; b < it.size(); b++) perm[b] = (int)b;
            if (cell_mode_) perm = item_order((int)it.size(), 1, h.atoms.data());
            for (size_t b = 0; b < it.size(); b++) it[b] = MembItem<real>{h.atoms[perm[b]], (real)h.scale[perm[b]]};
            D.push_back(MembD<real>{h.group, (int)it.size(), upload(it), (real)h.k, (real)(2 * h.k_m), (real)h.z_m, (real)h.center, h.kind, 0});
        }
        if (D.size() > AWSEM_MAX_MEMB) throw std::string("too many memb terms: at most " + std::to_string(AWSEM_MAX_MEMB) + " per system");
        S.n_memb = (int)D.size(); for (size_t t = 0; t < D.size(); t++) S.memb[t] = D[t];
    }
    {   // debye-hueckel
        std::vector<DhD<real>> D;
        for (auto& h : H.dh) {
            std::vector<DhItem<real>> it(h.qq.size());
            const std::vector<int> perm = spread_order((int)it.size(), 2, h.pairs.data(), natoms_);
            for (size_t b = 0; b < it.size(); b++) { int o = perm[b]; it[b] = DhItem<real>{h.pairs[2 * o], h.pairs[2 * o + 1], (real)h.qq[o], (real)0}; }
            D.push_back(DhD<real>{h.group, (int)it.size(), upload(it), (real)h.k, (real)(1.0 / h.lam)});
        }
        if (D.size() > AWSEM_MAX_DH) throw std::string("too many dh terms: at most " + std::to_string(AWSEM_MAX_DH) + " per system");
        S.n_dh = (int)D.size(); for (size_t t = 0; t < D.size(); t++) S.dh[t] = D[t];
    }
    {   // bond sums (pairs anywhere in the system, like Debye-Hueckel: never window-local on the cell-list path)
        std::vector<GaussD<real>> D;
        int ncv = 0;
        for (auto& h : H.gauss) {
            std::vector<GaussItem<real>> it(h.kg.size());
            const std::vector<int> perm = spread_order((int)it.size(), 2, h.pairs.data(), natoms_);
            for (size_t b = 0; b < it.size(); b++) {
                const int o = perm[b];
                it[b] = GaussItem<real>{h.pairs[2 * o], h.pairs[2 * o + 1], (real)(h.kind ? 0.0 : h.r0[o]),
                                        (real)(h.kind ? 0.0 : 0.5 / (h.sigma[o] * h.sigma[o])), (real)h.kg[o]};
            }
            D.push_back(GaussD<real>{h.group, (int)it.size(), upload(it), h.kind, h.cv, h.cv ? ncv++ : 0, 0, (real)h.cv_k, (real)h.cv_x0});
        }
        if (D.size() > AWSEM_MAX_GAUSS) throw std::string("too many bond-sum terms (Gaussian bonds, Q, Rg): at most " + std::to_string(AWSEM_MAX_GAUSS) + " per system");
        S.n_gauss = (int)D.size(); S.gauss = D.empty() ? nullptr : upload(D);
        hgauss_.assign(D.begin(), D.end());
        S.n_cv = ncv;
    }
    // pair lists: one slot per pair term, capacity = number of candidate pairs (cannot overflow)
    if (const char* env = getenv("AWSEM_SKIN")) {
        double v = atof(env);
        if (v >= 0.0 && v <= 1.0) { skin_ = v; skin_fixed_ = true; }
    }
    dstatus_ = dalloc<int>(1);
    HIP_TRY(hipMemset(dstatus_, 0, sizeof(int)));
    S.status = dstatus_;
    // phase cycles, then (start, end, hardware id) per replica (experiment builds), then the pacing records (two launch parities)
    S.prof = dalloc<long long>(pace_perm_offset(nrep_) + (size_t)nrep_);
    HIP_TRY(hipMemset(S.prof, 0, (pace_perm_offset(nrep_) + (size_t)nrep_) * sizeof(long long)));
    {
        std::vector<int> ident(nrep_);
        for (int r = 0; r < nrep_; r++) ident[r] = r;
        HIP_TRY(hipMemcpy(S.prof + pace_perm_offset(nrep_), ident.data(), (size_t)nrep_ * sizeof(int), hipMemcpyHostToDevice));
        if (const char* env = getenv("AWSEM_PLACE_EVERY")) place_every_ = atoi(env);
    }
    const size_t lds_max = lds_max_;
    lds_eval_ = lds_bytes<real>(lds_shape(S), false);
    lds_step_ = lds_bytes<real>(lds_shape(S), true);
    if (cell_mode_) {
        cell_setup(S, H);
    } else {
        if (natoms_ > 65535) throw std::string("the replica-resident path addresses atoms with 16 bits: at most 65535 atoms");
        long long off = 0;
        int id = 0;
        auto slot = [&](long long cap) {
            ListSlot ls{(int)off, (int)cap, id++};
            off += (cap + 63) & ~63LL;   // 256-byte aligned starts
            if (off > 0x7fffffffLL) throw std::string("pair lists of one replica exceed 2^31 entries");
            return ls;
        };
        // A list is rebuilt when a MASSIVE particle has moved further than skin/2.  A virtual site then has moved at most
        // sum|w| times that, so a list screened on atoms a and b is cut at cutoff + (amp(a) + amp(b)) / 2 * skin, amp = 1 for a
        // massive particle and sum|w| for a site (N of the AWSEM backbone: 1.373 -> beta and P-AP lists: 1.19 skin).
        std::vector<double> amp(natoms_, 1.0);
        for (size_t v = 0; v < H.vs_site.size(); v++) amp[H.vs_site[v]] = std::fabs(H.vs_w[3 * v]) + std::fabs(H.vs_w[3 * v + 1]) + std::fabs(H.vs_w[3 * v + 2]);
        auto amp_max = [&](const std::vector<int>& atoms, int stride) { double m = 0; for (size_t i = 0; i < atoms.size(); i += stride) if (atoms[i] >= 0) m = std::max(m, amp[atoms[i]]); return atoms.empty() ? 1.0 : m; };
        list_params_.clear();
        auto rc2 = [&](double cutoff, double margin) { list_params_.push_back({cutoff, margin}); double r = cutoff + margin * skin_; return (real)(r * r); };   // recorded in the order contact, excl, beta, pap: see apply_skin
        for (int t = 0; t < S.n_contact; t++) {
            ContactD<real>& d = S.contact[t];
            if (d.nsites > 1023) throw std::string("contact term: at most 1023 sites on the replica-resident path");
            d.ls = slot((long long)d.nsites * (d.nsites - 1) / 2);
            d.list_rc2 = rc2((H.contact[t].flags & AWSEM_CONTACT_NOCUTOFF) ? 1e15 : H.contact[t].p[AWSEM_CP_CUTOFF], amp_max(H.contact[t].atom, 1));
        }
        for (int t = 0; t < S.n_excl; t++) {
            ExclD<real>& d = S.excl[t];
            long long cap = 0;
            for (int c = 0; c < d.nclass; c++) { long long n = d.class_start[c + 1] - d.class_start[c]; cap += n * (n - 1) / 2; }
            d.ls = slot(cap);
            d.list_rc2 = rc2(H.excl[t].r0, amp_max(H.excl[t].atom, 1));
        }
        // hydrogen-bond terms: HbEntry records (atoms in 15 bits).  Terms of one force group with the same constants share a list and a walk.
        if ((S.n_beta || S.n_pap) && natoms_ > 32767) throw std::string("hydrogen-bond terms on the replica-resident path address atoms with 15 bits: at most 32767 atoms");
        const long long ewords = (long long)(sizeof(HbEntry<real>) / sizeof(unsigned));
        for (int t = 0; t < S.n_beta; t++) {
            BetaD<real>& d = S.beta[t];
            if (d.nacc > 65535 || d.ndon > 65535) throw std::string("hydrogen-bond term: at most 65535 acceptors / donors");
            d.list_rc2 = rc2(std::sqrt((double)d.screen2), 0.5 * (amp_max(H.beta[t].acc_atoms, 3) + amp_max(H.beta[t].don_atoms, 3)));
        }
        for (int t = 0; t < S.n_pap; t++) {
            PapD<real>& d = S.pap[t];
            if (d.nacc > 65535 || d.ndon > 65535) throw std::string("hydrogen-bond term: at most 65535 acceptors / donors");
            d.list_rc2 = rc2(std::sqrt((double)d.cutoff2), 0.5 * (amp_max(H.pap[t].acc_atoms, 2) + amp_max(H.pap[t].don_atoms, 2)));
        }
        const bool share = !(getenv("AWSEM_HB_SHARE") && atoi(getenv("AWSEM_HB_SHARE")) == 0);   // 0: one list per term (A/B runs)
        auto share_lists = [&](auto* terms, int n, auto same) {
            for (int t = 0; t < n; t++) {
                int first = t;
                for (int u = 0; u < t && share; u++) if (terms[u].list_owner && same(terms[u], terms[t])) { first = u; break; }
                if (first != t) continue;
                long long cap = 0;
                for (int u = t; u < n; u++) if (u == t || (share && same(terms[t], terms[u]))) cap += (long long)terms[u].nacc * terms[u].ndon;
                if (cap > 0x7fffffffLL / ewords) throw std::string("hydrogen-bond pair lists of one replica exceed 2^31 words");
                ListSlot ls = slot(cap * ewords);
                ls.list_cap = (int)cap;   // in records
                for (int u = t; u < n; u++) if (u == t || (share && same(terms[t], terms[u]))) { terms[u].ls = ls; terms[u].list_owner = u == t; }
            }
        };
        share_lists(S.beta, S.n_beta, [](const BetaD<real>& a, const BetaD<real>& b) {
            return a.group == b.group && a.r_ON == b.r_ON && a.r_OH == b.r_OH && a.inv_sNO2 == b.inv_sNO2 && a.inv_sHO2 == b.inv_sHO2 && a.cutoff2 == b.cutoff2 &&
                   a.screen2 == b.screen2 && a.max_expo == b.max_expo; });
        share_lists(S.pap, S.n_pap, [](const PapD<real>& a, const PapD<real>& b) {
            return a.group == b.group && a.eta2 == b.eta2 && a.r0 == b.r0 && a.d_i4 == b.d_i4 && a.cutoff2 == b.cutoff2 && a.min_arg == b.min_arg; });
        {   // may the step kernel place the virtual sites inside the first force phase?  Only if that phase reads massive particles only
            bool ok = true;
            auto massive = [&](const std::vector<int>& atoms) { for (int a : atoms) if (a >= 0 && !(H.mass[a] > 0)) ok = false; };
            for (auto& h : H.contact) massive(h.atom);
            for (auto& h : H.fm) massive(h.pairs);
            for (auto& h : H.gauss) if (h.cv) massive(h.pairs);
            S.fuse_vs = ok && !(getenv("AWSEM_FUSE_VS") && atoi(getenv("AWSEM_FUSE_VS")) == 0) ? 1 : 0;
        }
        S.n_lists = id;
        S.list_stride = off ? off : 64;
        const double bytes = (double)S.list_stride * 4.0 * nrep_;
        if (2 * bytes > 128e9) {   // two buffers: the dynamics' lists and the queries'
            char msg[256];
            snprintf(msg, sizeof msg, "pair lists need %.1f GB for %d replicas of this system; use fewer replicas per handle", bytes / 1e9, nrep_);
            throw std::string(msg);
        }
        S.lists = dalloc<unsigned>((size_t)S.list_stride * nrep_);
        lists_eval_ = dalloc<unsigned>((size_t)S.list_stride * nrep_);
        S.list_ref = dalloc<real>((size_t)nrep_ * 3 * std::max(1, S.nmassive));
        S.list_counts = dalloc<unsigned>((size_t)nrep_ * AWSEM_MAX_LISTS);
        list_reuse_ = !(getenv("AWSEM_LIST_REUSE") && atoi(getenv("AWSEM_LIST_REUSE")) == 0);
        query_reuse_ = !(getenv("AWSEM_QUERY_REUSE") && atoi(getenv("AWSEM_QUERY_REUSE")) == 0);
        if (!h_lmean_) { HIP_TRY(hipHostMalloc((void**)&h_lmean_, 2 * sizeof(double), hipHostMallocDefault)); h_lmean_[0] = h_lmean_[1] = 0; }
        skin_sites_ = S.max_contact_sites > 0 ? (double)S.max_contact_sites : (double)natoms_ / 6.0;
        S.skin_half2 = (real)(0.25 * skin_ * skin_ * 0.999);
        dcounts_ = dalloc<unsigned>((size_t)nrep_ * AWSEM_MAX_LISTS);
    }
    dsys_ = S;

    // replica state and outputs
    size_t nstate = (size_t)nrep_ * 3 * natoms_;
    pos_ = dalloc<real>(nstate); vel_raw_ = dalloc<real>(nstate);
    HIP_TRY(hipMemset(pos_, 0, nstate * sizeof(real)));
    HIP_TRY(hipMemset(vel_raw_, 0, nstate * sizeof(real)));
    dT_ = dalloc<double>(nrep_); dtmp_ = dalloc<double>(nrep_); dE_ = dalloc<double>((size_t)nrep_ * AWSEM_NGROUPS); dF_ = dalloc<double>(nstate);

    if (cell_mode_) {   // the cell-list kernels use no dynamic LDS
        cell_.f = dalloc<u64>(2 * nstate);   // two parities: a step sums into one while the integrator clears the other for the next (cell_integrate_kernel)
        HIP_TRY(hipMemset(cell_.f, 0, 2 * nstate * sizeof(u64)));
        cell_.ref = dalloc<real>(nstate);
        return;
    }
    // launch geometry: one workgroup per replica, LDS sized by the system
    step_ok_ = lds_step_ <= lds_max;   // energies/forces need less LDS than dynamics (no velocities, no list reference positions)
    HIP_TRY(hipFuncSetAttribute((const void*)awsem_eval_kernel<true, real, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eval_));
    HIP_TRY(hipFuncSetAttribute((const void*)awsem_eval_kernel<false, real, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eval_));
    HIP_TRY(hipFuncSetAttribute((const void*)awsem_eval_kernel<true, real, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eval_));
    HIP_TRY(hipFuncSetAttribute((const void*)awsem_eval_kernel<false, real, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eval_));
    HIP_TRY(hipFuncSetAttribute((const void*)awsem_eval_kernel<true, real, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eval_));
    HIP_TRY(hipFuncSetAttribute((const void*)awsem_eval_kernel<false, real, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eval_));
    HIP_TRY(hipFuncSetAttribute((const void*)awsem_eval_kernel<true, real, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eval_));
    HIP_TRY(hipFuncSetAttribute((const void*)awsem_eval_kernel<false, real, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eval_));
    if (step_ok_) HIP_TRY(hipFuncSetAttribute((const void*)awsem_step_kernel<real, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step_));
    if (step_ok_) HIP_TRY(hipFuncSetAttribute((const void*)awsem_step_kernel<real, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step_));
    if (step_ok_) HIP_TRY(hipFuncSetAttribute((const void*)awsem_step_kernel<real, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step_));
    if (step_ok_) HIP_TRY(hipFuncSetAttribute((const void*)awsem_step_kernel<real, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step_));
    if (step_ok_) HIP_TRY(hipFuncSetAttribute((const void*)awsem_minimize_kernel<real>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step_));
    HIP_TRY(hipFuncSetAttribute((const void*)awsem_list_kernel<real, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eval_));
    HIP_TRY(hipFuncSetAttribute((const void*)awsem_list_kernel<real, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eval_));
    // Workgroup size: the waves a CU can hold at this kernel's register budget (16) divided among the replicas whose LDS
    // images fit one CU -- four small replicas share a CU with 4 waves each (more waves per replica were measured slower
    // there: 1r69, 320 / 384 threads), a replica that has the CU to itself (2xov: 104 KB) takes all 16 (measured +45 %
    // over 512 threads).
    {
        size_t per_cu = std::max<size_t>(1, lds_max / std::max<size_t>(1, step_ok_ ? lds_step_ : lds_eval_));
        per_cu = std::min(per_cu, std::max<size_t>(1, ((size_t)nrep_ + n_cu_ - 1) / n_cu_));   // fewer replicas than slots: a CU holds fewer (1r69, <= 256 replicas: 14.3 -> 12.1 us per step)
        block_ = per_cu >= 4 ? 256 : per_cu >= 2 && per_cu * 512 <= 1024 ? 512 : per_cu == 1 ? 1024 : 256;   // powers of two only: 320 threads (three replicas per CU) were measured 70 % slower than 256
        // The priority bias is learnt from the end times of the workgroups that shared a CU from the first step to the last: that is a
        // grid of one wave.  With more replicas than slots workgroups start whenever a slot frees up and their end times say nothing
        // about each other (measured: 2048 / 4096 replicas of 1r69 lose 4-6 % with the feedback on, 0.8 % with the gain at zero): the
        // plain rotation takes over.
        if ((dsys_.prio & 0xff) == 3 && (size_t)nrep_ > (size_t)n_cu_ * std::min<size_t>(per_cu, 4)) { dsys_.prio = 1; place_every_ = 0; }
    }
    if (sizeof(real) == 8) block_ = std::min(block_, 512);   // the fp64 kernels are compiled for at most 512 threads (AWSEM_BOUNDS)
    if (getenv("AWSEM_INFO"))
        fprintf(stderr, "[awsem] %d atoms x %d replicas: LDS %zu B per replica stepping (%zu B evaluating) of %zu B per CU -> %zu per CU, %d threads per replica\n",
                natoms_, nrep_, lds_step_, lds_eval_, lds_max, lds_max / std::max<size_t>(1, lds_step_), block_);
    if (const char* env = getenv("AWSEM_BLOCK")) {
        int b = atoi(env);
        if (b >= 64 && b <= (sizeof(real) == 8 ? 512 : 1024) && (b & (b - 1)) == 0) block_ = b;   // a power of two: the kernels deal work to waves with masks
    }
    // One distance scan for the hydrogen-bond terms that share a list (build_hbond_group): the members must screen the same atoms
    // against the same radius, and a wave's share of the donors must fit a 64-bit mask (ndon <= 64 waves = block size).
    if (!(getenv("AWSEM_HB_SCAN") && atoi(getenv("AWSEM_HB_SCAN")) == 0)) {
        auto group_scan = [&](auto* terms, int n, const auto& hterms, int stride) {
            for (int t = 0; t < n; t++) {
                if (!terms[t].list_owner) continue;
                std::vector<int> members;
                for (int u = t; u < n; u++) if (terms[u].ls.list_id == terms[t].ls.list_id) members.push_back(u);
                bool ok = members.size() >= 2 && terms[t].ndon <= block_ && terms[t].nacc > 0;
                for (int u : members) {
                    if (!ok) break;
                    ok = terms[u].nacc == terms[t].nacc && terms[u].ndon == terms[t].ndon && terms[u].nres == terms[t].nres && terms[u].list_rc2 == terms[t].list_rc2 &&
                         hterms[u].acc_res == hterms[t].acc_res && hterms[u].don_res == hterms[t].don_res;
                    for (int i = 0; ok && i < terms[t].nacc; i++) ok = hterms[u].acc_atoms[stride * i] == hterms[t].acc_atoms[stride * i];
                    for (int i = 0; ok && i < terms[t].ndon; i++) ok = hterms[u].don_atoms[stride * i] == hterms[t].don_atoms[stride * i];
                }
                if (!ok) continue;
                const size_t words = ((size_t)terms[t].nres * terms[t].nres + 31) / 32;
                const bool fits = ((size_t)terms[t].nacc + terms[t].ndon + members.size() * words) * 4 <= (size_t)natoms_ * (sizeof(real) == 4 ? 16 : 24);
                for (int u : members) { terms[u].scan_mode = 1; if (!fits) terms[u].nz_lds = 0; }
            }
        };
        group_scan(dsys_.beta, dsys_.n_beta, H.beta, 3);
        group_scan(dsys_.pap, dsys_.n_pap, H.pap, 2);
        if (dsys_.n_beta) dsys_.beta_dev = upload(std::vector<BetaD<real>>(dsys_.beta, dsys_.beta + dsys_.n_beta));
        if (dsys_.n_pap) dsys_.pap_dev = upload(std::vector<PapD<real>>(dsys_.pap, dsys_.pap + dsys_.n_pap));
    }
    {   // the lean instantiation of the step kernel (evaluate_terms<LEAN>): none of the rarer term shapes present
        bool lean = dsys_.n_gauss == 0 && dsys_.n_cv == 0 && dsys_.n_helical <= 1 && dsys_.n_memb <= 1 && dsys_.n_contact <= 1 && (dsys_.n_rama == 0 || dsys_.ramam.n > 0);
        for (int t = 0; t < dsys_.n_fm; t++) lean = lean && dsys_.fm[t].nbundles > 0;
        // flavours: 1 plain contact_term; 2 the same with one helical term; 3 the z-dependent contact term with one membrane term
        const bool zdep = dsys_.n_contact == 1 && dsys_.contact[0].flags == 1;
        const int flavour = dsys_.n_memb || zdep ? 3 : dsys_.n_helical ? 2 : 1;
        if (flavour == 3) lean = lean && zdep && dsys_.n_memb == 1 && dsys_.n_helical == 0;
        else for (int t = 0; t < dsys_.n_contact; t++) lean = lean && dsys_.contact[t].flags == 0;
        // every term scans for itself, dense tables, and a wave's share of the donors fits the 64-bit survivor mask of build_hbond<LEAN>
        // (the evaluation kernel runs at the same block size or a larger one)
        for (int t = 0; t < dsys_.n_beta; t++) lean = lean && dsys_.beta[t].scan_mode == 0 && !dsys_.beta[t].fn && dsys_.beta[t].ndon <= block_;
        for (int t = 0; t < dsys_.n_pap; t++) lean = lean && dsys_.pap[t].scan_mode == 0 && !dsys_.pap[t].fn && dsys_.pap[t].ndon <= block_;
        lean = lean && dsys_.fuse_vs == 1 && dsys_.nmassive <= 2 * block_ && (dsys_.prio & 0xff) != 2 && dsys_.remove_cm != 0;
        // one term of a kind (the harmonic terms merged into one list), the beta / P-AP terms behind one shared list each, owned by the first
        lean = lean && dsys_.n_harm <= 1 && dsys_.n_chi <= 1 && dsys_.n_fm <= 1 && dsys_.n_dh <= 1 && dsys_.n_excl <= 1 && dsys_.n_beta <= 3 && dsys_.n_pap <= 2;
        // the lean list builders stage a position + word record per site in the accumulators' memory (SiteRec: 16 B, 32 B in fp64)
        const size_t rec = sizeof(real) == 4 ? 16 : 32, room = (size_t)natoms_ * (sizeof(real) == 4 ? 16 : 24);
        for (int t = 0; t < dsys_.n_excl; t++) lean = lean && (size_t)dsys_.excl[t].nsites * rec <= room;
        for (int t = 1; t < dsys_.n_beta; t++) lean = lean && !dsys_.beta[t].list_owner;
        for (int t = 1; t < dsys_.n_pap; t++) lean = lean && !dsys_.pap[t].list_owner;
        step_lean_ = lean && !(getenv("AWSEM_STEP_LEAN") && atoi(getenv("AWSEM_STEP_LEAN")) == 0) ? flavour : 0;
        if (getenv("AWSEM_INFO")) {
            fprintf(stderr, "[awsem] step kernel: %s instantiation", step_lean_ == 3 ? "lean + membrane" : step_lean_ == 2 ? "lean + helical" : step_lean_ ? "lean" : "full");
            if (!step_lean_) {
                fprintf(stderr, " (gauss %d cv %d helical %d memb %d contact %d rama %d/%d fuse_vs %d massive %d block %d prio %d; contact flags", dsys_.n_gauss, dsys_.n_cv, dsys_.n_helical,
                        dsys_.n_memb, dsys_.n_contact, dsys_.n_rama, dsys_.ramam.n, dsys_.fuse_vs, dsys_.nmassive, block_, dsys_.prio & 0xff);
                for (int t = 0; t < dsys_.n_contact; t++) fprintf(stderr, " %d", dsys_.contact[t].flags);
                fprintf(stderr, "; beta scan/fn");
                for (int t = 0; t < dsys_.n_beta; t++) fprintf(stderr, " %d/%d", dsys_.beta[t].scan_mode, dsys_.beta[t].fn != nullptr);
                fprintf(stderr, "; pap scan/fn");
                for (int t = 0; t < dsys_.n_pap; t++) fprintf(stderr, " %d/%d", dsys_.pap[t].scan_mode, dsys_.pap[t].fn != nullptr);
                fprintf(stderr, "; fm bundles");
                for (int t = 0; t < dsys_.n_fm; t++) fprintf(stderr, " %d", dsys_.fm[t].nbundles);
                fprintf(stderr, ")");
            }
            fprintf(stderr, "\n");
        }
    }
}

// Pair set of one pair term: rebuild that term's list with cutoff + skin and decode the entries.
template <class real> int Engine<real>::neighbors(int rep, int kind, int index, double skin, int32_t* pairs, int capacity) {
    HIP_TRY(hipSetDevice(device_));
    if (cell_mode_) return cell_neighbors(rep, kind, index, skin, pairs, capacity);
    DevSys<real> S = eval_sys();
    ListSlot ls;
    int group;
    auto rc2 = [&](double cutoff) { double r = cutoff + skin; return (real)(r * r); };
    const auto& Hs = *host_;
    switch (kind) {
    case AWSEM_LIST_EXCL:
        if (index < 0 || index >= S.n_excl) throw std::string("neighbors: no such excl term");
        S.excl[index].list_rc2 = rc2(Hs.excl[index].r0); ls = S.excl[index].ls; group = S.excl[index].group; break;
    case AWSEM_LIST_CONTACT:
        if (index < 0 || index >= S.n_contact) throw std::string("neighbors: no such contact term");
        S.contact[index].list_rc2 = rc2((Hs.contact[index].flags & AWSEM_CONTACT_NOCUTOFF) ? 1e15 : Hs.contact[index].p[AWSEM_CP_CUTOFF]);
        ls = S.contact[index].ls; group = S.contact[index].group; break;
    case AWSEM_LIST_BETA:
        if (index < 0 || index >= S.n_beta) throw std::string("neighbors: no such beta term");
        for (int t = 0; t < S.n_beta; t++) S.beta[t].scan_mode = 0;   // this term alone, by its own scan
        S.beta[index].list_rc2 = rc2(std::sqrt((double)S.beta[index].screen2)); ls = S.beta[index].ls; group = S.beta[index].group;
        for (int t = 0; t < S.n_beta; t++) if (t != index) S.beta[t].nacc = S.beta[t].ndon = 0;   // the terms that share the list stay out of it
        break;
    case AWSEM_LIST_PAP:
        if (index < 0 || index >= S.n_pap) throw std::string("neighbors: no such pap term");
        for (int t = 0; t < S.n_pap; t++) S.pap[t].scan_mode = 0;
        S.pap[index].list_rc2 = rc2(std::sqrt((double)S.pap[index].cutoff2)); ls = S.pap[index].ls; group = S.pap[index].group;
        for (int t = 0; t < S.n_pap; t++) if (t != index) S.pap[t].nacc = S.pap[t].ndon = 0;
        break;
    default: throw std::string("neighbors: kind must be AWSEM_LIST_EXCL, _CONTACT, _BETA or _PAP");
    }
    if (step_lean_) hipLaunchKernelGGL((awsem_list_kernel<real, true>), dim3(nrep_), dim3(block_), lds_eval_, stream_, S, pos_, 1u << group, dcounts_);
    else hipLaunchKernelGGL((awsem_list_kernel<real, false>), dim3(nrep_), dim3(block_), lds_eval_, stream_, S, pos_, 1u << group, dcounts_);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream_));
    check_status();
    unsigned n = 0;
    HIP_TRY(hipMemcpy(&n, dcounts_ + (size_t)rep * AWSEM_MAX_LISTS + ls.list_id, sizeof n, hipMemcpyDeviceToHost));
    std::vector<std::pair<int, int>> out(n);
    std::vector<unsigned> e(kind == AWSEM_LIST_BETA || kind == AWSEM_LIST_PAP ? 0 : n);
    if (!e.empty()) HIP_TRY(hipMemcpy(e.data(), S.lists + (size_t)rep * S.list_stride + ls.list_off, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToHost));
    if (e.empty() && n) {   // HbEntry records: the pair is named by its first acceptor and donor atoms
        std::vector<HbEntry<real>> he(n);
        HIP_TRY(hipMemcpy(he.data(), S.lists + (size_t)rep * S.list_stride + ls.list_off, (size_t)n * sizeof(HbEntry<real>), hipMemcpyDeviceToHost));
        const int nacc = kind == AWSEM_LIST_BETA ? dsys_.beta[index].nacc : dsys_.pap[index].nacc, ndon = kind == AWSEM_LIST_BETA ? dsys_.beta[index].ndon : dsys_.pap[index].ndon;
        std::vector<Int4> acc(nacc), don(ndon);
        HIP_TRY(hipMemcpy(acc.data(), kind == AWSEM_LIST_BETA ? dsys_.beta[index].acc : dsys_.pap[index].acc, sizeof(Int4) * nacc, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(don.data(), kind == AWSEM_LIST_BETA ? dsys_.beta[index].don : dsys_.pap[index].don, sizeof(Int4) * ndon, hipMemcpyDeviceToHost));
        std::map<int, int> ia, id;
        for (int i = 0; i < nacc; i++) ia[acc[i].x] = i;
        for (int i = 0; i < ndon; i++) id[don[i].x] = i;
        for (unsigned k = 0; k < n; k++) out[k] = {ia.at((int)(he[k].w0 & 0x7fffu)), id.at((int)((he[k].w0 >> 16) & 0x7fffu))};
    }
    for (unsigned k = 0; k < n && !e.empty(); k++) {
        if (kind == AWSEM_LIST_CONTACT) out[k] = {(int)(e[k] & 0x3ff), (int)((e[k] >> 10) & 0x3ff)};
        else out[k] = {(int)(e[k] >> 16), (int)(e[k] & 0xffff)};
        if (kind == AWSEM_LIST_EXCL && !to_orig_.empty()) out[k] = {to_orig_[out[k].first], to_orig_[out[k].second]};
        if (kind == AWSEM_LIST_EXCL && out[k].first > out[k].second) std::swap(out[k].first, out[k].second);
    }
    std::sort(out.begin(), out.end());
    for (unsigned k = 0; k < n && (int)k < capacity; k++) { pairs[2 * k] = out[k].first; pairs[2 * k + 1] = out[k].second; }
    return (int)n;
}

// ------------------------------------------------------------------------------------------------ cell-list path
template <class real> void Engine<real>::cell_setup(DevSys<real>& S, const HostSystem& H) {
    if ((S.n_beta || S.n_pap) && !S.n_excl)
        throw std::string("cell-list path: the hydrogen-bond terms need the excluded-volume term in the system (its CA and O atoms drive the list rebuilds)");
    // rows of the hydrogen-bond terms: screening distance + 1.2 skin (the donor's N is a virtual site, see awsem_cell.h)
    for (int t = 0; t < S.n_beta; t++) { const double r = std::sqrt((double)S.beta[t].screen2) + 1.2 * skin_; cell_hb_rc2_beta_[t] = (real)(r * r); }
    for (int t = 0; t < S.n_pap; t++) { const double r = std::sqrt((double)S.pap[t].cutoff2) + 1.2 * skin_; cell_hb_rc2_pap_[t] = (real)(r * r); }
    dsys_beta_n_ = S.n_beta; dsys_pap_n_ = S.n_pap;
    cell_hb_ = HbTerms<real>{};
    cell_hb_.n_beta = S.n_beta; cell_hb_.n_pap = S.n_pap;
    if (S.n_beta) cell_hb_.beta = upload(std::vector<BetaD<real>>(S.beta, S.beta + S.n_beta));
    if (S.n_pap) cell_hb_.pap = upload(std::vector<PapD<real>>(S.pap, S.pap + S.n_pap));
    if (S.n_beta + S.n_pap) cell_hb_.pos4 = dalloc<Rec<real>>((size_t)nrep_ * natoms_);
    cell_hb_maxacc_ = 0;
    for (int t = 0; t < S.n_beta; t++) cell_hb_maxacc_ = std::max(cell_hb_maxacc_, S.beta[t].nacc);
    for (int t = 0; t < S.n_pap; t++) cell_hb_maxacc_ = std::max(cell_hb_maxacc_, S.pap[t].nacc);
    for (int t = 0; t < S.n_beta; t++) { cell_nacc_beta_[t] = S.beta[t].nacc; cell_ndon_beta_[t] = S.beta[t].ndon; }
    for (int t = 0; t < S.n_pap; t++) { cell_nacc_pap_[t] = S.pap[t].nacc; cell_ndon_pap_[t] = S.pap[t].ndon; }
    if (S.n_contact > 1 || S.n_excl > 1) throw std::string("cell-list path: at most one contact and one excluded-volume term");
    if (S.n_contact && S.contact[0].nsites >= (1 << 24)) throw std::string("cell-list path: at most 2^24 contact sites");
    if (const char* env = getenv("AWSEM_CELL_NEIGHBORS")) { int v = atoi(env); if (v >= 16 && v <= 4096) cell_per_site_ = v; }
    CellSys<real>& C = cell_;
    memset(&C, 0, sizeof C);
    C.natoms = natoms_; C.nrep = nrep_; C.nsites = S.n_contact ? S.contact[0].nsites : 0;
    cell_rc_contact_ = S.n_contact ? H.contact[0].p[AWSEM_CP_CUTOFF] : 0.0;
    // NoCutoff (standalone burial_term): a cell list needs a radius.  The density switch theta_I(r) carries the factor
    // sig(2 eta (rmax - r)) <= e^-60 = 8.8e-27 beyond rmax + 30 / eta (1.25 nm with the reference's constants): below the
    // resolution of the Q31.32 density sums (2.3e-10) and of fp64 next to any density that matters, so rows cut there give
    // the NoCutoff result bit for bit in the accumulators.
    if (S.n_contact && (H.contact[0].flags & AWSEM_CONTACT_NOCUTOFF))
        cell_rc_contact_ = H.contact[0].p[AWSEM_CP_RMAX] + 30.0 / H.contact[0].p[AWSEM_CP_ETA];
    cell_rc_excl_ = S.n_excl ? H.excl[0].r0 : 0.0;
    C.skin_half2 = (real)(0.25 * skin_ * skin_ * 0.999);
    C.status = dstatus_;
    const size_t R = nrep_;
    // bin 0: the contact sites; bin 1: the excluded-volume atoms (class-sorted table of the term)
    CellBin<real>& B0 = C.bin[0];
    CellBin<real>& B1 = C.bin[1];
    B0.n = C.nsites; B0.atom = S.n_contact ? S.contact[0].atom : nullptr;
    B1.n = S.n_excl ? (int)excl_sorted_atom_.size() : 0; B1.atom = S.n_excl ? S.excl[0].atom : nullptr;
    if (S.n_excl) { C.x_partners = S.excl[0].partners; C.x_class = upload(excl_sorted_cls_); }
    for (int i = 0; i < B0.n; i++) if (H.mass[H.contact[0].atom[i]] <= 0) throw std::string("cell-list path: contact sites must be massive particles");
    for (int a : excl_sorted_atom_) if (S.n_excl && H.mass[a] <= 0) throw std::string("cell-list path: excluded-volume sites must be massive particles");
    for (int b = 0; b < 2; b++) {
        CellBin<real>& B = C.bin[b];
        const double rc = (b == 0 ? cell_rc_contact_ : cell_rc_excl_) + skin_;
        B.edge_min = (real)rc; B.rc2 = (real)(rc * rc);
        const size_t n = std::max(1, B.n);
        B.grid = dalloc<real>(R * 4); B.dims = dalloc<int>(R * 4);
        B.cell_count = dalloc<int>(R * AWSEM_CELL_MAXCELLS); B.cell_start = dalloc<int>(R * (AWSEM_CELL_MAXCELLS + 1));
        B.tile_sum = dalloc<int>(R * (AWSEM_CELL_MAXCELLS / 1024));
        B.cell_of = dalloc<int>(R * n); B.sorted = dalloc<int>(R * n);
        B.rec = dalloc<real>(R * n * 4);
        B.row_count = dalloc<int>(R * n);
        HIP_TRY(hipMemset(B.row_count, 0, R * n * sizeof(int)));
        HIP_TRY(hipMemset(B.dims, 0, R * 4 * sizeof(int)));
    }
    {
        const long long n = std::max<long long>(std::max(S.nvs, S.nfixed), (long long)B0.n + B1.n);
        C.nblk_pre = (int)std::max(1LL, (n + 255) / 256);
        C.bbox = dalloc<int>(R * C.nblk_pre * 8);
    }
    C.flag = dalloc<unsigned>(R * 2);
    HIP_TRY(hipMemset(C.flag, 0, R * 2 * sizeof(unsigned)));
    C.nblk_int = (int)cell_grid(S.nmassive).x;
    {   // round 3: what the fused step needs (awsem_cell.h: cell_integrate_kernel with next_cur, cell_window_kernel<VS>, cell_rebuild_kernel)
        C.nblk_box = std::max(C.nblk_pre, C.nblk_int);
        std::vector<int> box((size_t)R * C.nblk_box * 8, 0);
        for (size_t w = 0; w < (size_t)R * C.nblk_box; w++) for (int k = 0; k < 3; k++) { box[8 * w + k] = 0x7fffffff; box[8 * w + 4 + k] = (int)0x80000000; }
        dfree(C.bbox);
        C.bbox = const_cast<int*>(upload(box));   // boxes nobody writes stay neutral
        std::vector<int> m0(natoms_, -1), m1(natoms_, -1), vso(natoms_, -1);
        for (int i = 0; i < B0.n; i++) m0[H.contact[0].atom[i]] = i;
        for (int i = 0; i < B1.n; i++) m1[excl_sorted_atom_[i]] = i;
        for (size_t t = 0; t < vs_dev_site_.size(); t++) vso[vs_dev_site_[t]] = (int)t;   // index into DevSys::vs (its own order)
        C.vs_of = upload(vso);
        std::vector<Int4> pre;
        for (int i = 0; i < natoms_; i++) if (H.mass[i] > 0)   // DevSys::massive's order
            pre.push_back(Int4{m0[i], m1[i], m0[i] >= 0 ? (H.contact[0].resid[m0[i]] | (H.contact[0].chain[m0[i]] << 16) | (H.contact[0].restype[m0[i]] << 27)) : 0, 0});
        C.pre_items = upload(pre);
        C.bar = dalloc<unsigned>(R * 2 * 4);
        HIP_TRY(hipMemset(C.bar, 0, R * 2 * 4 * sizeof(unsigned)));
        for (int b = 0; b < 2; b++) {   // the rebuild kernel's invariant: counts are zero between rebuilds
            HIP_TRY(hipMemset(C.bin[b].cell_count, 0, R * AWSEM_CELL_MAXCELLS * sizeof(int)));
            HIP_TRY(hipMemset(C.bin[b].tile_sum, 0, R * (AWSEM_CELL_MAXCELLS / 1024) * sizeof(int)));
        }
        cell_fused_ = !(getenv("AWSEM_CELL_FUSED") && atoi(getenv("AWSEM_CELL_FUSED")) == 0);
    }
    C.cm = dalloc<u64>(2 * R * C.nblk_int * 3);
    HIP_TRY(hipMemset(C.cm, 0, 2 * R * C.nblk_int * 3 * sizeof(u64)));
    C.eg = dalloc<u64>(R * AWSEM_NGROUPS);
    const size_t ns = std::max(1, C.nsites);
    C.rho = dalloc<u64>(R * ns); C.drho = dalloc<u64>(R * ns);
    C.rhov = dalloc<real>(R * ns); C.alpha = dalloc<real>(R * ns); C.dalpha = dalloc<real>(R * ns);
    cell_alloc_lists();
    // bonded terms: the items local to a window go through cell_window_kernel (LDS accumulation); the others (and
    // Debye-Hueckel, whose pairs span the whole system) one job per 256 items with global atomics
    std::vector<CellJob> jobs;
    int slot = 0;
    auto add = [&](int kind, int term, int n, bool local_kind) {
        const int first = local_kind ? cell_nlocal_[slot++] : 0;
        for (int f = first; f < n; f += 256) jobs.push_back(CellJob{kind, term, f});
    };
    for (int t = 0; t < S.n_harm; t++) add(CJ_HARM, t, S.harm[t].n, true);
    for (int t = 0; t < S.n_chi; t++) add(CJ_CHI, t, S.chi[t].n, true);
    for (int t = 0; t < S.n_rama; t++) add(CJ_RAMA, t, S.rama[t].n, true);
    for (int t = 0; t < S.n_helical; t++) add(CJ_HELICAL, t, S.helical[t].n, true);
    for (int t = 0; t < S.n_fm; t++) add(CJ_FM, t, S.fm[t].n, true);
    for (int t = 0; t < S.n_memb; t++) add(CJ_MEMB, t, S.memb[t].n, true);
    for (int t = 0; t < S.n_dh; t++) add(CJ_DH, t, S.dh[t].n, false);
    std::vector<CellJob> cv_sum_jobs, cv_apply_jobs;   // collective-variable terms: the sums in one launch, energy + forces in the next
    for (int t = 0; t < S.n_gauss; t++) {
        if (!hgauss_[t].cv) { add(CJ_GAUSS, t, hgauss_[t].n, false); continue; }
        for (int f = 0; f < std::max(1, hgauss_[t].n); f += 256) { cv_sum_jobs.push_back(CellJob{CJ_CV_SUM, t, f}); cv_apply_jobs.push_back(CellJob{CJ_CV_APPLY, t, f}); }
    }
    cell_ncvjobs_ = (int)cv_sum_jobs.size();
    cell_cv_sum_jobs_ = upload(cv_sum_jobs); cell_cv_apply_jobs_ = upload(cv_apply_jobs);
    C.n_cv = S.n_cv;
    C.cv = dalloc<u64>(R * std::max(1, S.n_cv));
    HIP_TRY(hipMemset(C.cv, 0, R * std::max(1, S.n_cv) * sizeof(u64)));
    cell_nslots_ = slot;
    cell_nwin_ = (natoms_ + AWSEM_WIN - 1) / AWSEM_WIN;
    if ((size_t)cell_nslots_ * (cell_nwin_ + 1) != cell_wstart_.size()) throw std::string("internal error: window tables out of step with the bonded terms");
    cell_wstart_dev_ = upload(cell_wstart_);
    cell_njobs_ = (int)jobs.size();
    cell_jobs_ = upload(jobs);
    S.lists = nullptr; S.list_stride = 0; S.n_lists = 0; S.skin_half2 = C.skin_half2;
}
template <class real> void Engine<real>::cell_hb_sync() {
    const int nb = dsys_beta_n_, np = dsys_pap_n_, n = nb + np;
    if (!n) return;
    std::vector<HbRows> rows(n);
    std::vector<real> rc2(n);
    for (int t = 0; t < nb; t++) { rows[t] = cell_hb_beta_[t]; rc2[t] = cell_hb_rc2_beta_[t]; }
    for (int t = 0; t < np; t++) { rows[nb + t] = cell_hb_pap_[t]; rc2[nb + t] = cell_hb_rc2_pap_[t]; }
    if (!cell_hb_.rows) { cell_hb_.rows = dalloc<HbRows>(n); cell_hb_.rc2 = dalloc<real>(n); }
    HIP_TRY(hipStreamSynchronize(stream_));   // nothing in flight reads the tables while they change
    HIP_TRY(hipMemcpy(const_cast<HbRows*>(cell_hb_.rows), rows.data(), n * sizeof(HbRows), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(const_cast<real*>(cell_hb_.rc2), rc2.data(), n * sizeof(real), hipMemcpyHostToDevice));
}
// neighbour rows: `cell_per_site_` entries per contact site, a quarter of that (at least 16) per excluded-volume atom
template <class real> void Engine<real>::cell_alloc_lists() {
    CellSys<real>& C = cell_;
    C.bin[0].stride = (cell_per_site_ + AWSEM_ROW - 1) / AWSEM_ROW * AWSEM_ROW;
    C.bin[1].stride = std::max(AWSEM_ROW, (cell_per_site_ / 4 + AWSEM_ROW - 1) / AWSEM_ROW * AWSEM_ROW);
    for (int b = 0; b < 2; b++) {
        const size_t n = (size_t)std::max(1, C.bin[b].n) * C.bin[b].stride;
        if (n > 0x7fffffffULL) throw std::string("cell-list path: neighbour rows exceed 2^31 entries per replica");
        if (C.bin[b].rows) { HIP_TRY(hipStreamSynchronize(stream_)); dfree(C.bin[b].rows); }
        C.bin[b].rows = dalloc<unsigned>(n * nrep_);
    }
    auto hb_alloc = [&](HbRows& Hb, int nacc, int ndon) {
        Hb.stride = C.bin[0].stride;
        if (!Hb.spheres) {   // (a term without donors has no block: nothing is read or written for it)
            if (ndon < 0 || ndon > (1 << 28)) throw std::string("cell-list path: donor count out of range");
            Hb.nblocks = (ndon + AWSEM_HB_BLOCK - 1) / AWSEM_HB_BLOCK;
            Hb.spheres = dalloc<float>((size_t)std::max(1, Hb.nblocks) * 4 * nrep_);
        }
        const size_t n = (size_t)std::max(1, nacc) * Hb.stride;
        if (n > 0x7fffffffULL) throw std::string("cell-list path: hydrogen-bond rows exceed 2^31 entries per replica");
        if (Hb.rows) { HIP_TRY(hipStreamSynchronize(stream_)); dfree(Hb.rows); }
        Hb.rows = dalloc<unsigned>(n * nrep_);
        if (!Hb.row_count) { Hb.row_count = dalloc<int>((size_t)std::max(1, nacc) * nrep_); HIP_TRY(hipMemset(Hb.row_count, 0, (size_t)std::max(1, nacc) * nrep_ * sizeof(int))); }
    };
    cell_hb_maxblocks_ = 1;
    for (int t = 0; t < dsys_beta_n_; t++) { hb_alloc(cell_hb_beta_[t], cell_nacc_beta_[t], cell_ndon_beta_[t]); cell_hb_maxblocks_ = std::max(cell_hb_maxblocks_, cell_hb_beta_[t].nblocks); }
    for (int t = 0; t < dsys_pap_n_; t++) { hb_alloc(cell_hb_pap_[t], cell_nacc_pap_[t], cell_ndon_pap_[t]); cell_hb_maxblocks_ = std::max(cell_hb_maxblocks_, cell_hb_pap_[t].nblocks); }
    cell_hb_sync();
    cell_rows_gen_++;
    cell_stale_ = true; lists_valid_ = false;
}
// member records, displacement test, bounding box (and, when asked, the virtual sites)
template <class real> void Engine<real>::cell_pre(bool step) {
    const CellSys<real>& C = cell_;
    const dim3 g(C.nblk_pre, nrep_), b(256);
    cell_cur_ ^= 1;
    if (step) hipLaunchKernelGGL((cell_pre_kernel<true, real>), g, b, 0, stream_, dsys_, C, pos_, cell_cur_);
    else hipLaunchKernelGGL((cell_pre_kernel<false, real>), g, b, 0, stream_, dsys_, C, pos_, cell_cur_);
}
// the list build for the replicas that need it (kernels of other replicas exit at once)
template <class real> void Engine<real>::cell_refresh(const CellSys<real>& C) {
    const int force = cell_stale_ ? 1 : 0, cur = cell_cur_;
    const long long nm = std::max(C.bin[0].n, C.bin[1].n);
    const dim3 gm((unsigned)std::max(1LL, std::min((nm + 255) / 256, 4096LL)), 2 * nrep_), b(256);
    const dim3 gr((unsigned)std::max(1LL, std::min((nm * AWSEM_ROW + 255) / 256, 16384LL)), 2 * nrep_);
    // Round 3: the sort (geometry, count, scan, fill) is one launch of 32 workgroups per (replica, bin) that synchronise among
    // themselves (cell_sort_kernel: all of them must be resident at once, which the replica count bounds); the rows follow as before.
    // cell_one_step does without these launches altogether (the sort rides in the window kernel, the rows are built where they are walked).
    if (cell_sort_ok()) {
        hipLaunchKernelGGL(cell_sort_kernel<real>, dim3(AWSEM_SORT_G, 2 * nrep_), b, 0, stream_, C, (const real*)pos_, cur, force);
        hipLaunchKernelGGL(cell_rows_kernel<real>, gr, b, 0, stream_, dsys_, C, cur, force);
        if (cell_hb_.n_beta + cell_hb_.n_pap) {
            const dim3 gh((unsigned)std::max(1LL, std::min(((long long)cell_hb_maxacc_ * AWSEM_ROW + 255) / 256, 16384LL)), nrep_, cell_hb_.n_beta + cell_hb_.n_pap);
            cell_hb_prepare(C, cur, force);
            hipLaunchKernelGGL(cell_hb_rows_kernel<real>, gh, b, 0, stream_, cell_hb_, C, pos_, cur, force);
        }
        HIP_TRY(hipGetLastError());
        cell_stale_ = false;
        return;
    }
    hipLaunchKernelGGL(cell_geom_kernel<real>, dim3(2 * nrep_), dim3(1024), 0, stream_, C, cur, force);
    hipLaunchKernelGGL(cell_count_kernel<real>, dim3(std::min(gm.x, 64u), gm.y), b, 0, stream_, C, pos_, cur, force);   // few workgroups: see the tile histogram
    hipLaunchKernelGGL(cell_scan_kernel<real>, dim3(AWSEM_CELL_MAXCELLS / 1024, 2 * nrep_), dim3(1024), 0, stream_, C, cur, force);
    hipLaunchKernelGGL(cell_fill_kernel<real>, gm, b, 0, stream_, C, cur, force);
    hipLaunchKernelGGL(cell_rows_kernel<real>, gr, b, 0, stream_, dsys_, C, cur, force);
    if (cell_hb_.n_beta + cell_hb_.n_pap) {
        const dim3 gh((unsigned)std::max(1LL, std::min(((long long)cell_hb_maxacc_ * AWSEM_ROW + 255) / 256, 16384LL)), nrep_, cell_hb_.n_beta + cell_hb_.n_pap);
        cell_hb_prepare(C, cur, force);
        hipLaunchKernelGGL(cell_hb_rows_kernel<real>, gh, b, 0, stream_, cell_hb_, C, pos_, cur, force);
    }
    HIP_TRY(hipGetLastError());
    cell_stale_ = false;
}
template <class real> void Engine<real>::cell_forces_pass(bool energy, unsigned mask, const CellSys<real>* Cp, bool window, int rows_cur, int rows_force) {
    const CellSys<real>& C = Cp ? *Cp : cell_;
    const DevSys<real>& S = dsys_;
    const dim3 b(256);
    if (cell_nslots_ && window) {
        if (energy) hipLaunchKernelGGL((cell_window_kernel<true, false, real>), dim3(cell_nwin_, nrep_), b, 0, stream_, S, C, pos_, cell_wstart_dev_, cell_nwin_, mask);
        else hipLaunchKernelGGL((cell_window_kernel<false, false, real>), dim3(cell_nwin_, nrep_), b, 0, stream_, S, C, pos_, cell_wstart_dev_, cell_nwin_, mask);
    }
    auto launch_jobs = [&]() {
        if (cell_njobs_) {
            if (energy) hipLaunchKernelGGL((cell_bonded_kernel<true, real>), dim3(cell_njobs_, nrep_), b, 0, stream_, S, C, pos_, cell_jobs_, mask);
            else hipLaunchKernelGGL((cell_bonded_kernel<false, real>), dim3(cell_njobs_, nrep_), b, 0, stream_, S, C, pos_, cell_jobs_, mask);
        }
        if (cell_ncvjobs_) {
            if (energy) {
                hipLaunchKernelGGL((cell_bonded_kernel<true, real>), dim3(cell_ncvjobs_, nrep_), b, 0, stream_, S, C, pos_, cell_cv_sum_jobs_, mask);
                hipLaunchKernelGGL((cell_bonded_kernel<true, real>), dim3(cell_ncvjobs_, nrep_), b, 0, stream_, S, C, pos_, cell_cv_apply_jobs_, mask);
            } else {
                hipLaunchKernelGGL((cell_bonded_kernel<false, real>), dim3(cell_ncvjobs_, nrep_), b, 0, stream_, S, C, pos_, cell_cv_sum_jobs_, mask);
                hipLaunchKernelGGL((cell_bonded_kernel<false, real>), dim3(cell_ncvjobs_, nrep_), b, 0, stream_, S, C, pos_, cell_cv_apply_jobs_, mask);
            }
        }
    };
    // dynamics: the windows (which place the virtual sites) ride in the density launch below, so the item-by-item jobs, which may read
    // virtual sites, follow it
    const bool jobs_after = rows_cur >= 0 && cell_nslots_ > 0 && !window;
    if (!jobs_after) launch_jobs();
    auto row_blocks = [](int n) { return (int)std::max(1LL, std::min(((long long)n * AWSEM_ROW + 255) / 256, 16384LL)); };
    const bool contact_on = S.n_contact && ((mask >> S.contact[0].group) & 1u);
    const bool excl_on = S.n_excl && ((mask >> S.excl[0].group) & 1u);
    if (contact_on || excl_on) {
        const int nbc = contact_on ? row_blocks(C.bin[0].n) : 0;
        const int nbx = excl_on ? (int)std::max(1LL, std::min(((long long)C.bin[1].n * AWSEM_ROW_X + 255) / 256, 16384LL)) : 0;   // four lanes per excluded-volume row
        const bool win = rows_cur >= 0 && cell_nslots_ > 0;   // dynamics: the bonded windows ride in this launch (cell_one_step)
        if (energy) hipLaunchKernelGGL((cell_density_excl_kernel<true, false, real>), dim3(nbc + nbx, nrep_), b, 0, stream_, S, C, pos_, nbc, nbx, mask, rows_cur, rows_force, (const int*)nullptr, 0);
        else if (win) hipLaunchKernelGGL((cell_density_excl_kernel<false, true, real>), dim3(nbc + nbx + cell_nwin_, nrep_), b, 0, stream_, S, C, pos_, nbc, nbx, mask, rows_cur, rows_force, cell_wstart_dev_, cell_nwin_);
        else hipLaunchKernelGGL((cell_density_excl_kernel<false, false, real>), dim3(nbc + nbx, nrep_), b, 0, stream_, S, C, pos_, nbc, nbx, mask, rows_cur, rows_force, (const int*)nullptr, 0);
    }
    if (jobs_after) launch_jobs();
    if (contact_on) {
        const dim3 g(row_blocks(C.bin[0].n), nrep_);
        if (!(S.contact[0].flags & 2)) {
            if (energy) hipLaunchKernelGGL((cell_pairs_kernel<true, real>), g, b, 0, stream_, S, C, pos_);
            else hipLaunchKernelGGL((cell_pairs_kernel<false, real>), g, b, 0, stream_, S, C, pos_);
        }
        hipLaunchKernelGGL(cell_chain_kernel<real>, g, b, 0, stream_, S, C, pos_);
    }
    if (cell_hb_.n_beta + cell_hb_.n_pap) {
        const dim3 g(row_blocks(cell_hb_maxacc_), nrep_, cell_hb_.n_beta + cell_hb_.n_pap);
        if (rows_cur >= 0) cell_hb_prepare(C, rows_cur, rows_force);   // dynamics: the evaluation kernel builds the rows it walks on a rebuild step (otherwise cell_refresh has prepared)
        if (energy) hipLaunchKernelGGL((cell_hb_eval_kernel<true, real>), g, b, 0, stream_, cell_hb_, C, pos_, mask, rows_cur, rows_force);
        else hipLaunchKernelGGL((cell_hb_eval_kernel<false, real>), g, b, 0, stream_, cell_hb_, C, pos_, mask, rows_cur, rows_force);
    }
    HIP_TRY(hipGetLastError());
}
template <class real> void Engine<real>::cell_evaluate(bool energy, unsigned mask, int redistribute, double* e, double* f) {
    const size_t R = nrep_;
    synchronize();   // drains pending awsem_step launches and raises THEIR status (a row overflow during dynamics) before the word is reused here
    for (int attempt = 0;; attempt++) {
        HIP_TRY(hipEventRecord(ev0_, stream_));
        HIP_TRY(hipMemsetAsync(cell_.f, 0, R * 3 * natoms_ * sizeof(u64), stream_));
        if (energy) HIP_TRY(hipMemsetAsync(cell_.eg, 0, R * AWSEM_NGROUPS * sizeof(u64), stream_));
        cell_pre(false);
        cell_refresh(cell_);
        cell_forces_pass(energy, mask);
        if (f && redistribute) hipLaunchKernelGGL(cell_redistribute_kernel<real>, cell_grid(dsys_.nvs), dim3(256), 0, stream_, dsys_, cell_, pos_);
        hipLaunchKernelGGL(cell_output_kernel<real>, cell_grid(natoms_), dim3(256), 0, stream_, cell_, f ? dF_ : (double*)nullptr, e ? dE_ : (double*)nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev1_, stream_));
        HIP_TRY(hipStreamSynchronize(stream_));
        record_ms();
        int st = 0;
        HIP_TRY(hipMemcpy(&st, dstatus_, sizeof st, hipMemcpyDeviceToHost));
        if (!(st & 2)) break;
        HIP_TRY(hipMemset(dstatus_, 0, sizeof st));
        if (attempt >= 5) throw std::string("cell-list path: neighbour rows keep overflowing (more than 2048 partners per site)");
        cell_per_site_ *= 2;   // denser than expected: reserve more and rebuild
        cell_alloc_lists();
    }
    if (e) HIP_TRY(hipMemcpy(e, dE_, sizeof(double) * nrep_ * AWSEM_NGROUPS, hipMemcpyDeviceToHost));
    if (f) HIP_TRY(hipMemcpy(f, dF_, sizeof(double) * 3 * natoms_ * nrep_, hipMemcpyDeviceToHost));
}
// one MD step on the stream; `offset` is added to the device-resident step base by the integrator
template <class real> void Engine<real>::cell_one_step(double dt, double friction, uint64_t seed, int offset) {
    if (!cell_fused_) {   // AWSEM_CELL_FUSED=0: the eleven-launch step of rounds 1 and 2
        cell_pre(true);
        cell_refresh(cell_);
        cell_forces_pass(false, 0xffffffffu);
        hipLaunchKernelGGL(cell_integrate_kernel<real>, cell_grid(dsys_.nmassive), dim3(256), 0, stream_, dsys_, cell_, pos_, vel(), dT_, dt, friction,
                           (unsigned long long)seed, (const long long*)cell_step_base_, offset, -1, (u64*)nullptr);
        return;
    }
    // Round 3, five launches: sort (64 small workgroups; leaves at once unless a member moved skin/2) -> density + excluded volume, which
    // build the rows they walk when the flag is set, with the bonded windows (which place the virtual sites) in the same launch -> pair
    // pass -> chain rule [-> hydrogen bonds, rows built likewise] -> integrator, which also leaves the member records, the displacement
    // flag and the bounding boxes of the NEXT step and clears that step's virtual-site accumulators (the force arrays alternate with
    // the flag parity).  The launch-start cell_pre(true) of cell_step stands in for a preceding integrator.
    const size_t fstride = (size_t)nrep_ * 3 * natoms_;
    const int cur = cell_cur_;
    CellSys<real> C = cell_;
    C.f = cell_.f + (size_t)cur * fstride;
    const int force = cell_stale_ ? 1 : 0;
    // a replica whose contact or excluded-volume term is switched off would never have its rows rebuilt where they are walked
    const bool inline_rows = cell_sort_ok() && (!dsys_.n_contact || dsys_.contact[0].group >= 0) && (!dsys_.n_excl || dsys_.excl[0].group >= 0);
    if (inline_rows) {
        // the sort: 64 small workgroups that leave at once on most steps; then the contact / excluded-volume / bonded launch builds the rows
        // where it walks them and places the virtual sites in its windows
        const bool win_rides = cell_nslots_ > 0 && (dsys_.n_contact > 0 || dsys_.n_excl > 0);
        if (!win_rides) {
            if (cell_nslots_) hipLaunchKernelGGL((cell_window_kernel<false, true, real>), dim3(cell_nwin_, nrep_), dim3(256), 0, stream_, dsys_, C, pos_, cell_wstart_dev_, cell_nwin_, 0xffffffffu);
            else hipLaunchKernelGGL(cell_vsite_kernel<real>, cell_grid(dsys_.nvs), dim3(256), 0, stream_, dsys_, pos_);
        }
        hipLaunchKernelGGL(cell_sort_kernel<real>, dim3(AWSEM_SORT_G, 2 * nrep_), dim3(256), 0, stream_, C, (const real*)pos_, cur, force);
        cell_stale_ = false;
        cell_forces_pass(false, 0xffffffffu, &C, false, cur, force);
    } else {
        if (cell_nslots_) hipLaunchKernelGGL((cell_window_kernel<false, true, real>), dim3(cell_nwin_, nrep_), dim3(256), 0, stream_, dsys_, C, pos_, cell_wstart_dev_, cell_nwin_, 0xffffffffu);
        else hipLaunchKernelGGL(cell_vsite_kernel<real>, cell_grid(dsys_.nvs), dim3(256), 0, stream_, dsys_, pos_);
        cell_refresh(C);
        cell_forces_pass(false, 0xffffffffu, &C, false);
    }
    hipLaunchKernelGGL(cell_integrate_kernel<real>, cell_grid(dsys_.nmassive), dim3(256), 0, stream_, dsys_, C, pos_, vel(), dT_, dt, friction,
                       (unsigned long long)seed, (const long long*)cell_step_base_, offset, cur ^ 1, cell_.f + (size_t)(cur ^ 1) * fstride);
    cell_cur_ = cur ^ 1;
}
// A launch of `nsteps` is played as captured HIP graphs of AWSEM_CELL_CHUNK steps (the ~11 kernels of a step cost more
// to enqueue from the host than to run) plus single steps for the remainder; AWSEM_CELL_GRAPH=0 disables the graphs.
#define AWSEM_CELL_CHUNK 10
template <class real> void Engine<real>::cell_replay_if_overflowed() {
    const size_t nstate = (size_t)nrep_ * 3 * natoms_;
    for (int attempt = 0;; attempt++) {
        int st = 0;
        HIP_TRY(hipMemcpy(&st, dstatus_, sizeof st, hipMemcpyDeviceToHost));
        if (!(st & 2)) return;
        st &= ~2;
        HIP_TRY(hipMemcpy(dstatus_, &st, sizeof st, hipMemcpyHostToDevice));
        if (attempt >= 6 || cell_per_site_ > 8192) throw std::string("cell-list path: neighbour rows keep overflowing during dynamics (more than 8192 partners per site)");
        std::vector<PendingStep> pend;
        pend.swap(cell_pending_);
        HIP_TRY(hipMemcpy(pos_, cell_snap_pos_, nstate * sizeof(real), hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(vel(), cell_snap_vel_, nstate * sizeof(real), hipMemcpyDeviceToDevice));
        cell_per_site_ *= 2;
        cell_alloc_lists();   // marks the rows stale: the first replayed launch sizes and rebuilds them from the snapshot
        if (getenv("AWSEM_VERBOSE")) fprintf(stderr, "[awsem] neighbour row overflow during dynamics: rows regrown to %d per contact site, replaying %zu launch(es)\n", cell_per_site_, pend.size());
        cell_replaying_ = true;
        try {
            for (const PendingStep& p : pend) cell_step(p.nsteps, p.T.data(), p.dt, p.friction, p.seed, p.step0);
        } catch (...) { cell_replaying_ = false; throw; }
        cell_replaying_ = false;
        HIP_TRY(hipStreamSynchronize(stream_));
        record_ring_ms();
    }
}
template <class real> void Engine<real>::cell_step(int nsteps, const double* T, double dt, double friction, uint64_t seed, int64_t step0) {
    HIP_TRY(hipSetDevice(device_));
    if (!cell_step_base_) cell_step_base_ = dalloc<long long>(1);
    if (npend_ == NRING) synchronize();   // here, not inside begin_async below: this launch must not be on the pending list while older ones are resolved
    if (cell_stale_) {
        // first launch after the coordinates were set: build the rows now and leave head room (twice the fullest row), so that
        // the replay a later overflow costs (cell_replay_if_overflowed) stays a rare event
        synchronize();
        for (int attempt = 0;; attempt++) {
            cell_pre(false);
            cell_stale_ = true; lists_valid_ = false;
            cell_refresh(cell_);
            HIP_TRY(hipStreamSynchronize(stream_));
            int st = 0;
            HIP_TRY(hipMemcpy(&st, dstatus_, sizeof st, hipMemcpyDeviceToHost));
            int fullest = 0;
            if (!(st & 2) && cell_.bin[0].n) {
                std::vector<int> cnt((size_t)cell_.bin[0].n * nrep_);
                HIP_TRY(hipMemcpy(cnt.data(), cell_.bin[0].row_count, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
                for (int c : cnt) fullest = std::max(fullest, c);
            }
            auto hb_fullest = [&](const HbRows& Hb, int nacc) {
                std::vector<int> cnt((size_t)nacc * nrep_);
                if (cnt.empty()) return;
                HIP_TRY(hipMemcpy(cnt.data(), Hb.row_count, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
                for (int c : cnt) fullest = std::max(fullest, c);
            };
            for (int t = 0; t < dsys_.n_beta && !(st & 2); t++) hb_fullest(cell_hb_beta_[t], dsys_.beta[t].nacc);
            for (int t = 0; t < dsys_.n_pap && !(st & 2); t++) hb_fullest(cell_hb_pap_[t], dsys_.pap[t].nacc);
            // AWSEM_CELL_HEADROOM (tests only): rows sized to that multiple of the fullest row; below 1 the rows are deliberately
            // left SHORTER than the start structure needs, so that the first rebuild of the run overflows and is replayed
            const double hr_env = getenv("AWSEM_CELL_HEADROOM") ? atof(getenv("AWSEM_CELL_HEADROOM")) : 2.0;
            const double headroom = std::max(1.0, hr_env);
            if (getenv("AWSEM_VERBOSE")) fprintf(stderr, "[awsem] sizing the neighbour rows: %d reserved per contact site (stride %d), fullest row %d%s\n", cell_per_site_, cell_.bin[0].stride, fullest, (st & 2) ? ", overflowed" : "");
            if (!(st & 2) && (int)std::ceil(headroom * fullest) <= cell_.bin[0].stride) {
                if (hr_env > 0.0 && hr_env < 1.0 && !cell_replaying_) { cell_per_site_ = std::max(16, (int)std::ceil(hr_env * fullest)); cell_alloc_lists(); }
                break;
            }
            HIP_TRY(hipMemset(dstatus_, 0, sizeof st));
            if (attempt >= 6) throw std::string("cell-list path: neighbour rows keep overflowing");
            cell_per_site_ = std::max(2 * cell_per_site_ * ((st & 2) ? 1 : 0), (int)std::ceil(headroom * fullest));
            cell_alloc_lists();
        }
        cell_stale_ = true; lists_valid_ = false;   // the step sequence below rebuilds once more through the ordinary path (virtual sites placed first)
    }
    if (T && cell_pending_.empty() && !cell_replaying_) {   // first launch since the last synchronisation: the state a replay would start from
        const size_t nstate = (size_t)nrep_ * 3 * natoms_;
        if (!cell_snap_pos_) { cell_snap_pos_ = dalloc<real>(nstate); cell_snap_vel_ = dalloc<real>(nstate); }
        HIP_TRY(hipMemcpyAsync(cell_snap_pos_, pos_, nstate * sizeof(real), hipMemcpyDeviceToDevice, stream_));
        HIP_TRY(hipMemcpyAsync(cell_snap_vel_, vel(), nstate * sizeof(real), hipMemcpyDeviceToDevice, stream_));
    }
    if (T) cell_pending_.push_back(PendingStep{nsteps, std::vector<double>(T, T + nrep_), dt, friction, seed, step0});
    else cell_pending_.clear();   // device-resident temperatures (replica exchange): the sweeps in between are not recorded, so such a launch is not replayable
    begin_async(T);
    static const bool use_graph = !(getenv("AWSEM_CELL_GRAPH") && atoi(getenv("AWSEM_CELL_GRAPH")) == 0);
    HIP_TRY(hipMemsetAsync(cell_.f, 0, (size_t)2 * nrep_ * 3 * natoms_ * sizeof(u64), stream_));
    if (cell_fused_ && nsteps > 0) { cell_pre(true); HIP_TRY(hipGetLastError()); }   // what the integrator of a preceding step would have left (cell_one_step)
    int s = 0;
    auto set_base = [&](long long v) { hipLaunchKernelGGL(cell_set_word_kernel<long long>, dim3(1), dim3(1), 0, stream_, cell_step_base_, v); };
    if (dsys_.remove_cm) {   // the momentum the first step removes (later steps get it from the integrator before them)
        set_base((long long)step0);
        hipLaunchKernelGGL(cell_momentum_kernel<real>, cell_grid(dsys_.nmassive), dim3(256), 0, stream_, dsys_, cell_, vel(), (const long long*)cell_step_base_);
    }
    while (s < nsteps) {
        // the rows are rebuilt unconditionally (a host-side decision baked into the launch) right after the coordinates were set
        if (use_graph && !cell_stale_ && nsteps - s >= AWSEM_CELL_CHUNK) {
            CellGraph& G = cell_graph_[cell_cur_];   // the flag parity is baked into the captured launches
            if (!G.exec || G.dt != dt || G.friction != friction || G.seed != seed || G.gen != cell_rows_gen_) {
                if (G.exec) { (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; }
                const int cur0 = cell_cur_;
                hipGraph_t graph = nullptr;
                {
                    std::lock_guard<std::mutex> hold(capture_mu_);   // keeps the reporters' thread (snapshot_end) off the stream's events meanwhile
                    HIP_TRY(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
                    try { for (int k = 0; k < AWSEM_CELL_CHUNK; k++) cell_one_step(dt, friction, seed, k); }
                    catch (...) { (void)hipStreamEndCapture(stream_, &graph); if (graph) (void)hipGraphDestroy(graph); cell_cur_ = cur0; throw; }
                    HIP_TRY(hipStreamEndCapture(stream_, &graph));
                }
                HIP_TRY(hipGraphInstantiate(&G.exec, graph, nullptr, nullptr, 0));
                HIP_TRY(hipGraphDestroy(graph));
                cell_cur_ = cur0;   // capturing ran nothing; AWSEM_CELL_CHUNK is even, so a replay ends on the parity it started on
                G.dt = dt; G.friction = friction; G.seed = seed; G.gen = cell_rows_gen_;
            }
            set_base((long long)(step0 + s));
            HIP_TRY(hipGraphLaunch(G.exec, stream_));
            s += AWSEM_CELL_CHUNK;
        } else {
            set_base((long long)(step0 + s));
            cell_one_step(dt, friction, seed, 0);
            s++;
        }
    }
    hipLaunchKernelGGL(cell_vsite_kernel<real>, cell_grid(dsys_.nvs), dim3(256), 0, stream_, dsys_, pos_);
    HIP_TRY(hipGetLastError());
    end_async();
}
// FIRE on the cell-list path: the resident kernel's iteration (awsem_minimize_kernel) as a sequence of grid-wide kernels;
// the host looks at the per-replica `done` flags every 25 iterations.
template <class real> void Engine<real>::cell_minimize(double tolerance, int max_iterations, int32_t* iterations, double* rms) {
    HIP_TRY(hipSetDevice(device_));
    synchronize();
    const size_t R = nrep_, nstate = R * 3 * natoms_;
    real* fv = dalloc<real>(nstate);                               // FIRE velocities (the trajectory's velocities stay untouched)
    FireState<real>* st = dalloc<FireState<real>>(2 * R);
    u64* sums = dalloc<u64>(2 * 3 * R);
    std::vector<FireState<real>> h(2 * R);
    for (auto& s : h) { s.dt = (real)0.001; s.alpha = (real)0.1; s.npos = 0; s.done = 0; s.iters = 0; s.pad = 0; s.rms = 0; }
    HIP_TRY(hipMemcpy(st, h.data(), h.size() * sizeof(FireState<real>), hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(fv, 0, nstate * sizeof(real), stream_));
    HIP_TRY(hipMemsetAsync(sums, 0, 2 * 3 * R * sizeof(u64), stream_));
    HIP_TRY(hipMemsetAsync(cell_.f, 0, nstate * sizeof(u64), stream_));
    HIP_TRY(hipEventRecord(ev0_, stream_));
    cell_stale_ = true; lists_valid_ = false;
    const dim3 g = cell_grid(dsys_.nmassive), b(256);
    int it = 0;
    bool all_done = false;
    while (it < max_iterations && !all_done) {
        const int chunk = std::min(25, max_iterations - it);
        for (int c = 0; c < chunk; c++, it++) {
            const int p = it & 1;
            cell_pre(true);
            cell_refresh(cell_);
            cell_forces_pass(false, 0xffffffffu);
            hipLaunchKernelGGL(cell_fire_sums_kernel<real>, g, b, 0, stream_, dsys_, cell_, pos_, fv, sums + 3 * R * p, st + R * p);
            hipLaunchKernelGGL(cell_fire_update_kernel<real>, g, b, 0, stream_, dsys_, cell_, pos_, fv, sums + 3 * R * p, sums + 3 * R * (p ^ 1),
                               st + R * p, st + R * (p ^ 1), tolerance);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream_));
        int stw = 0;
        HIP_TRY(hipMemcpy(&stw, dstatus_, sizeof stw, hipMemcpyDeviceToHost));
        if (stw & 2) {   // a neighbour row overflowed while the structure relaxed: reserve more and carry on (the rows are rebuilt)
            HIP_TRY(hipMemset(dstatus_, 0, sizeof stw));
            cell_per_site_ *= 2;
            if (cell_per_site_ > 8192) throw std::string("cell-list path: neighbour rows keep overflowing during the minimisation");
            cell_alloc_lists();
        }
        HIP_TRY(hipMemcpy(h.data(), st + R * (it & 1), R * sizeof(FireState<real>), hipMemcpyDeviceToHost));
        all_done = true;
        for (size_t r = 0; r < R; r++) all_done = all_done && h[r].done;
    }
    hipLaunchKernelGGL(cell_vsite_kernel<real>, cell_grid(dsys_.nvs), dim3(256), 0, stream_, dsys_, pos_);
    HIP_TRY(hipMemsetAsync(cell_.f, 0, nstate * sizeof(u64), stream_));
    HIP_TRY(hipEventRecord(ev1_, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    record_ms();
    HIP_TRY(hipMemcpy(h.data(), st + R * (it & 1), R * sizeof(FireState<real>), hipMemcpyDeviceToHost));
    for (size_t r = 0; r < R; r++) { if (iterations) iterations[r] = h[r].iters; if (rms) rms[r] = h[r].rms; }
    dfree(fv); dfree(st); dfree(sums);
    cell_stale_ = true; lists_valid_ = false;
}
template <class real> int Engine<real>::cell_neighbors(int rep, int kind, int index, double skin, int32_t* pairs, int capacity) {
    const bool hb = kind == AWSEM_LIST_BETA || kind == AWSEM_LIST_PAP;
    synchronize();   // as in cell_evaluate: a pending launch's status is reported, not mistaken for this call's
    if (!hb && (index != 0 || (kind != AWSEM_LIST_EXCL && kind != AWSEM_LIST_CONTACT))) throw std::string("neighbors (cell-list path): kind must be AWSEM_LIST_EXCL / _CONTACT (index 0), _BETA or _PAP");
    if ((kind == AWSEM_LIST_EXCL && !dsys_.n_excl) || (kind == AWSEM_LIST_CONTACT && !dsys_.n_contact) || index < 0 ||
        (kind == AWSEM_LIST_BETA && index >= dsys_.n_beta) || (kind == AWSEM_LIST_PAP && index >= dsys_.n_pap)) throw std::string("neighbors: no such term");
    const int b = kind == AWSEM_LIST_EXCL ? 1 : 0;
    // the list as the caller defines it: screening distance + skin (the rows of a run use their own margins)
    real* hb_rc2 = kind == AWSEM_LIST_BETA ? &cell_hb_rc2_beta_[index] : &cell_hb_rc2_pap_[index];
    const real hb_saved = hb ? *hb_rc2 : (real)0;
    double r = (b ? cell_rc_excl_ : cell_rc_contact_) + skin;
    if (hb) {
        r = (kind == AWSEM_LIST_BETA ? std::sqrt((double)dsys_.beta[index].screen2) : std::sqrt((double)dsys_.pap[index].cutoff2)) + skin;
        *hb_rc2 = (real)(r * r);
        cell_hb_sync();
    }
    for (int attempt = 0;; attempt++) {
        CellSys<real> C = cell_;
        if (!hb) {
            C.bin[b].edge_min = (real)std::max((double)C.bin[b].edge_min, r);
            C.bin[b].rc2 = (real)(r * r);
        }
        cell_pre(false);
        cell_stale_ = true; lists_valid_ = false;
        cell_refresh(C);
        cell_stale_ = true; lists_valid_ = false;   // the rows now hold a non-standard cutoff: rebuild before the next evaluation
        HIP_TRY(hipStreamSynchronize(stream_));
        int st = 0;
        HIP_TRY(hipMemcpy(&st, dstatus_, sizeof st, hipMemcpyDeviceToHost));
        if (!(st & 2)) break;
        HIP_TRY(hipMemset(dstatus_, 0, sizeof st));
        if (attempt >= 5) { if (hb) { *hb_rc2 = hb_saved; cell_hb_sync(); } throw std::string("cell-list path: neighbour rows keep overflowing"); }
        cell_per_site_ *= 2;
        cell_alloc_lists();
    }
    std::vector<std::pair<int, int>> out;
    if (hb) {
        *hb_rc2 = hb_saved;
        cell_hb_sync();
        const HbRows& Hb = kind == AWSEM_LIST_BETA ? cell_hb_beta_[index] : cell_hb_pap_[index];
        const int nacc = kind == AWSEM_LIST_BETA ? dsys_.beta[index].nacc : dsys_.pap[index].nacc;
        std::vector<int> cnt(nacc);
        std::vector<unsigned> rows((size_t)nacc * Hb.stride);
        if (nacc) {
            HIP_TRY(hipMemcpy(cnt.data(), Hb.row_count + (size_t)rep * nacc, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(rows.data(), Hb.rows + (size_t)rep * nacc * Hb.stride, rows.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        }
        for (int i = 0; i < nacc; i++)
            for (int s = 0; s < cnt[i]; s++) out.push_back({i, (int)rows[(size_t)i * Hb.stride + s]});   // (acceptor index, donor index)
    } else {
        const CellBin<real>& B = cell_.bin[b];
        std::vector<int> cnt(B.n), sorted(B.n);
        std::vector<unsigned> rows((size_t)B.n * B.stride);
        if (B.n) {
            HIP_TRY(hipMemcpy(cnt.data(), B.row_count + (size_t)rep * B.n, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(sorted.data(), B.sorted + (size_t)rep * B.n, sorted.size() * sizeof(int), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(rows.data(), B.rows + (size_t)rep * B.n * B.stride, rows.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        }
        for (int k = 0; k < B.n; k++)   // row k belongs to member sorted[k]
            for (int s = 0; s < cnt[k]; s++) {
                const int i = sorted[k];
                const int j = (int)(rows[(size_t)k * B.stride + s] & 0xffffffu);
                if (b == 0) { if (i < j) out.push_back({i, j}); }   // every pair sits in both rows: keep the one of the lower site
                else { const int a = excl_sorted_atom_[i], c = excl_sorted_atom_[j]; if (a < c) out.push_back({a, c}); }
            }
    }
    std::sort(out.begin(), out.end());
    const int n = (int)out.size();
    for (int k = 0; k < n && k < capacity; k++) { pairs[2 * k] = out[k].first; pairs[2 * k + 1] = out[k].second; }
    return n;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
struct awsem_context {
    int natoms, nrep, device, precision;
    bool finalized = false;
    HostSystem H;
    std::unique_ptr<EngineBase> eng;
    int64_t step = 0;
};

#define GUARD_BEGIN try {
#define GUARD_END                                                                                              \
    }                                                                                                          \
    catch (const HipError& e) { return fail(AWSEM_EDEVICE, "HIP error %d (%s) in %s", (int)e.e, hipGetErrorString(e.e), e.what); } \
    catch (const std::string& s) { return fail(AWSEM_ELIMIT, "%s", s.c_str()); }                               \
    catch (const StateError& e) { return fail(AWSEM_ESTATE, "%s", e.what); }                                   \
    catch (const std::bad_alloc&) { return fail(AWSEM_ENOMEM, "out of host memory"); }

#define NEED_BUILD(h, group)                                                              \
    if (!(h)) return fail(AWSEM_EINVAL, "null handle");                                   \
    if ((h)->finalized) return fail(AWSEM_ESTATE, "system already finalized");            \
    if ((group) < 0 || (group) >= AWSEM_NGROUPS) return fail(AWSEM_EINVAL, "force group out of range 0..31");
#define NEED_RUN(h)                                                                       \
    if (!(h)) return fail(AWSEM_EINVAL, "null handle");                                   \
    if (!(h)->finalized) return fail(AWSEM_ESTATE, "system not finalized");

static bool atoms_ok(awsem_handle h, const int32_t* idx, size_t n, bool allow_minus1 = false) {
    for (size_t i = 0; i < n; i++) {
        if (idx[i] < 0) { if (!allow_minus1 || idx[i] != -1) return false; }
        else if (idx[i] >= h->natoms) return false;
    }
    return true;
}
template <class T> static std::vector<T> vec(const T* p, size_t n) { return p ? std::vector<T>(p, p + n) : std::vector<T>(); }

extern "C" {

const char* awsem_last_error(void) { return g_err; }
const char* awsem_version(void) { return "awsem_hip 0.1 gfx950"; }

int awsem_create(int natoms, int n_replicas, int device, int precision, awsem_handle* out) {
    if (natoms <= 0 || n_replicas <= 0 || !out) return fail(AWSEM_EINVAL, "awsem_create: natoms and n_replicas must be positive");
    if (precision != AWSEM_FP32 && precision != AWSEM_FP64) return fail(AWSEM_EINVAL, "awsem_create: precision must be AWSEM_FP32 or AWSEM_FP64");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(AWSEM_EDEVICE, "awsem_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(AWSEM_EINVAL, "awsem_create: device %d out of range (%d devices)", device, ndev);
    GUARD_BEGIN
    awsem_context* h = new awsem_context();
    h->natoms = natoms; h->nrep = n_replicas; h->device = device; h->precision = precision;
    h->H.natoms = natoms;
    h->H.mass.assign(natoms, 1.0);
    *out = h;
    return AWSEM_OK;
    GUARD_END
}

void awsem_destroy(awsem_handle h) { delete h; }

int awsem_set_masses(awsem_handle h, const double* m, int natoms) {
    NEED_BUILD(h, 0);
    if (!m || natoms != h->natoms) return fail(AWSEM_EINVAL, "set_masses: natoms mismatch");
    for (int i = 0; i < natoms; i++) if (!(m[i] >= 0)) return fail(AWSEM_EINVAL, "set_masses: negative mass");
    h->H.mass.assign(m, m + natoms);
    return AWSEM_OK;
}
int awsem_set_cm_motion_removal(awsem_handle h, int enabled) {
    NEED_BUILD(h, 0);
    h->H.remove_cm = enabled != 0;
    return AWSEM_OK;
}
int awsem_set_virtual_sites(awsem_handle h, int n, const int32_t* site, const int32_t* par, const double* w) {
    NEED_BUILD(h, 0);
    if (n < 0 || (n && (!site || !par || !w))) return fail(AWSEM_EINVAL, "set_virtual_sites: bad arguments");
    if (!atoms_ok(h, site, n) || !atoms_ok(h, par, 3 * (size_t)n)) return fail(AWSEM_EINVAL, "set_virtual_sites: atom index out of range");
    h->H.vs_site = vec(site, n); h->H.vs_par = vec(par, 3 * (size_t)n); h->H.vs_w = vec(w, 3 * (size_t)n);
    return AWSEM_OK;
}
int awsem_add_harmonic_bonds(awsem_handle h, int group, int n, const int32_t* pairs, const double* r0, const double* k) {
    NEED_BUILD(h, group);
    if (n < 0 || (n && (!pairs || !r0 || !k)) || !atoms_ok(h, pairs, 2 * (size_t)n)) return fail(AWSEM_EINVAL, "add_harmonic_bonds: bad arguments");
    h->H.harm.push_back(HHarm{group, vec(pairs, 2 * (size_t)n), vec(r0, n), vec(k, n)});
    return AWSEM_OK;
}
int awsem_add_chi(awsem_handle h, int group, int n, const int32_t* atoms, double k, double chi0) {
    NEED_BUILD(h, group);
    if (n < 0 || (n && !atoms) || !atoms_ok(h, atoms, 4 * (size_t)n)) return fail(AWSEM_EINVAL, "add_chi: bad arguments");
    h->H.chi.push_back(HChi{group, vec(atoms, 4 * (size_t)n), k, chi0});
    return AWSEM_OK;
}
int awsem_add_rama(awsem_handle h, int group, int n, const int32_t* atoms, int nwell, const double* wp, const double* bw, double k) {
    NEED_BUILD(h, group);
    if (n < 0 || nwell < 1 || nwell > AWSEM_MAX_WELLS || !wp || (n && !atoms) || !atoms_ok(h, atoms, 5 * (size_t)n))
        return fail(AWSEM_EINVAL, "add_rama: bad arguments (1..%d wells)", AWSEM_MAX_WELLS);
    HRama r{group, nwell, vec(atoms, 5 * (size_t)n), vec(wp, 6 * (size_t)nwell), {}, k};
    r.bw = bw ? vec(bw, (size_t)n * nwell) : std::vector<double>((size_t)n * nwell, 1.0);
    h->H.rama.push_back(std::move(r));
    return AWSEM_OK;
}
int awsem_add_excl(awsem_handle h, int group, int nsites, const int32_t* atom, const int32_t* cls, int nexcl, const int32_t* excl, double k, double r0) {
    NEED_BUILD(h, group);
    if (nsites < 0 || nexcl < 0 || (nsites && (!atom || !cls)) || (nexcl && !excl) || !atoms_ok(h, atom, nsites) || !atoms_ok(h, excl, 2 * (size_t)nexcl))
        return fail(AWSEM_EINVAL, "add_excl: bad arguments");
    for (int i = 0; i < nsites; i++) if (cls[i] < 0 || cls[i] >= 8) return fail(AWSEM_EINVAL, "add_excl: class out of range 0..7");
    h->H.excl.push_back(HExcl{group, vec(atom, nsites), vec(cls, nsites), vec(excl, 2 * (size_t)nexcl), k, r0});
    return AWSEM_OK;
}
int awsem_add_contact(awsem_handle h, int group, int nsites, const int32_t* atom, const int32_t* restype, const int32_t* resid, const int32_t* chain,
                      int env, int min_seq_sep, int flags, const double* dparams, const double* gamma, const double* bgamma) {
    NEED_BUILD(h, group);
    if (nsites < 0 || !dparams || (nsites && (!atom || !restype || !resid || !chain)) || !atoms_ok(h, atom, nsites) || env < 0 || env > 1)
        return fail(AWSEM_EINVAL, "add_contact: bad arguments");
    if (!(flags & AWSEM_CONTACT_BURIAL_ONLY) && !gamma) return fail(AWSEM_EINVAL, "add_contact: gamma table required");
    if ((flags & AWSEM_CONTACT_MEMBRANE_BURIAL) && (!(flags & AWSEM_CONTACT_ZDEP) || !bgamma))
        return fail(AWSEM_EINVAL, "add_contact: membrane burial wells need the z-dependent variant and a [2][20][3] burial table");
    const bool indexed = flags & AWSEM_CONTACT_INDEX_BASED;
    if (indexed && (flags & (AWSEM_CONTACT_MEMBRANE_BURIAL | AWSEM_CONTACT_BURIAL_ONLY)))
        return fail(AWSEM_EINVAL, "add_contact: the index-based variant has its own burial table and no burial-only form");
    for (int i = 0; i < nsites; i++) if (restype[i] < 0 || restype[i] >= 20) return fail(AWSEM_EINVAL, "add_contact: residue type out of range 0..19");
    HContact c;
    c.group = group; c.env = env; c.min_seq_sep = min_seq_sep; c.flags = flags; c.has_burial = bgamma != nullptr;
    c.atom = vec(atom, nsites); c.restype = vec(restype, nsites); c.resid = vec(resid, nsites); c.chain = vec(chain, nsites);
    memcpy(c.p, dparams, sizeof c.p);
    if (indexed) {   // tables by site index: [3][nsites][nsites] ([2][3][nsites][nsites] for the z-dependent blend: water, membrane) and [nsites][3]
        c.gamma = vec(gamma, ((flags & AWSEM_CONTACT_ZDEP) ? 6 : 3) * (size_t)nsites * nsites);
        c.bgamma = bgamma ? vec(bgamma, 3 * (size_t)nsites) : std::vector<double>(3 * (size_t)nsites, 0.0);
        if (c.gamma.empty()) c.gamma.assign(3, 0.0);
        if (c.bgamma.empty()) c.bgamma.assign(3, 0.0);
    } else {
        c.gamma = gamma ? vec(gamma, 2400) : std::vector<double>(2400, 0.0);
        c.bgamma = bgamma ? vec(bgamma, (flags & AWSEM_CONTACT_MEMBRANE_BURIAL) ? 120 : 60) : std::vector<double>(60, 0.0);
        c.bgamma.resize(120, 0.0);
    }
    h->H.contact.push_back(std::move(c));
    return AWSEM_OK;
}
int awsem_add_hbond_beta(awsem_handle h, int group, int two_theta, int nacc, const int32_t* acc_atoms, const int32_t* acc_res, int ndon,
                         const int32_t* don_atoms, const int32_t* don_res, int nres, const double* lambda, double k, double r_ON, double s_NO,
                         double r_OH, double s_HO, double cutoff) {
    NEED_BUILD(h, group);
    if (nacc < 0 || ndon < 0 || nres <= 0 || !lambda || (nacc && (!acc_atoms || !acc_res)) || (ndon && (!don_atoms || !don_res)))
        return fail(AWSEM_EINVAL, "add_hbond_beta: bad arguments");
    if (!atoms_ok(h, acc_atoms, 3 * (size_t)nacc, true) || !atoms_ok(h, don_atoms, 3 * (size_t)ndon, true))
        return fail(AWSEM_EINVAL, "add_hbond_beta: atom index out of range");
    for (int i = 0; i < nacc; i++) {
        if (acc_res[i] < 0 || acc_res[i] >= nres || acc_atoms[3 * i] < 0) return fail(AWSEM_EINVAL, "add_hbond_beta: bad acceptor %d", i);
        if (two_theta && (acc_atoms[3 * i + 1] < 0 || acc_atoms[3 * i + 2] < 0)) return fail(AWSEM_EINVAL, "add_hbond_beta: acceptor %d needs a2,a3", i);
    }
    for (int i = 0; i < ndon; i++) {
        if (don_res[i] < 0 || don_res[i] >= nres || don_atoms[3 * i] < 0 || don_atoms[3 * i + 1] < 0) return fail(AWSEM_EINVAL, "add_hbond_beta: bad donor %d", i);
        if (two_theta && don_atoms[3 * i + 2] < 0) return fail(AWSEM_EINVAL, "add_hbond_beta: donor %d needs d3", i);
    }
    h->H.beta.push_back(HBeta{group, two_theta ? 1 : 0, nres, vec(acc_atoms, 3 * (size_t)nacc), vec(acc_res, nacc), vec(don_atoms, 3 * (size_t)ndon),
                              vec(don_res, ndon), vec(lambda, (size_t)nres * nres), k, r_ON, s_NO, r_OH, s_HO, cutoff, HFn{}});
    return AWSEM_OK;
}
int awsem_add_hbond_pap(awsem_handle h, int group, int nacc, const int32_t* acc_atoms, const int32_t* acc_res, int ndon, const int32_t* don_atoms,
                        const int32_t* don_res, int nres, const double* gamma, double k, double eta, double r0, double d_i4, double cutoff) {
    NEED_BUILD(h, group);
    if (nacc < 0 || ndon < 0 || nres <= 0 || !gamma || (nacc && (!acc_atoms || !acc_res)) || (ndon && (!don_atoms || !don_res)))
        return fail(AWSEM_EINVAL, "add_hbond_pap: bad arguments");
    if (!atoms_ok(h, acc_atoms, 2 * (size_t)nacc) || !atoms_ok(h, don_atoms, 2 * (size_t)ndon)) return fail(AWSEM_EINVAL, "add_hbond_pap: atom index out of range");
    for (int i = 0; i < nacc; i++) if (acc_res[i] < 0 || acc_res[i] >= nres) return fail(AWSEM_EINVAL, "add_hbond_pap: bad acceptor residue");
    for (int i = 0; i < ndon; i++) if (don_res[i] < 0 || don_res[i] >= nres) return fail(AWSEM_EINVAL, "add_hbond_pap: bad donor residue");
    h->H.pap.push_back(HPap{group, nres, vec(acc_atoms, 2 * (size_t)nacc), vec(acc_res, nacc), vec(don_atoms, 2 * (size_t)ndon), vec(don_res, ndon),
                            vec(gamma, (size_t)nres * nres), k, eta, r0, d_i4, cutoff, HFn{}});
    return AWSEM_OK;
}
static int check_fn(const char* who, int kind, bool beta, int nres, const int32_t* info, const double* coef) {
    if (!info || !coef) return fail(AWSEM_EINVAL, "%s: res_info and coef are required", who);
    if (beta ? (kind < AWSEM_HB_LAMBDA_1 || kind > AWSEM_HB_LAMBDA_3) : (kind != AWSEM_HB_GAMMA_PAP_1 && kind != AWSEM_HB_GAMMA_PAP_2))
        return fail(AWSEM_EINVAL, "%s: kind %d does not belong to this term", who, kind);
    for (int i = 0; i < nres; i++) {
        if (info[4 * i] < 0 || info[4 * i] >= 20) return fail(AWSEM_EINVAL, "%s: residue type out of range 0..19 at residue %d", who, i);
        if (info[4 * i + 1] < 0 || info[4 * i + 1] >= (1 << 14)) return fail(AWSEM_EINVAL, "%s: chain index out of range 0..16383 at residue %d", who, i);
    }
    return AWSEM_OK;
}
int awsem_add_hbond_beta_fn(awsem_handle h, int group, int two_theta, int nacc, const int32_t* acc_atoms, const int32_t* acc_res, int ndon,
                            const int32_t* don_atoms, const int32_t* don_res, int nres, int kind, const int32_t* res_info, const double* coef,
                            double k, double r_ON, double s_NO, double r_OH, double s_HO, double cutoff) {
    NEED_BUILD(h, group);
    if (nres <= 0) return fail(AWSEM_EINVAL, "add_hbond_beta_fn: bad arguments");
    if (int rc = check_fn("add_hbond_beta_fn", kind, true, nres, res_info, coef)) return rc;
    std::vector<double> none;   // no dense table
    if (nacc < 0 || ndon < 0 || (nacc && (!acc_atoms || !acc_res)) || (ndon && (!don_atoms || !don_res))) return fail(AWSEM_EINVAL, "add_hbond_beta_fn: bad arguments");
    if (!atoms_ok(h, acc_atoms, 3 * (size_t)nacc, true) || !atoms_ok(h, don_atoms, 3 * (size_t)ndon, true)) return fail(AWSEM_EINVAL, "add_hbond_beta_fn: atom index out of range");
    for (int i = 0; i < nacc; i++) {
        if (acc_res[i] < 0 || acc_res[i] >= nres || acc_atoms[3 * i] < 0) return fail(AWSEM_EINVAL, "add_hbond_beta_fn: bad acceptor %d", i);
        if (two_theta && (acc_atoms[3 * i + 1] < 0 || acc_atoms[3 * i + 2] < 0)) return fail(AWSEM_EINVAL, "add_hbond_beta_fn: acceptor %d needs a2,a3", i);
    }
    for (int i = 0; i < ndon; i++) {
        if (don_res[i] < 0 || don_res[i] >= nres || don_atoms[3 * i] < 0 || don_atoms[3 * i + 1] < 0) return fail(AWSEM_EINVAL, "add_hbond_beta_fn: bad donor %d", i);
        if (two_theta && don_atoms[3 * i + 2] < 0) return fail(AWSEM_EINVAL, "add_hbond_beta_fn: donor %d needs d3", i);
    }
    HBeta b{group, two_theta ? 1 : 0, nres, vec(acc_atoms, 3 * (size_t)nacc), vec(acc_res, nacc), vec(don_atoms, 3 * (size_t)ndon),
            vec(don_res, ndon), none, k, r_ON, s_NO, r_OH, s_HO, cutoff, HFn{}};
    b.fn.kind = kind; b.fn.info = vec(res_info, 4 * (size_t)nres); b.fn.coef = vec(coef, AWSEM_HB_FN_COEFS);
    h->H.beta.push_back(std::move(b));
    return AWSEM_OK;
}
int awsem_add_hbond_pap_fn(awsem_handle h, int group, int nacc, const int32_t* acc_atoms, const int32_t* acc_res, int ndon, const int32_t* don_atoms,
                           const int32_t* don_res, int nres, int kind, const int32_t* res_info, const double* coef, double k, double eta, double r0,
                           double d_i4, double cutoff) {
    NEED_BUILD(h, group);
    if (nres <= 0) return fail(AWSEM_EINVAL, "add_hbond_pap_fn: bad arguments");
    if (int rc = check_fn("add_hbond_pap_fn", kind, false, nres, res_info, coef)) return rc;
    if (nacc < 0 || ndon < 0 || (nacc && (!acc_atoms || !acc_res)) || (ndon && (!don_atoms || !don_res))) return fail(AWSEM_EINVAL, "add_hbond_pap_fn: bad arguments");
    if (!atoms_ok(h, acc_atoms, 2 * (size_t)nacc) || !atoms_ok(h, don_atoms, 2 * (size_t)ndon)) return fail(AWSEM_EINVAL, "add_hbond_pap_fn: atom index out of range");
    for (int i = 0; i < nacc; i++) if (acc_res[i] < 0 || acc_res[i] >= nres) return fail(AWSEM_EINVAL, "add_hbond_pap_fn: bad acceptor residue");
    for (int i = 0; i < ndon; i++) if (don_res[i] < 0 || don_res[i] >= nres) return fail(AWSEM_EINVAL, "add_hbond_pap_fn: bad donor residue");
    HPap p{group, nres, vec(acc_atoms, 2 * (size_t)nacc), vec(acc_res, nacc), vec(don_atoms, 2 * (size_t)ndon), vec(don_res, ndon),
           std::vector<double>(), k, eta, r0, d_i4, cutoff, HFn{}};
    p.fn.kind = kind; p.fn.info = vec(res_info, 4 * (size_t)nres); p.fn.coef = vec(coef, AWSEM_HB_FN_COEFS);
    h->H.pap.push_back(std::move(p));
    return AWSEM_OK;
}
int awsem_add_helical(awsem_handle h, int group, int n, const int32_t* atoms, const double* coef, double r_ON, double s_NO, double r_OH, double s_HO) {
    NEED_BUILD(h, group);
    if (n < 0 || (n && (!atoms || !coef)) || !atoms_ok(h, atoms, 3 * (size_t)n)) return fail(AWSEM_EINVAL, "add_helical: bad arguments");
    HHelical t{group, std::vector<int>(4 * (size_t)n, -1), vec(coef, n), r_ON, s_NO, r_OH, s_HO};
    for (int b = 0; b < n; b++) for (int k = 0; k < 3; k++) t.atoms[4 * (size_t)b + k] = atoms[3 * (size_t)b + k];
    h->H.helical.push_back(std::move(t));
    return AWSEM_OK;
}
int awsem_add_helical_z(awsem_handle h, int group, int n, const int32_t* atoms, const double* coef, const double* coef_m, double r_ON, double s_NO,
                        double r_OH, double s_HO, double eta, double z_m, double center) {
    NEED_BUILD(h, group);
    if (n < 0 || (n && (!atoms || !coef || !coef_m)) || !atoms_ok(h, atoms, 4 * (size_t)n)) return fail(AWSEM_EINVAL, "add_helical_z: bad arguments");
    HHelical t{group, vec(atoms, 4 * (size_t)n), vec(coef, n), r_ON, s_NO, r_OH, s_HO};
    t.coef_m = vec(coef_m, n); t.eta = eta; t.z_m = z_m; t.center = center;
    h->H.helical.push_back(std::move(t));
    return AWSEM_OK;
}
int awsem_add_fragment_memory(awsem_handle h, int group, int nb, const int32_t* pairs, int ntab, const double* table, double k, double dr) {
    NEED_BUILD(h, group);
    if (nb < 0 || ntab < 2 || !(dr > 0) || (nb && (!pairs || !table)) || !atoms_ok(h, pairs, 2 * (size_t)nb))
        return fail(AWSEM_EINVAL, "add_fragment_memory: bad arguments");
    h->H.fm.push_back(HFm{group, ntab, vec(pairs, 2 * (size_t)nb), vec(table, (size_t)nb * ntab), k, dr});
    return AWSEM_OK;
}
int awsem_add_external(awsem_handle h, int group, int kind, int n, const int32_t* atoms, const double* param, double k, double center) {
    NEED_BUILD(h, group);
    if (kind < AWSEM_EXT_LINEAR_X || kind > AWSEM_EXT_ABS_Z) return fail(AWSEM_EINVAL, "add_external: unknown kind %d", kind);
    if (n < 0 || (n && (!atoms || !param)) || !atoms_ok(h, atoms, n)) return fail(AWSEM_EINVAL, "add_external: bad arguments");
    HMemb m{group, vec(atoms, n), vec(param, n), k, 0.0, 0.0, center};
    m.kind = kind;
    h->H.memb.push_back(std::move(m));
    return AWSEM_OK;
}
int awsem_add_membrane(awsem_handle h, int group, int n, const int32_t* atoms, const double* scale, double k, double k_m, double z_m, double center) {
    NEED_BUILD(h, group);
    if (n < 0 || (n && (!atoms || !scale)) || !atoms_ok(h, atoms, n)) return fail(AWSEM_EINVAL, "add_membrane: bad arguments");
    h->H.memb.push_back(HMemb{group, vec(atoms, n), vec(scale, n), k, k_m, z_m, center});
    return AWSEM_OK;
}
int awsem_add_debye_huckel(awsem_handle h, int group, int n, const int32_t* pairs, const double* qq, double k, double lam) {
    NEED_BUILD(h, group);
    if (n < 0 || !(lam > 0) || (n && (!pairs || !qq)) || !atoms_ok(h, pairs, 2 * (size_t)n)) return fail(AWSEM_EINVAL, "add_debye_huckel: bad arguments");
    h->H.dh.push_back(HDh{group, vec(pairs, 2 * (size_t)n), vec(qq, n), k, lam});
    return AWSEM_OK;
}

int awsem_add_gaussian_bonds(awsem_handle h, int group, int n, const int32_t* pairs, const double* r0, const double* sigma, const double* kgamma) {
    NEED_BUILD(h, group);
    if (n < 0 || (n && (!pairs || !r0 || !sigma || !kgamma)) || !atoms_ok(h, pairs, 2 * (size_t)n)) return fail(AWSEM_EINVAL, "add_gaussian_bonds: bad arguments");
    for (int b = 0; b < n; b++) if (!(sigma[b] > 0)) return fail(AWSEM_EINVAL, "add_gaussian_bonds: sigma must be positive");
    h->H.gauss.push_back(HGauss{group, vec(pairs, 2 * (size_t)n), vec(r0, n), vec(sigma, n), vec(kgamma, n)});
    return AWSEM_OK;
}
int awsem_add_bond_sum(awsem_handle h, int group, int bond_kind, int n, const int32_t* pairs, const double* r0, const double* sigma,
                       const double* coef, int cv_kind, double cv_k, double cv_x0) {
    NEED_BUILD(h, group);
    if (bond_kind < 0 || bond_kind > AWSEM_BOND_R2_XY || cv_kind < 0 || cv_kind > 3) return fail(AWSEM_EINVAL, "add_bond_sum: unknown bond or collective-variable kind");
    if (n < 0 || (n && (!pairs || !coef || (bond_kind == 0 && (!r0 || !sigma)))) || !atoms_ok(h, pairs, 2 * (size_t)n)) return fail(AWSEM_EINVAL, "add_bond_sum: bad arguments");
    for (int b = 0; b < n && bond_kind == 0; b++) if (!(sigma[b] > 0)) return fail(AWSEM_EINVAL, "add_bond_sum: sigma must be positive");
    HGauss g{group, vec(pairs, 2 * (size_t)n), bond_kind == 0 ? vec(r0, n) : std::vector<double>(), bond_kind == 0 ? vec(sigma, n) : std::vector<double>(), vec(coef, n)};
    if (bond_kind == 0) for (auto& c : g.kg) c = -c;   // stored as E = -c exp(...), the sign convention of awsem_add_gaussian_bonds
    g.kind = bond_kind; g.cv = cv_kind; g.cv_k = cv_k; g.cv_x0 = cv_x0;
    h->H.gauss.push_back(std::move(g));
    return AWSEM_OK;
}
int awsem_finalize(awsem_handle h) {
    if (!h) return fail(AWSEM_EINVAL, "null handle");
    if (h->finalized) return fail(AWSEM_ESTATE, "system already finalized");
    GUARD_BEGIN
    if (h->precision == AWSEM_FP64) h->eng.reset(new Engine<double>(h->natoms, h->nrep, h->device));
    else h->eng.reset(new Engine<float>(h->natoms, h->nrep, h->device));
    h->eng->finalize(h->H);
    h->finalized = true;
    return AWSEM_OK;
    GUARD_END
}

int awsem_set_positions(awsem_handle h, int rep, const double* xyz, int natoms) {
    NEED_RUN(h);
    if (!xyz || natoms != h->natoms || rep < -1 || rep >= h->nrep) return fail(AWSEM_EINVAL, "set_positions: bad arguments");
    GUARD_BEGIN h->eng->set_positions(rep, xyz); return AWSEM_OK; GUARD_END
}
int awsem_get_positions(awsem_handle h, int rep, double* xyz, int natoms) {
    NEED_RUN(h);
    if (!xyz || natoms != h->natoms || rep < 0 || rep >= h->nrep) return fail(AWSEM_EINVAL, "get_positions: bad arguments");
    GUARD_BEGIN h->eng->get_positions(rep, xyz); return AWSEM_OK; GUARD_END
}
int awsem_set_velocities(awsem_handle h, int rep, const double* v, int natoms) {
    NEED_RUN(h);
    if (!v || natoms != h->natoms || rep < -1 || rep >= h->nrep) return fail(AWSEM_EINVAL, "set_velocities: bad arguments");
    GUARD_BEGIN h->eng->set_velocities(rep, v); return AWSEM_OK; GUARD_END
}
int awsem_get_velocities(awsem_handle h, int rep, double* v, int natoms) {
    NEED_RUN(h);
    if (!v || natoms != h->natoms || rep < 0 || rep >= h->nrep) return fail(AWSEM_EINVAL, "get_velocities: bad arguments");
    GUARD_BEGIN h->eng->get_velocities(rep, v); return AWSEM_OK; GUARD_END
}
int awsem_compute_virtual_sites(awsem_handle h) {
    NEED_RUN(h);
    GUARD_BEGIN h->eng->compute_virtual_sites(); return AWSEM_OK; GUARD_END
}
int awsem_set_velocities_to_temperature(awsem_handle h, const double* T, uint64_t seed) {
    NEED_RUN(h);
    if (!T) return fail(AWSEM_EINVAL, "set_velocities_to_temperature: null temperatures");
    GUARD_BEGIN h->eng->maxwell(T, seed); return AWSEM_OK; GUARD_END
}
int awsem_scale_velocities(awsem_handle h, const double* scale) {
    NEED_RUN(h);
    if (!scale) return fail(AWSEM_EINVAL, "scale_velocities: null factors");
    GUARD_BEGIN h->eng->scale_velocities(scale); return AWSEM_OK; GUARD_END
}
int awsem_minimize(awsem_handle h, double tolerance, int max_iterations, int32_t* iterations, double* rms_force) {
    NEED_RUN(h);
    if (!(tolerance > 0) || max_iterations < 0) return fail(AWSEM_EINVAL, "minimize: tolerance must be positive, max_iterations >= 0");
    GUARD_BEGIN h->eng->minimize(tolerance, max_iterations, iterations, rms_force); return AWSEM_OK; GUARD_END
}
int awsem_kinetic_energy(awsem_handle h, double* ke) {
    NEED_RUN(h);
    if (!ke) return fail(AWSEM_EINVAL, "kinetic_energy: null output");
    GUARD_BEGIN h->eng->kinetic_energy(ke); return AWSEM_OK; GUARD_END
}
int awsem_set_state(awsem_handle h, const double* xyz, const double* vxyz) {
    NEED_RUN(h);
    GUARD_BEGIN h->eng->set_state_all(xyz, vxyz); return AWSEM_OK; GUARD_END
}
int awsem_get_state(awsem_handle h, double* xyz, double* vxyz) {
    NEED_RUN(h);
    GUARD_BEGIN h->eng->get_state_all(xyz, vxyz); return AWSEM_OK; GUARD_END
}
int awsem_snapshot_begin(awsem_handle h, int slot, uint32_t group_mask, int energy_replica, int replica, int with_state) {
    NEED_RUN(h);
    GUARD_BEGIN h->eng->snapshot_begin(slot, group_mask, energy_replica, replica, with_state); return AWSEM_OK; GUARD_END
}
int awsem_snapshot_end(awsem_handle h, int slot, double* e, double* ke, double* xyz_replica, double* xyz_all, double* vxyz_all) {
    NEED_RUN(h);
    GUARD_BEGIN h->eng->snapshot_end(slot, e, ke, xyz_replica, xyz_all, vxyz_all); return AWSEM_OK; GUARD_END
}
int awsem_energy_by_group(awsem_handle h, double* e) {
    NEED_RUN(h);
    if (!e) return fail(AWSEM_EINVAL, "energy_by_group: null output");
    GUARD_BEGIN h->eng->energy_by_group(0xffffffffu, e); return AWSEM_OK; GUARD_END
}
int awsem_energy(awsem_handle h, uint32_t mask, double* e) {
    NEED_RUN(h);
    if (!e) return fail(AWSEM_EINVAL, "energy: null output");
    GUARD_BEGIN
    std::vector<double> eg((size_t)h->nrep * AWSEM_NGROUPS);
    h->eng->energy_by_group(mask, eg.data());
    for (int r = 0; r < h->nrep; r++) {
        double s = 0;
        for (int g = 0; g < AWSEM_NGROUPS; g++) if ((mask >> g) & 1u) s += eg[(size_t)r * AWSEM_NGROUPS + g];
        e[r] = s;
    }
    return AWSEM_OK;
    GUARD_END
}
int awsem_forces(awsem_handle h, uint32_t mask, int redistribute, double* f) {
    NEED_RUN(h);
    if (!f) return fail(AWSEM_EINVAL, "forces: null output");
    GUARD_BEGIN h->eng->forces(mask, redistribute, f); return AWSEM_OK; GUARD_END
}
int awsem_step(awsem_handle h, int nsteps, const double* T, double dt, double friction, uint64_t seed) {
    NEED_RUN(h);
    if (nsteps < 0 || !T || !(dt > 0) || friction < 0) return fail(AWSEM_EINVAL, "step: bad arguments");
    GUARD_BEGIN
    if (nsteps > 0) h->eng->step(nsteps, T, dt, friction, seed, h->step);
    h->step += nsteps;
    return AWSEM_OK;
    GUARD_END
}
int awsem_step_stages(awsem_handle h, int nstages, int steps_per_stage, const double* T, double dt, double friction, uint64_t seed) {
    NEED_RUN(h);
    if (nstages < 0 || steps_per_stage < 0 || !T || !(dt > 0) || friction < 0) return fail(AWSEM_EINVAL, "step_stages: bad arguments");
    GUARD_BEGIN
    if (nstages > 0 && steps_per_stage > 0) h->eng->step_stages(nstages, steps_per_stage, T, dt, friction, seed, h->step);
    h->step += (int64_t)nstages * steps_per_stage;
    return AWSEM_OK;
    GUARD_END
}
int64_t awsem_step_count(awsem_handle h) { return h ? h->step : -1; }
int awsem_set_step_count(awsem_handle h, int64_t s) {
    if (!h) return fail(AWSEM_EINVAL, "null handle");
    h->step = s;
    return AWSEM_OK;
}
int awsem_neighbors(awsem_handle h, int replica, int kind, int index, double skin, int32_t* pairs, int capacity, int* n_pairs) {
    NEED_RUN(h);
    if (replica < 0 || replica >= h->nrep || capacity < 0 || (capacity && !pairs) || !n_pairs || !(skin >= 0))
        return fail(AWSEM_EINVAL, "neighbors: bad arguments");
    GUARD_BEGIN *n_pairs = h->eng->neighbors(replica, kind, index, skin, pairs, capacity); return AWSEM_OK; GUARD_END
}
int awsem_last_kernel_ms(awsem_handle h, double* ms) {
    NEED_RUN(h);
    if (!ms) return fail(AWSEM_EINVAL, "last_kernel_ms: null output");
    GUARD_BEGIN h->eng->synchronize(); *ms = h->eng->last_ms; return AWSEM_OK; GUARD_END
}
int awsem_synchronize(awsem_handle h) {
    NEED_RUN(h);
    GUARD_BEGIN h->eng->synchronize(); return AWSEM_OK; GUARD_END
}
int awsem_remd_setup(awsem_handle h, int n_total, const double* ladder_T, const int32_t* gather_index, int local_start, uint64_t seed) {
    NEED_RUN(h);
    if (n_total <= 0 || !ladder_T) return fail(AWSEM_EINVAL, "remd_setup: bad arguments");
    GUARD_BEGIN h->eng->remd_setup(n_total, ladder_T, gather_index, local_start, seed); return AWSEM_OK; GUARD_END
}
int awsem_step_remd(awsem_handle h, int nsteps, double dt, double friction, uint64_t seed) {
    NEED_RUN(h);
    if (nsteps < 0 || !(dt > 0) || friction < 0) return fail(AWSEM_EINVAL, "step_remd: bad arguments");
    GUARD_BEGIN
    if (nsteps > 0) h->eng->step_remd(nsteps, dt, friction, seed, h->step);
    h->step += nsteps;
    return AWSEM_OK;
    GUARD_END
}
int awsem_remd_energies(awsem_handle h, uint32_t mask, double* d_energy) {
    NEED_RUN(h);
    GUARD_BEGIN h->eng->remd_energies(mask, d_energy); return AWSEM_OK; GUARD_END
}
int awsem_remd_sweep(awsem_handle h, const double* d_all, int64_t iteration) {
    NEED_RUN(h);
    if (iteration < 0) return fail(AWSEM_EINVAL, "remd_sweep: negative iteration");
    GUARD_BEGIN h->eng->remd_sweep(d_all, iteration); return AWSEM_OK; GUARD_END
}
int awsem_remd_set_mixing(awsem_handle h, int scheme) {
    NEED_RUN(h);
    GUARD_BEGIN h->eng->remd_set_mixing(scheme); return AWSEM_OK; GUARD_END
}
int awsem_remd_state(awsem_handle h, int32_t* ladder_index, int64_t* attempted, int64_t* accepted, double* T_local) {
    NEED_RUN(h);
    GUARD_BEGIN h->eng->remd_state(ladder_index, attempted, accepted, T_local); return AWSEM_OK; GUARD_END
}
int awsem_energy_by_group_noalloc(awsem_handle h, int n_replicas, const double* d_xyz, double* d_energy, void* hip_stream) {
    NEED_RUN(h);
    if (n_replicas <= 0 || !d_xyz || !d_energy) return fail(AWSEM_EINVAL, "energy_by_group_noalloc: n_replicas must be positive and both device pointers non-null");
    GUARD_BEGIN h->eng->energy_noalloc(n_replicas, d_xyz, d_energy, (hipStream_t)hip_stream); return AWSEM_OK; GUARD_END
}

}  // extern "C"
