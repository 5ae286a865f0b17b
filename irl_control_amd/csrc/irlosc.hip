// libirlosc.so — C ABI (include/irlosc.h) over the gfx950 OSC kernels.  No CPU fallback: every
// compute entry point needs a HIP device and reports IRLOSC_ERR_HIP otherwise.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <algorithm>
#include <vector>

#include "../../include/irlosc.h"
#include "osc_common.hpp"
#include "osc_generic.hpp"
#ifndef IRLOSC_NO_GROUP_KERNEL
#include "osc_group.hpp"
#endif

using namespace irlosc;

static thread_local std::string g_create_error;

struct irlosc_ctx {
    irlosc_cfg cfg{};
    int k = 0;
    size_t esz = 4;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // resident inputs, one set per slot
    std::vector<void*> dM, dJ, ddq, dbias, dee, dwrench, dtgt, dtvel;
    std::vector<int> has_wrench, has_tvel, uploaded, targeted;
    // Outputs and the stage-2 hand-off buffers exist twice: irlosc_step_resident pipelines consecutive steps on
    // the group path (stage 2 of step i rides in the stage-1 launch of step i+1), alternating the two sets.
    void* du2[2] = {nullptr, nullptr};
    uint32_t* dflags2[2] = {nullptr, nullptr};
    float* dside2[2] = {nullptr, nullptr};
    int32_t* dwl2[2] = {nullptr, nullptr};
    int32_t* dwc2[2] = {nullptr, nullptr};
    int cur = 0;                       // output set written by the most recent step
    hipEvent_t tev_begin = nullptr, tev_end = nullptr;   // timing events handed to the next group launch (or null)
    std::vector<hipEvent_t> tev_pool;
    bool pending = false;              // a deferred stage 2 (of the step that wrote set `pending_set`) is outstanding
    bool defer_next = false;           // set by irlosc_step_resident around its launches
    int pending_nfast = 0;
    int pending_set = 0;
    KParams<float> pending_p{};
    void* du = nullptr;                // = du2[cur]
    uint32_t* dflags = nullptr;        // = dflags2[cur]
    void* dgains = nullptr;   // [nb][ndev][12] in dtype
    void* dnullkv = nullptr;  // [nb]
    int gains_nb = 0;
    unsigned long long* ddbg = nullptr;  // IRLOSC_PHASE_TIMING=1: 8 cycle stamps + 2 wall-clock stamps per stage-1 wave
    int kernel = IRLOSC_KERNEL_GENERIC;
    bool stage1_only = false; // set only inside irlosc_time_dominant_kernel
    std::string kernel_name;
    std::string err;
};

static int fail(irlosc_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((c), IRLOSC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

extern "C" int irlosc_abi_version(void) { return IRLOSC_ABI_VERSION; }

extern "C" int irlosc_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        g_create_error = std::string("hipGetDeviceCount failed: ") + hipGetErrorString(e);
        return IRLOSC_ERR_HIP;
    }
    return n;
}

extern "C" const char* irlosc_last_error(const irlosc_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

extern "C" const char* irlosc_kernel_name(const irlosc_ctx* ctx) {
    return ctx ? ctx->kernel_name.c_str() : "";
}

static int validate(const irlosc_cfg* c, int* k_out) {
    if (!c) return fail(nullptr, IRLOSC_ERR_ARG, "cfg is NULL");
    if (c->dtype != IRLOSC_F32 && c->dtype != IRLOSC_F64)
        return fail(nullptr, IRLOSC_ERR_ARG, "dtype must be IRLOSC_F32 or IRLOSC_F64");
    if (c->n < 1 || c->n > IRLOSC_MAX_N) return fail(nullptr, IRLOSC_ERR_ARG, "n=%d out of [1,%d]", c->n, IRLOSC_MAX_N);
    if (c->ndev < 1 || c->ndev > IRLOSC_MAX_DEV)
        return fail(nullptr, IRLOSC_ERR_ARG, "ndev=%d out of [1,%d]", c->ndev, IRLOSC_MAX_DEV);
    if (c->max_batch < 1) return fail(nullptr, IRLOSC_ERR_ARG, "max_batch must be >= 1");
    if (c->n_slots < 1) return fail(nullptr, IRLOSC_ERR_ARG, "n_slots must be >= 1");
    int k = 0;
    for (int d = 0; d < c->ndev; ++d) {
        int pc = 0;
        for (int i = 0; i < 6; ++i) pc += c->ctrlr_dof[d][i] ? 1 : 0;
        if (pc != c->dev_rows[d])
            return fail(nullptr, IRLOSC_ERR_ARG, "dev_rows[%d]=%d != popcount(ctrlr_dof)=%d", d, c->dev_rows[d], pc);
        if (c->n < 32 && (c->joint_mask[d] >> c->n))
            return fail(nullptr, IRLOSC_ERR_ARG, "joint_mask[%d] has bits >= n", d);
        if (c->j_idx0[d] < 0) return fail(nullptr, IRLOSC_ERR_ARG, "j_idx0[%d] negative", d);
        k += pc;
    }
    if (k < 1 || k > IRLOSC_MAX_K) return fail(nullptr, IRLOSC_ERR_ARG, "k=%d out of [1,%d]", k, IRLOSC_MAX_K);
    if (c->kernel < IRLOSC_KERNEL_AUTO || c->kernel > IRLOSC_KERNEL_GROUP)
        return fail(nullptr, IRLOSC_ERR_ARG, "unknown kernel id %d", c->kernel);
    *k_out = k;
    return IRLOSC_OK;
}

static void free_all(irlosc_ctx* c) {
    auto fr = [](std::vector<void*>& v) { for (void* p : v) if (p) (void)hipFree(p); v.clear(); };
    fr(c->dM); fr(c->dJ); fr(c->ddq); fr(c->dbias); fr(c->dee); fr(c->dwrench); fr(c->dtgt); fr(c->dtvel);
    for (int k = 0; k < 2; ++k) {
        if (c->du2[k]) (void)hipFree(c->du2[k]);
        if (c->dflags2[k]) (void)hipFree(c->dflags2[k]);
        if (c->dside2[k]) (void)hipFree(c->dside2[k]);
        if (c->dwl2[k]) (void)hipFree(c->dwl2[k]);
        if (c->dwc2[k]) (void)hipFree(c->dwc2[k]);
    }
    if (c->dgains) (void)hipFree(c->dgains);
    if (c->dnullkv) (void)hipFree(c->dnullkv);
    if (c->ddbg) (void)hipFree(c->ddbg);
    for (hipEvent_t ev : c->tev_pool) (void)hipEventDestroy(ev);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
}

static int create_impl(irlosc_ctx* c) {
    const irlosc_cfg& g = c->cfg;
    HIPCHK(nullptr, hipSetDevice(g.hip_device));
    HIPCHK(nullptr, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(nullptr, hipEventCreate(&c->ev0));
    HIPCHK(nullptr, hipEventCreate(&c->ev1));
    const size_t B = (size_t)g.max_batch, n = (size_t)g.n, k = (size_t)c->k, nd = (size_t)g.ndev, e = c->esz;
    auto alloc_slots = [&](std::vector<void*>& v, size_t bytes) -> hipError_t {
        v.assign(g.n_slots, nullptr);
        for (int s = 0; s < g.n_slots; ++s) {
            hipError_t r = hipMalloc(&v[s], bytes);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    };
    HIPCHK(nullptr, alloc_slots(c->dM, B * n * n * e));
    HIPCHK(nullptr, alloc_slots(c->dJ, B * k * n * e));
    HIPCHK(nullptr, alloc_slots(c->ddq, B * n * e));
    HIPCHK(nullptr, alloc_slots(c->dbias, B * n * e));
    HIPCHK(nullptr, alloc_slots(c->dee, B * nd * 7 * e));
    HIPCHK(nullptr, alloc_slots(c->dwrench, B * nd * 6 * e));
    HIPCHK(nullptr, alloc_slots(c->dtgt, B * nd * 7 * e));
    HIPCHK(nullptr, alloc_slots(c->dtvel, B * nd * 6 * e));
    c->has_wrench.assign(g.n_slots, 0);
    c->has_tvel.assign(g.n_slots, 0);
    c->uploaded.assign(g.n_slots, 0);
    c->targeted.assign(g.n_slots, 0);
    for (int k2 = 0; k2 < 2; ++k2) {
        HIPCHK(nullptr, hipMalloc(&c->du2[k2], B * n * e));
        HIPCHK(nullptr, hipMalloc((void**)&c->dflags2[k2], B * sizeof(uint32_t)));
        HIPCHK(nullptr, hipMalloc((void**)&c->dwl2[k2], B * sizeof(int32_t)));
        HIPCHK(nullptr, hipMalloc((void**)&c->dwc2[k2], 64 * sizeof(int32_t)));
        if (c->kernel == IRLOSC_KERNEL_GROUP)
            HIPCHK(nullptr, hipMalloc((void**)&c->dside2[k2], (B + 16 * 64) * 104 * sizeof(float)));
        HIPCHK(nullptr, hipMemsetAsync(c->dflags2[k2], 0, B * sizeof(uint32_t), c->stream));
        HIPCHK(nullptr, hipMemsetAsync(c->dwc2[k2], 0, 64 * sizeof(int32_t), c->stream));
    }
    c->du = c->du2[0];
    c->dflags = c->dflags2[0];
    HIPCHK(nullptr, hipMalloc(&c->dgains, B * nd * IRLOSC_GAIN_WORDS * e));
    HIPCHK(nullptr, hipMalloc(&c->dnullkv, B * e));
    if (c->kernel == IRLOSC_KERNEL_GROUP && getenv("IRLOSC_PHASE_TIMING"))
        HIPCHK(nullptr, hipMalloc((void**)&c->ddbg, (B / 16 + 1) * 10 * sizeof(unsigned long long)));
    HIPCHK(nullptr, hipStreamSynchronize(c->stream));
    return IRLOSC_OK;
}

static bool group_supported(const irlosc_ctx* c) {
#ifndef IRLOSC_NO_GROUP_KERNEL
    return group_kernel_supports(c->cfg.dtype, c->cfg.n, c->k, c->cfg.ndev);
#else
    (void)c;
    return false;
#endif
}

extern "C" int irlosc_create(const irlosc_cfg* cfg, irlosc_ctx** out) {
    if (!out) return fail(nullptr, IRLOSC_ERR_ARG, "out is NULL");
    *out = nullptr;
    int k = 0;
    int rc = validate(cfg, &k);
    if (rc) return rc;
    int ndevs = 0;
    if (hipGetDeviceCount(&ndevs) != hipSuccess || ndevs < 1)
        return fail(nullptr, IRLOSC_ERR_HIP, "no HIP device available (libirlosc has no CPU fallback)");
    if (cfg->hip_device < 0 || cfg->hip_device >= ndevs)
        return fail(nullptr, IRLOSC_ERR_ARG, "hip_device=%d but %d device(s) visible", cfg->hip_device, ndevs);
    irlosc_ctx* c = new (std::nothrow) irlosc_ctx();
    if (!c) return fail(nullptr, IRLOSC_ERR_HIP, "out of host memory");
    c->cfg = *cfg;
    c->k = k;
    c->esz = cfg->dtype == IRLOSC_F64 ? 8 : 4;
    if (cfg->kernel == IRLOSC_KERNEL_GROUP && !group_supported(c)) {
        delete c;
        return fail(nullptr, IRLOSC_ERR_ARG, "group kernel not available for n=%d k=%d ndev=%d", cfg->n, k, cfg->ndev);
    }
    c->kernel = (cfg->kernel == IRLOSC_KERNEL_GENERIC || !group_supported(c)) ? IRLOSC_KERNEL_GENERIC
                                                                              : IRLOSC_KERNEL_GROUP;
    char nm[96];
    snprintf(nm, sizeof nm, "%s_%s_n%d_k%d", c->kernel == IRLOSC_KERNEL_GROUP ? "osc_group" : "osc_generic",
             cfg->dtype == IRLOSC_F64 ? "f64" : "f32", cfg->n, k);
    c->kernel_name = nm;
    rc = create_impl(c);
    if (rc) {
        free_all(c);
        delete c;
        return rc;
    }
    *out = c;
    return IRLOSC_OK;
}

extern "C" void irlosc_destroy(irlosc_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.hip_device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_all(c);
    delete c;
}

template <typename T>
static void convert(const double* src, std::vector<unsigned char>& dst, size_t count) {
    dst.resize(count * sizeof(T));
    T* d = reinterpret_cast<T*>(dst.data());
    for (size_t i = 0; i < count; ++i) d[i] = (T)src[i];
}

extern "C" int irlosc_set_gains(irlosc_ctx* c, const double* gains, const double* null_kv, int32_t nb) {
    if (!c) return IRLOSC_ERR_ARG;
    if (!gains) return fail(c, IRLOSC_ERR_ARG, "gains is NULL");
    if (nb != 1 && nb != c->cfg.max_batch)
        return fail(c, IRLOSC_ERR_ARG, "nb must be 1 (broadcast) or max_batch=%d, got %d", c->cfg.max_batch, nb);
    if ((c->cfg.flags & IRLOSC_NULLSPACE) && !null_kv)
        return fail(c, IRLOSC_ERR_ARG, "null_kv required with IRLOSC_NULLSPACE");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    const size_t cnt = (size_t)nb * c->cfg.ndev * IRLOSC_GAIN_WORDS;
    std::vector<double> zero(nb, 0.0);
    const double* nk = null_kv ? null_kv : zero.data();
    std::vector<unsigned char> a, b;
    if (c->cfg.dtype == IRLOSC_F64) { convert<double>(gains, a, cnt); convert<double>(nk, b, nb); }
    else { convert<float>(gains, a, cnt); convert<float>(nk, b, nb); }
    HIPCHK(c, hipMemcpyAsync(c->dgains, a.data(), a.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dnullkv, b.data(), b.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->gains_nb = nb;
    return IRLOSC_OK;
}

static int check_slot(irlosc_ctx* c, int slot, int B) {
    if (slot < 0 || slot >= c->cfg.n_slots) return fail(c, IRLOSC_ERR_ARG, "slot %d out of [0,%d)", slot, c->cfg.n_slots);
    if (B < 0 || B > c->cfg.max_batch) return fail(c, IRLOSC_ERR_ARG, "B=%d out of [0,%d]", B, c->cfg.max_batch);
    return IRLOSC_OK;
}

extern "C" int irlosc_upload(irlosc_ctx* c, int32_t slot, int32_t B, const void* M, const void* J, const void* dq,
                             const void* bias, const void* ee_pose, const void* wrench) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (B == 0) { c->uploaded[slot] = 1; return IRLOSC_OK; }
    if (!M || !J || !dq || !ee_pose) return fail(c, IRLOSC_ERR_ARG, "M, J, dq and ee_pose are required");
    if ((c->cfg.flags & IRLOSC_USE_G) && !bias) return fail(c, IRLOSC_ERR_ARG, "bias required with IRLOSC_USE_G");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    const size_t b = (size_t)B, n = (size_t)c->cfg.n, k = (size_t)c->k, nd = (size_t)c->cfg.ndev, e = c->esz;
    HIPCHK(c, hipMemcpyAsync(c->dM[slot], M, b * n * n * e, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dJ[slot], J, b * k * n * e, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->ddq[slot], dq, b * n * e, hipMemcpyHostToDevice, c->stream));
    if (bias) HIPCHK(c, hipMemcpyAsync(c->dbias[slot], bias, b * n * e, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dee[slot], ee_pose, b * nd * 7 * e, hipMemcpyHostToDevice, c->stream));
    if (wrench) HIPCHK(c, hipMemcpyAsync(c->dwrench[slot], wrench, b * nd * 6 * e, hipMemcpyHostToDevice, c->stream));
    c->has_wrench[slot] = wrench != nullptr;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->uploaded[slot] = 1;
    return IRLOSC_OK;
}

extern "C" int irlosc_set_targets(irlosc_ctx* c, int32_t slot, int32_t B, const void* tgt_pose, const void* tgt_vel) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (B == 0) { c->targeted[slot] = 1; return IRLOSC_OK; }
    if (!tgt_pose) return fail(c, IRLOSC_ERR_ARG, "tgt_pose is NULL");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    const size_t b = (size_t)B, nd = (size_t)c->cfg.ndev, e = c->esz;
    HIPCHK(c, hipMemcpyAsync(c->dtgt[slot], tgt_pose, b * nd * 7 * e, hipMemcpyHostToDevice, c->stream));
    if (tgt_vel) HIPCHK(c, hipMemcpyAsync(c->dtvel[slot], tgt_vel, b * nd * 6 * e, hipMemcpyHostToDevice, c->stream));
    c->has_tvel[slot] = tgt_vel != nullptr;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->targeted[slot] = 1;
    return IRLOSC_OK;
}

template <typename T>
static void fill_params(const irlosc_ctx* c, KParams<T>& p, int B, const void* M, const void* J, const void* dq,
                        const void* bias, const void* ee, const void* tgt, const void* tvel, const void* wrench,
                        void* u, uint32_t* flags) {
    memset(&p, 0, sizeof p);
    p.M = (const T*)M; p.J = (const T*)J; p.dq = (const T*)dq; p.bias = (const T*)bias;
    p.ee = (const T*)ee; p.tgt = (const T*)tgt; p.tvel = (const T*)tvel; p.wrench = (const T*)wrench;
    p.u = (T*)u; p.flags = flags;
    p.gains = (const T*)c->dgains; p.null_kv = (const T*)c->dnullkv;
    p.index = nullptr;
    p.dbg = c->ddbg;
    p.gains_per_instance = c->gains_nb > 1;
    p.B = B; p.n = c->cfg.n; p.k = c->k; p.ndev = c->cfg.ndev; p.cfgflags = c->cfg.flags;
    int row = 0;
    for (int d = 0; d < c->cfg.ndev; ++d) {
        DevMeta& m = p.dev[d];
        m.row0 = row; m.rows = c->cfg.dev_rows[d]; row += m.rows;
        m.dofmask = 0;
        for (int i = 0; i < 6; ++i) if (c->cfg.ctrlr_dof[d][i]) m.dofmask |= 1u << i;
        m.calc = (c->cfg.calc_xyz[d] ? 1u : 0u) | (c->cfg.calc_abg[d] ? 2u : 0u);
        m.joint_mask = c->cfg.joint_mask[d];
        m.jidx0 = c->cfg.j_idx0[d];
    }
}

#ifndef IRLOSC_NO_GROUP_KERNEL
static GroupScratch scratch_for(const irlosc_ctx* c, int set) {
    GroupScratch gs{};
    gs.worklist2 = c->dwl2[set];
    gs.counts = c->dwc2[set];
    gs.side = c->dside2[set];
    gs.side_cap = c->cfg.max_batch + 16 * 64;
    return gs;
}

// Run the stage 2 that a pipelined irlosc_step_resident left outstanding (no-op otherwise).
static int flush_pending(irlosc_ctx* c, hipStream_t st) {
    if (!c->pending) return IRLOSC_OK;
    c->pending = false;
    const GroupScratch gs = scratch_for(c, c->pending_set);
    int rc = launch_group_stage2<float>(c->pending_p, make_s2(c->pending_p, c->pending_nfast, gs), st);
    if (rc) return fail(c, IRLOSC_ERR_HIP, "stage-2 launch failed: %s", hipGetErrorString((hipError_t)rc));
    return IRLOSC_OK;
}
#else
static int flush_pending(irlosc_ctx*, hipStream_t) { return IRLOSC_OK; }
#endif

template <typename T>
static int launch_t(irlosc_ctx* c, int B, const void* M, const void* J, const void* dq, const void* bias,
                    const void* ee, const void* tgt, const void* tvel, const void* wrench, void* u,
                    uint32_t* flags, hipStream_t st) {
    KParams<T> p;
    fill_params<T>(c, p, B, M, J, dq, bias, ee, tgt, tvel, wrench, u, flags);
#ifndef IRLOSC_NO_GROUP_KERNEL
    if (c->kernel == IRLOSC_KERNEL_GROUP) {
        if constexpr (sizeof(T) == 4) {
            // which output set does `u` belong to?  (caller-owned buffers of irlosc_step_device use set `cur`)
            const int set = (u == c->du2[1]) ? 1 : (u == c->du2[0] ? 0 : c->cur);
            GroupScratch gs = scratch_for(c, set);
            gs.stage1_only = c->stage1_only;
            gs.ev_begin = c->tev_begin;
            gs.ev_end = c->tev_end;
            gs.defer_stage2 = c->defer_next && !c->stage1_only;
            if (c->pending && c->pending_set != set && !c->stage1_only) {      // previous step's stage 2 rides along
                const GroupScratch gp = scratch_for(c, c->pending_set);
                gs.have_prev = true;
                gs.prev = make_s2(c->pending_p, c->pending_nfast, gp);
                gs.prev_p = c->pending_p;
                c->pending = false;
            } else if (c->pending && !c->stage1_only) {
                int rc0 = flush_pending(c, st);                              // same set: must finish first
                if (rc0) return rc0;
            }
            int rc = launch_group<float>(p, gs, st);
            if (rc) return fail(c, IRLOSC_ERR_HIP, "group kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
            if (gs.defer_stage2) {
                const int tile1 = 16;
                c->pending = true;
                c->pending_set = set;
                c->pending_nfast = (B / tile1) * tile1;
                c->pending_p = p;
            }
            return IRLOSC_OK;
        } else {
            return fail(c, IRLOSC_ERR_ARG, "no fp64 group kernel");
        }
    }
#endif
    size_t smem = generic_smem_bytes<T>(p.n, p.k, p.ndev);
    hipLaunchKernelGGL(osc_generic_kernel<T>, dim3(B), dim3(64), smem, st, p);
    HIPCHK(c, hipGetLastError());
    return IRLOSC_OK;
}

static int launch(irlosc_ctx* c, int B, const void* M, const void* J, const void* dq, const void* bias,
                  const void* ee, const void* tgt, const void* tvel, const void* wrench, void* u,
                  uint32_t* flags, hipStream_t st) {
    if (B == 0) return IRLOSC_OK;
    if (c->gains_nb == 0) return fail(c, IRLOSC_ERR_STATE, "irlosc_set_gains has not been called");
    if (c->cfg.dtype == IRLOSC_F64) return launch_t<double>(c, B, M, J, dq, bias, ee, tgt, tvel, wrench, u, flags, st);
    return launch_t<float>(c, B, M, J, dq, bias, ee, tgt, tvel, wrench, u, flags, st);
}

static int launch_slot(irlosc_ctx* c, int slot, int B) {
    if (!c->uploaded[slot] || !c->targeted[slot])
        return fail(c, IRLOSC_ERR_STATE, "slot %d: irlosc_upload and irlosc_set_targets must precede a step", slot);
    return launch(c, B, c->dM[slot], c->dJ[slot], c->ddq[slot], c->dbias[slot], c->dee[slot], c->dtgt[slot],
                  c->has_tvel[slot] ? c->dtvel[slot] : nullptr, c->has_wrench[slot] ? c->dwrench[slot] : nullptr,
                  c->du, c->dflags, c->stream);
}

extern "C" int irlosc_download(irlosc_ctx* c, int32_t B, void* u_host, uint32_t* flags_host) {
    if (!c) return IRLOSC_ERR_ARG;
    if (B < 0 || B > c->cfg.max_batch) return fail(c, IRLOSC_ERR_ARG, "B out of range");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    if (u_host && B) HIPCHK(c, hipMemcpyAsync(u_host, c->du, (size_t)B * c->cfg.n * c->esz, hipMemcpyDeviceToHost, c->stream));
    if (flags_host && B) HIPCHK(c, hipMemcpyAsync(flags_host, c->dflags, (size_t)B * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->ddbg && B >= 16) {   // debug: mean cycles per stage-1 phase over all waves
        const int tiles = B / 16;
        std::vector<unsigned long long> h((size_t)tiles * 10);
        HIPCHK(c, hipMemcpy(h.data(), c->ddbg, h.size() * 8, hipMemcpyDeviceToHost));
        static const char* nm[7] = {"vec-wait", "M-stream+Cholesky", "J+fwd-subst", "task-error", "A=YtY", "kxk", "torques+store"};
        double acc[7] = {0}, rt = 0;
        unsigned long long rmin = ~0ull, rmax = 0;
        for (int t = 0; t < tiles; ++t) {
            for (int i = 0; i < 7; ++i) acc[i] += (double)(h[(size_t)t * 10 + i + 1] - h[(size_t)t * 10 + i]);
            const unsigned long long r0 = h[(size_t)t * 10 + 8] & 0x0fffffffffffffffull;
            rt += (double)(h[(size_t)t * 10 + 9] - r0);
            rmin = std::min(rmin, r0);
            rmax = std::max(rmax, h[(size_t)t * 10 + 9]);
        }
        {   // where do blocks land?  XCC of tile t vs t % 8, and vs the XCC of tile t + 2048; start order of the second round
            int same_mod = 0, same_next = 0, cnt_next = 0;
            for (int t = 0; t < tiles; ++t) {
                const int x = (int)(h[(size_t)t * 10 + 8] >> 60);
                same_mod += (x == (t % 8));
                if (t + 2048 < tiles) { ++cnt_next; same_next += (x == (int)(h[(size_t)(t + 2048) * 10 + 8] >> 60)); }
            }
            fprintf(stderr, "[irlosc placement] xcc==tile%%8: %d/%d, xcc(t)==xcc(t+2048): %d/%d; xcc of tiles 0..15:", same_mod, tiles, same_next, cnt_next);
            for (int t = 0; t < 16 && t < tiles; ++t) fprintf(stderr, " %d", (int)(h[(size_t)t * 10 + 8] >> 60));
            fprintf(stderr, "\n");
        }
        double tot = 0; for (int i = 0; i < 7; ++i) tot += acc[i];
        fprintf(stderr, "[irlosc phase timing] %d waves, mean cycles/wave %.0f:", tiles, tot / tiles);
        for (int i = 0; i < 7; ++i) fprintf(stderr, " %s=%.0f", nm[i], acc[i] / tiles);
        // s_memrealtime ticks at 100 MHz: wave residency in us, the shader clock it implies, first start -> last end
        fprintf(stderr, " | wave %.2f us => %.0f MHz, launch span %.2f us\n", rt / tiles / 100.0,
                (tot / tiles) / (rt / tiles / 100.0), (double)(rmax - rmin) / 100.0);
    }
    return IRLOSC_OK;
}

extern "C" int irlosc_step(irlosc_ctx* c, int32_t slot, int32_t B, void* u_host, uint32_t* flags_host) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    rc = launch_slot(c, slot, B);
    if (rc) return rc;
    if (u_host || flags_host) return irlosc_download(c, B, u_host, flags_host);
    return IRLOSC_OK;
}

extern "C" int irlosc_step_resident(irlosc_ctx* c, int32_t first_slot, int32_t B, int32_t iters, float* ms_total,
                                    float* ms_kernel_avg) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, first_slot, B);
    if (rc) return rc;
    if (iters < 1) return fail(c, IRLOSC_ERR_ARG, "iters must be >= 1");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    // Group path: consecutive steps are pipelined - step i writes output set i % 2 and its stage 2 rides in the
    // stage-1 launch of step i + 1 (the last one is flushed below), all on one stream.
    const bool pipe = c->kernel == IRLOSC_KERNEL_GROUP && !getenv("IRLOSC_NO_PIPELINE");
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < iters; ++i) {
        if (pipe) {
            c->cur ^= 1;
            c->du = c->du2[c->cur];
            c->dflags = c->dflags2[c->cur];
        }
        c->defer_next = pipe;
        rc = launch_slot(c, (first_slot + i) % c->cfg.n_slots, B);
        c->defer_next = false;
        if (rc) return rc;
    }
    rc = flush_pending(c, c->stream);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (ms_total) *ms_total = ms;
    if (ms_kernel_avg) *ms_kernel_avg = ms / (float)iters;
    return IRLOSC_OK;
}

extern "C" int irlosc_time_dominant_kernel(irlosc_ctx* c, int32_t slot, int32_t B, int32_t iters, float* ms_avg) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (iters < 1 || iters > 256 || !ms_avg) return fail(c, IRLOSC_ERR_ARG, "iters must be in [1,256] and ms_avg non-NULL");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    if (c->kernel != IRLOSC_KERNEL_GROUP) {          // generic path: a step IS the dominant kernel
        float tot = 0.f;
        rc = irlosc_step_resident(c, slot, B, iters, &tot, ms_avg);
        return rc;
    }
    // Group path: the same pipelined launches as irlosc_step_resident, with a HIP event pair around each
    // dominant launch (stage 1 of step i fused with the riding stage 2 of step i-1) - this is the kernel a
    // rocprofv3 kernel trace of the timed region shows, so the two averages are comparable.
    while ((int)c->tev_pool.size() < 2 * iters) {
        hipEvent_t ev;
        HIPCHK(c, hipEventCreate(&ev));
        c->tev_pool.push_back(ev);
    }
    const bool pipe = !getenv("IRLOSC_NO_PIPELINE");
    for (int i = -1; i < iters; ++i) {               // i = -1: untimed first launch (nothing rides in it yet)
        if (pipe) {
            c->cur ^= 1;
            c->du = c->du2[c->cur];
            c->dflags = c->dflags2[c->cur];
        }
        c->defer_next = pipe;
        c->tev_begin = i >= 0 ? c->tev_pool[2 * i] : nullptr;
        c->tev_end = i >= 0 ? c->tev_pool[2 * i + 1] : nullptr;
        rc = launch_slot(c, (slot + i + 1) % c->cfg.n_slots, B);
        c->defer_next = false;
        c->tev_begin = c->tev_end = nullptr;
        if (rc) return rc;
    }
    rc = flush_pending(c, c->stream);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (int i = 0; i < iters; ++i) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->tev_pool[2 * i], c->tev_pool[2 * i + 1]));
        tot += ms;
    }
    *ms_avg = (float)(tot / iters);
    return IRLOSC_OK;
}

extern "C" int irlosc_sync(irlosc_ctx* c) {
    if (!c) return IRLOSC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return IRLOSC_OK;
}

extern "C" int irlosc_step_device(irlosc_ctx* c, int32_t B, const void* dM, const void* dJ, const void* ddq,
                                  const void* dbias, const void* dee_pose, const void* dtgt_pose,
                                  const void* dtgt_vel, const void* dwrench, void* du, uint32_t* dflags,
                                  void* hip_stream) {
    if (!c) return IRLOSC_ERR_ARG;
    if (B < 0 || B > c->cfg.max_batch) return fail(c, IRLOSC_ERR_ARG, "B=%d out of [0,%d]", B, c->cfg.max_batch);
    if (!dM || !dJ || !ddq || !dee_pose || !dtgt_pose || !du || !dflags)
        return fail(c, IRLOSC_ERR_ARG, "dM, dJ, ddq, dee_pose, dtgt_pose, du and dflags are required");
    if ((c->cfg.flags & IRLOSC_USE_G) && !dbias) return fail(c, IRLOSC_ERR_ARG, "dbias required with IRLOSC_USE_G");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    return launch(c, B, dM, dJ, ddq, dbias, dee_pose, dtgt_pose, dtgt_vel, dwrench, du, dflags, st);
}
