// libirlosc.so — C ABI (include/irlosc.h) over the gfx950 OSC kernels.  No CPU fallback: every
// compute entry point needs a HIP device and reports IRLOSC_ERR_HIP otherwise.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: the library is dlopen()ed on first use
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/irlosc.h"
#include "osc_common.hpp"
#include "osc_generic.hpp"
#include "osc_assemble.hpp"
#include "osc_row16.hpp"
#include "osc_frontend.hpp"
#include "osc_lane_types.hpp"
#include "launchers.hpp"

using namespace irlosc;
static_assert(FE_TRAIN == R16_TRAIN, "the walk and the OSC kernel chain the same number of steps per launch");

static thread_local std::string g_create_error;

struct irlosc_ctx {
    irlosc_cfg cfg{};
    int k = 0;
    size_t esz = 4;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // resident inputs, one set per slot
    std::vector<void*> dM, dJ, ddq, dbias, dee, dwrench, dtgt, dtvel;
    std::vector<int> has_wrench, has_tvel;
    std::vector<int> uploaded, targeted;    // instances of the slot that hold state / targets (0 = nothing yet, -1 = an empty batch)
    std::vector<int> fused_away;            // 1: the slot's dense records were invalidated by a fused step from joint coordinates (error text only)
    // Output sets (u, flags): step i of a row16 train writes set i; the generic path only ever uses set 0.
    static constexpr int NSETS_MAX = R16_TRAIN;
    int nsets = 1;
    int train = 1;                     // steps per launch in irlosc_step_resident
    void* du_set[NSETS_MAX] = {};
    uint32_t* dflags_set[NSETS_MAX] = {};
    int cur = 0;                       // output set written by the most recent step
    hipEvent_t tev_begin = nullptr, tev_end = nullptr;   // timing events handed to the next train launch (or null)
    unsigned long long* dspan = nullptr;   // irlosc_time_trains: [ntrains][2] wall-clock stamps written by the kernels
    int dspan_cap = 0;
    unsigned long long* span_next = nullptr;   // the pair the next train launch stamps (or null)
    std::vector<hipEvent_t> tev_pool;
    void* du = nullptr;                // = du_set[cur]
    uint32_t* dflags = nullptr;        // = dflags_set[cur]
    void* draw = nullptr;     // staging for irlosc_upload_raw (raw simulator arrays), grown on demand
    size_t draw_bytes = 0;
    // fp64 row16 path: zero page for the padding lanes, worklist of the instances handed to the generic kernel, and
    // two counters used alternately (the worklist pass of a step zeroes the counter of the next one)
    void* dzeros = nullptr;
    int32_t* dr16_list[R16_TRAIN] = {};    // give-up list of each step of a train
    int32_t* dr16_count = nullptr;         // [R16_TRAIN] give-up counters, zeroed in front of every train
    // rigid-body front end (irlosc_set_model): device copy of the tables, resident joint coordinates per slot
    FeModel* dmodel = nullptr;
    size_t fe_smem = 0;
    int fe_lane = 0;                  // 1: the model has the compiled Dual-UR5 shape -> lane-per-instance front end
    int fe_lane_s = 0;                // 1: ... and the structural constants of its MJCF -> the fused walk with them compiled in (TopoDualUr5S)
    double* fe_side = nullptr;        // side buffer of the lane kernel: [wave][entry][64]
    // fused path (irlosc_step_from_q / irlosc_step_resident_from_q on the row16 kernel): entry tables of the compact exchange
    // buffer and one buffer per step of a train
    FeCompactTables* dtables = nullptr;
    size_t fe_xentries = 0;
    double* fe_xside[R16_TRAIN] = {};
    double* dtrows[R16_TRAIN] = {};        // row16 path on dense records: task rows of each step of a train (osc_task_rows_dense_kernel);
                                           // allocated by the first train that needs them (ensure_trows)
    int task_pass = 1;                     // IRLOSC_TASK_PASS=0: part 1 of the task signal in the row16 kernel (A/B, tests)
    int fused = 0;
    int fused_train = R16_TRAIN;
    // the OSC step of the fused path in lane-per-robot form (osc_lane.hpp): the instantiation that holds the layout (-1: none: the row16
    // FROMQ kernel stays), its row map, the records + counters of the eigen pass behind it (allocated by the first fused step)
    // Consecutive trains of irlosc_step_resident_from_q rotate over BANKS of buffers (exchange buffers, eigen-pass records, counters,
    // give-up lists, output sets), each on a stream of its own: the walk and the lane kernel run one wave per SIMD, eight waves deep per
    // train, so every kernel boundary leaves SIMDs idle for up to a wave's lifetime (~50 us) -- measured as a fixed ~126 us per train of
    // 990 us (trains of 8 / 4 / 2 steps: 124 / 140 / 163 us per step).  With the NEXT trains independent and on other streams their first
    // waves fill those tails (two banks: 60 us of the 126 left; three: ~45).  Bank 0 = the buffers above on `stream`; the others are
    // allocated by the first call that chains that many trains.
    struct Bank {
        hipStream_t st = nullptr;
        hipEvent_t done = nullptr;
        double* xside[R16_TRAIN] = {};
        double* lane_rec[R16_TRAIN] = {};
        int32_t* lane_count = nullptr;
        int32_t* list[R16_TRAIN] = {};
        int32_t* count = nullptr;
        void* u[R16_TRAIN] = {};
        uint32_t* flags[R16_TRAIN] = {};
        double* trows[R16_TRAIN] = {};     // (dense-record trains: rows of the task pass)
    };
    static constexpr int MAX_XBANKS = 3;   // banks beside the context's own: trains of the fused path rotate over 1 + fq_xbanks of them
    Bank xb[MAX_XBANKS];
    int fq_xbanks = 2;                     // IRLOSC_FQ_BANKS = 2 .. 4 banks in all (default 3: k13 6.88 -> 7.00e8, four: 7.03e8, +2.4 GB each); the dense-record trains use xb[0] only
    hipEvent_t ev_join = nullptr;
    int fq_overlap = 1;                    // IRLOSC_FQ_OVERLAP=0: one bank, one stream (A/B measurements, tests)
    int r16_overlap = 1;                   // IRLOSC_R16_OVERLAP=0: the same switch for the trains of irlosc_step_resident on dense records
    int r16_xbanks = 1;                    // IRLOSC_R16_BANKS = 2 .. 4 banks in all for those trains (default 2)
    int32_t* count_cur = nullptr;          // give-up counters of the most recent train (irlosc_giveup_counts)
    int lane_tier = -1;
    lane::RowMap lane_map{};
    double* lane_rec[R16_TRAIN] = {};
    int32_t* dlane_count = nullptr;        // [R16_TRAIN]
    std::vector<double*> dqpos, dqvel;
    std::vector<double*> dqt;          // per slot: the same coordinates in the fused walk's layout [wave][2 n][64 robots] (irlosc_upload_q writes both)
    std::vector<int> has_q;
    // irlosc_tick: one pinned host block and one device block per direction, grown on demand
    void* tick_hin = nullptr; void* tick_din = nullptr; size_t tick_in_bytes = 0;
    void* tick_hout = nullptr; void* tick_dout = nullptr; size_t tick_out_bytes = 0;
    int32_t* dsym = nullptr;  // symmetry probe of the throughput paths: {count, first instance}
    // Tree-structured factorisation on dense records (row16 kernel): per slot, 1 when the records in it were verified to carry
    // the zero pattern of the compiled Dual-UR5 tree (probe at upload) or were written by the lane front end (by construction)
    std::vector<int> tree_ok;
    int tree_enabled = 1;              // IRLOSC_TREE=0 turns the form off (A/B measurements)
    StructureMasks tree_masks;
    int32_t* dstruct = nullptr;        // result word of the structure probe
    void* dgains = nullptr;   // [nb][ndev][12] in dtype
    void* dnullkv = nullptr;  // [nb]
    int gains_nb = 0;
    unsigned long long* ddbg = nullptr;  // IRLOSC_PHASE_TIMING=1: 8 cycle stamps + 2 wall-clock stamps per stage-1 wave
    int kernel = IRLOSC_KERNEL_GENERIC;
    int kernel_class = IRLOSC_CLASS_GENERIC;
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

extern "C" const char* irlosc_frontend_name(const irlosc_ctx* ctx) {
    if (!ctx || !ctx->dmodel) return "";
    const bool f64 = ctx->cfg.dtype == IRLOSC_F64;
    if (ctx->fe_lane) return f64 ? "osc_frontend_lane_dual_ur5_f64out" : "osc_frontend_lane_dual_ur5_f32out";
    return f64 ? "osc_frontend_generic_f64out" : "osc_frontend_generic_f32out";
}

extern "C" const char* irlosc_kernel_name(const irlosc_ctx* ctx) {
    return ctx ? ctx->kernel_name.c_str() : "";
}
extern "C" int irlosc_kernel_class(const irlosc_ctx* ctx) { return ctx ? ctx->kernel_class : IRLOSC_ERR_ARG; }

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
    if (c->kernel < IRLOSC_KERNEL_AUTO || c->kernel > IRLOSC_KERNEL_ROW16)
        return fail(nullptr, IRLOSC_ERR_ARG, "unknown kernel id %d", c->kernel);
    *k_out = k;
    return IRLOSC_OK;
}

static void free_all(irlosc_ctx* c) {
    auto fr = [](std::vector<void*>& v) { for (void* p : v) if (p) (void)hipFree(p); v.clear(); };
    fr(c->dM); fr(c->dJ); fr(c->ddq); fr(c->dbias); fr(c->dee); fr(c->dwrench); fr(c->dtgt); fr(c->dtvel);
    for (int k = 0; k < irlosc_ctx::NSETS_MAX; ++k) {
        if (c->du_set[k]) (void)hipFree(c->du_set[k]);
        if (c->dflags_set[k]) (void)hipFree(c->dflags_set[k]);
    }
    if (c->draw) (void)hipFree(c->draw);
    if (c->dmodel) (void)hipFree(c->dmodel);
    if (c->fe_side) (void)hipFree(c->fe_side);
    if (c->dtables) (void)hipFree(c->dtables);
    for (int k = 0; k < R16_TRAIN; ++k) if (c->fe_xside[k]) (void)hipFree(c->fe_xside[k]);
    for (int k = 0; k < R16_TRAIN; ++k) if (c->dtrows[k]) (void)hipFree(c->dtrows[k]);
    for (int k = 0; k < R16_TRAIN; ++k) if (c->lane_rec[k]) (void)hipFree(c->lane_rec[k]);
    for (irlosc_ctx::Bank& bk : c->xb) {
        for (int k = 0; k < R16_TRAIN; ++k) {
            if (bk.xside[k]) (void)hipFree(bk.xside[k]);
            if (bk.lane_rec[k]) (void)hipFree(bk.lane_rec[k]);
            if (bk.list[k]) (void)hipFree(bk.list[k]);
            if (bk.u[k]) (void)hipFree(bk.u[k]);
            if (bk.flags[k]) (void)hipFree(bk.flags[k]);
            if (bk.trows[k]) (void)hipFree(bk.trows[k]);
        }
        if (bk.lane_count) (void)hipFree(bk.lane_count);
        if (bk.count) (void)hipFree(bk.count);
        if (bk.done) (void)hipEventDestroy(bk.done);
        if (bk.st) (void)hipStreamDestroy(bk.st);
    }
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->dlane_count) (void)hipFree(c->dlane_count);
    for (double* p : c->dqpos) if (p) (void)hipFree(p);
    for (double* p : c->dqvel) if (p) (void)hipFree(p);
    for (double* p : c->dqt) if (p) (void)hipFree(p);
    if (c->tick_hin) (void)hipHostFree(c->tick_hin);
    if (c->tick_din) (void)hipFree(c->tick_din);
    if (c->tick_hout) (void)hipHostFree(c->tick_hout);
    if (c->tick_dout) (void)hipFree(c->tick_dout);
    if (c->dzeros) (void)hipFree(c->dzeros);
    for (int k = 0; k < R16_TRAIN; ++k) if (c->dr16_list[k]) (void)hipFree(c->dr16_list[k]);
    if (c->dr16_count) (void)hipFree(c->dr16_count);
    if (c->dsym) (void)hipFree(c->dsym);
    if (c->dspan) (void)hipFree(c->dspan);
    if (c->dstruct) (void)hipFree(c->dstruct);
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
    if (c->kernel == IRLOSC_KERNEL_ROW16) {
        c->train = R16_TRAIN;
        c->nsets = R16_TRAIN;               // a train completes (give-up pass included) before the next one starts
    }
    for (int k2 = 0; k2 < c->nsets; ++k2) {
        HIPCHK(nullptr, hipMalloc(&c->du_set[k2], B * n * e));
        HIPCHK(nullptr, hipMalloc((void**)&c->dflags_set[k2], B * sizeof(uint32_t)));
        HIPCHK(nullptr, hipMemsetAsync(c->dflags_set[k2], 0, B * sizeof(uint32_t), c->stream));
    }
    if (c->kernel == IRLOSC_KERNEL_ROW16) {
        constexpr size_t ZB = 64 * 1024;
        HIPCHK(nullptr, hipMalloc(&c->dzeros, ZB));
        HIPCHK(nullptr, hipMemsetAsync(c->dzeros, 0, ZB, c->stream));
        for (int k2 = 0; k2 < R16_TRAIN; ++k2) HIPCHK(nullptr, hipMalloc((void**)&c->dr16_list[k2], B * sizeof(int32_t)));
        HIPCHK(nullptr, hipMalloc((void**)&c->dr16_count, R16_TRAIN * sizeof(int32_t)));
        {   // part 1 of the task signal runs as a pass ahead of the row16 kernel (IRLOSC_TASK_PASS=0: in the kernel; A/B, tests); its
            // rows buffers are allocated by the first train that needs them (ensure_trows)
            const char* e = getenv("IRLOSC_TASK_PASS");
            c->task_pass = !(e && !strcmp(e, "0"));
            const char* ov = getenv("IRLOSC_R16_OVERLAP");
            c->r16_overlap = !(ov && !strcmp(ov, "0"));
            const char* nb = getenv("IRLOSC_R16_BANKS");
            c->r16_xbanks = std::max(1, std::min(nb ? atoi(nb) : 2, 1 + irlosc_ctx::MAX_XBANKS)) - 1;
        }
        HIPCHK(nullptr, hipMemsetAsync(c->dr16_count, 0, R16_TRAIN * sizeof(int32_t), c->stream));
    }
    c->du = c->du_set[0];
    c->dflags = c->dflags_set[0];
    HIPCHK(nullptr, hipMalloc((void**)&c->dsym, 2 * sizeof(int32_t)));
    HIPCHK(nullptr, hipMalloc((void**)&c->dstruct, sizeof(int32_t)));
    c->tree_ok.assign(c->cfg.n_slots, 0);
    row16_tree_masks(c->tree_masks.mrow, &c->tree_masks.jcols);
    {
        const char* e = getenv("IRLOSC_TREE");
        c->tree_enabled = !(e && !strcmp(e, "0"));
    }
    HIPCHK(nullptr, hipMalloc(&c->dgains, B * nd * IRLOSC_GAIN_WORDS * e));
    HIPCHK(nullptr, hipMalloc(&c->dnullkv, B * e));
    if (c->kernel != IRLOSC_KERNEL_GENERIC && getenv("IRLOSC_PHASE_TIMING"))     // debug aid: cycles per kernel phase
        HIPCHK(nullptr, hipMalloc((void**)&c->ddbg, (B / 4 + 1) * 10 * sizeof(unsigned long long)));
    HIPCHK(nullptr, hipStreamSynchronize(c->stream));
    return IRLOSC_OK;
}

static bool row16_supported(const irlosc_ctx* c) {
    return row16_kernel_supports(c->cfg.dtype, c->cfg.n, c->k, c->cfg.ndev);
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
    if (cfg->kernel == IRLOSC_KERNEL_REMOVED_GROUP) {
        delete c;
        return fail(nullptr, IRLOSC_ERR_ARG, "kernel id 2 (the fp32-arithmetic group kernel of ABI versions 1-2) was removed in ABI version 3: its error is "
                    "eps32 * cond(J M^-1 J^T), 14 %% of physical instances missed the 1e-5 contract; float32 RECORDS run on IRLOSC_KERNEL_AUTO (fp64 arithmetic)");
    }
    if (cfg->kernel == IRLOSC_KERNEL_ROW16 && !row16_supported(c)) {
        delete c;
        return fail(nullptr, IRLOSC_ERR_ARG, "row16 kernel not available for dtype=%d n=%d k=%d ndev=%d", cfg->dtype, cfg->n, k, cfg->ndev);
    }
    // Every kernel computes in fp64 (the reference's arithmetic, and what north_star's 1e-5 needs).  AUTO = the row16 kernel where the
    // shape has one -- on float64 records, and on float32 records too (the "mixed" path: fp32 storage, fp64 arithmetic) -- else the
    // generic kernel.
    c->kernel = IRLOSC_KERNEL_GENERIC;
    if (cfg->kernel == IRLOSC_KERNEL_ROW16) c->kernel = IRLOSC_KERNEL_ROW16;
    else if (cfg->kernel == IRLOSC_KERNEL_AUTO && row16_supported(c)) c->kernel = IRLOSC_KERNEL_ROW16;
    char nm[96];
    const bool mixed = c->kernel == IRLOSC_KERNEL_ROW16 && cfg->dtype == IRLOSC_F32;
    snprintf(nm, sizeof nm, "%s_%s_n%d_k%d",
             c->kernel == IRLOSC_KERNEL_ROW16 ? "osc_row16" : "osc_generic",
             mixed ? "f32in_f64" : cfg->dtype == IRLOSC_F64 ? "f64" : "f32", cfg->n, k);
    c->kernel_name = nm;
    c->kernel_class = c->kernel == IRLOSC_KERNEL_GENERIC ? IRLOSC_CLASS_GENERIC
                      : row16_kernel_exact(cfg->n, k, cfg->ndev) ? IRLOSC_CLASS_ROW16 : IRLOSC_CLASS_ROW16_PADDED;
    if (c->kernel_class == IRLOSC_CLASS_ROW16_PADDED) {      // the tier the launches will pick (tu_row16_pad_impl.hpp)
        snprintf(nm, sizeof nm, "_ndev%d_pad%d", cfg->ndev, row16_pad_tier(k));
        c->kernel_name += nm;
    }
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

// The throughput kernels read row j of M as its column j (include/irlosc.h, contracts): an asymmetric M would give a wrong
// answer without any flag, so records that come from the HOST are checked before they are accepted (the generic kernel uses
// M as given, like osc.py:49,151, and takes anything).  Small batches on the host, on the caller's own array (B = 1: under a
// microsecond, no kernel in the tick); large ones on the device, one pass over M behind the copy.
static constexpr int SYM_HOST_MAX_B = 32;
static bool sym_applies(const irlosc_ctx* c) { return c->kernel != IRLOSC_KERNEL_GENERIC; }

template <typename T>
static int symmetry_host_t(irlosc_ctx* c, const T* M, int B) {
    const int n = c->cfg.n;
    for (int b = 0; b < B; ++b) {
        const T* Mb = M + (size_t)b * n * n;
        double asym = 0.0, scale = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const double v = (double)Mb[i * n + j], w = (double)Mb[j * n + i];
                if (!std::isfinite(v) || !std::isfinite(w)) continue;      // a diverged robot is the kernel's business (per-instance flags)
                asym = std::max(asym, std::fabs(v - w));
                scale = std::max(scale, std::fabs(v));
            }
        if (asym > 1e-6 * std::max(scale, 1e-300))
            return fail(c, IRLOSC_ERR_ARG, "M of instance %d is not symmetric (max |M - M^T| = %.3g): the %s kernel reads rows of M as "
                        "columns; use IRLOSC_KERNEL_GENERIC for a non-symmetric M", b, asym, c->kernel_name.c_str());
    }
    return IRLOSC_OK;
}
static int symmetry_host(irlosc_ctx* c, const void* M, int B) {
    return c->cfg.dtype == IRLOSC_F64 ? symmetry_host_t<double>(c, (const double*)M, B) : symmetry_host_t<float>(c, (const float*)M, B);
}
// Device probe: zeroes the two result words at `dres` and enqueues the pass; the caller brings them back with whatever copy
// it makes anyway and hands them to symmetry_verdict.
static int symmetry_probe(irlosc_ctx* c, const void* dM, int B, int32_t* dres, hipStream_t st) {
    HIPCHK(c, hipMemsetAsync(dres, 0, 2 * sizeof(int32_t), st));
    const int rc = c->cfg.dtype == IRLOSC_F64 ? launch_symmetry_probe<double>((const double*)dM, c->cfg.n, B, dres, st)
                                              : launch_symmetry_probe<float>((const float*)dM, c->cfg.n, B, dres, st);
    HIPCHK(c, (hipError_t)rc);
    return IRLOSC_OK;
}
static int symmetry_verdict(irlosc_ctx* c, const int32_t res[2]) {
    if (res[0] > 0)
        return fail(c, IRLOSC_ERR_ARG, "M of instance %d is not symmetric (%d instance(s) with max |M - M^T| > 1e-6 max |M|): the %s "
                    "kernel reads rows of M as columns; use IRLOSC_KERNEL_GENERIC for a non-symmetric M", 0x7fffffff - res[1], res[0],
                    c->kernel_name.c_str());
    return IRLOSC_OK;
}

// the tree-structured form of the row16 kernel applies to the records of this slot
static bool slot_tree(const irlosc_ctx* c, int slot) {
    return c->tree_enabled && c->kernel == IRLOSC_KERNEL_ROW16 && c->tree_ok[slot] != 0;
}

// Zero pattern of the records in a slot (synchronous; a throughput feature: batches under 64 instances keep the dense form).
// The pattern is that of the compiled Dual-UR5 tree, so the question only arises for its shape (n = 25).
static int structure_probe(irlosc_ctx* c, int slot, int B) {
    c->tree_ok[slot] = 0;
    if (!c->tree_enabled || c->kernel != IRLOSC_KERNEL_ROW16 || B < 64) return IRLOSC_OK;
    int32_t bad = 0;
    HIPCHK(c, hipMemsetAsync(c->dstruct, 0, sizeof(int32_t), c->stream));
    const int rc = c->cfg.dtype == IRLOSC_F64
        ? launch_structure_probe<double>((const double*)c->dM[slot], (const double*)c->dJ[slot], c->cfg.n, c->k, B, c->tree_masks, c->dstruct, c->stream)
        : launch_structure_probe<float>((const float*)c->dM[slot], (const float*)c->dJ[slot], c->cfg.n, c->k, B, c->tree_masks, c->dstruct, c->stream);
    HIPCHK(c, (hipError_t)rc);
    HIPCHK(c, hipMemcpyAsync(&bad, c->dstruct, sizeof bad, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->tree_ok[slot] = bad == 0;
    return IRLOSC_OK;
}

extern "C" int irlosc_probe_structure(irlosc_ctx* c, int32_t slot, int32_t B) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (B > std::max(0, c->uploaded[slot]))
        return fail(c, IRLOSC_ERR_STATE, "slot %d holds records of %d instances, probe asked for %d", slot, std::max(0, c->uploaded[slot]), B);
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    rc = structure_probe(c, slot, B);
    if (rc) return rc;
    return slot_tree(c, slot) ? 1 : 0;
}

extern "C" int irlosc_slot_structure(const irlosc_ctx* c, int32_t slot) {
    if (!c || slot < 0 || slot >= c->cfg.n_slots) return 0;
    return slot_tree(c, slot) ? 1 : 0;
}

extern "C" int irlosc_upload(irlosc_ctx* c, int32_t slot, int32_t B, const void* M, const void* J, const void* dq,
                             const void* bias, const void* ee_pose, const void* wrench) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (B == 0) { c->uploaded[slot] = -1; return IRLOSC_OK; }
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
    c->uploaded[slot] = 0;                              // nothing usable in the slot until the records are accepted
    if (!c->fused_away.empty()) c->fused_away[slot] = 0;      // (and if they are refused, that is why the slot is empty -- not an earlier fused step)
    if (sym_applies(c)) {
        int rcs;
        if (B <= SYM_HOST_MAX_B) {
            rcs = symmetry_host(c, M, B);
        } else {
            int32_t res[2] = {0, 0};
            rcs = symmetry_probe(c, c->dM[slot], B, c->dsym, c->stream);
            if (rcs) return rcs;
            HIPCHK(c, hipMemcpyAsync(res, c->dsym, sizeof res, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            rcs = symmetry_verdict(c, res);
        }
        if (rcs) return rcs;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rcp = structure_probe(c, slot, B);
    if (rcp) return rcp;
    c->uploaded[slot] = B;
    if (!c->fused_away.empty()) c->fused_away[slot] = 0;
    return IRLOSC_OK;
}

// Enqueue the assembly kernel: dptr = device pointers {qM, qvel, qfrc_bias, jacp, jacr, ee_xpos, ee_xquat, site_xmat, sensordata}.
template <typename T>
static int assemble_launch(irlosc_ctx* c, int slot, int B, const irlosc_raw_desc* rd, const void* const* dptr, hipStream_t st,
                           const irlosc_qm_layout* qml = nullptr) {
    RawDesc d;
    memset(&d, 0, sizeof d);
    if (qml) {
        d.nM = qml->nM;
        for (int j = 0; j < IRLOSC_MAX_NV; ++j) d.pos[j] = -1;
        for (int j = 0; j < rd->nv; ++j) { d.madr[j] = (int16_t)qml->dof_Madr[j]; d.par[j] = (int16_t)qml->dof_parentid[j]; }
        for (int i = 0; i < c->cfg.n; ++i) d.pos[rd->joint_ids[i]] = (int16_t)i;
    }
    d.nv = rd->nv; d.n_sensor = rd->n_sensor; d.n = c->cfg.n; d.k = c->k; d.ndev = c->cfg.ndev;
    for (int i = 0; i < c->cfg.n; ++i) { d.joint_ids[i] = rd->joint_ids[i]; d.dq_src[i] = rd->dq_src[i]; }
    for (int dv = 0; dv < c->cfg.ndev; ++dv) {
        d.ft_force0[dv] = rd->ft_force0[dv]; d.ft_torque0[dv] = rd->ft_torque0[dv];
        for (int i = 0; i < 6; ++i) if (c->cfg.ctrlr_dof[dv][i]) d.dofmask[dv] |= 1u << i;
    }
    const bool ft = dptr[7] && dptr[8] && rd->n_sensor > 0;
    RawPtrs<T> r;
    r.qM = (const T*)dptr[0]; r.qvel = (const T*)dptr[1]; r.qfrc_bias = (const T*)dptr[2];
    r.jacp = (const T*)dptr[3]; r.jacr = (const T*)dptr[4]; r.ee_xpos = (const T*)dptr[5]; r.ee_xquat = (const T*)dptr[6];
    r.site_xmat = ft ? (const T*)dptr[7] : nullptr; r.sensordata = ft ? (const T*)dptr[8] : nullptr;
    r.M = (T*)c->dM[slot]; r.J = (T*)c->dJ[slot]; r.dq = (T*)c->ddq[slot]; r.bias = (T*)c->dbias[slot];
    r.ee = (T*)c->dee[slot]; r.wrench = (T*)c->dwrench[slot];
    HIPCHK(c, (hipError_t)launch_assemble<T>(d, r, B, st));
    return IRLOSC_OK;
}

static int check_raw_desc(irlosc_ctx* c, const irlosc_raw_desc* rd) {
    if (rd->nv < 1 || rd->n_sensor < 0) return fail(c, IRLOSC_ERR_ARG, "bad nv / n_sensor");
    for (int i = 0; i < c->cfg.n; ++i) {
        if (rd->joint_ids[i] < 0 || rd->joint_ids[i] >= rd->nv) return fail(c, IRLOSC_ERR_ARG, "joint_ids[%d] out of [0,nv)", i);
        if (rd->dq_src[i] >= rd->nv) return fail(c, IRLOSC_ERR_ARG, "dq_src[%d] out of range", i);
    }
    for (int dv = 0; dv < c->cfg.ndev; ++dv) {
        const int f0 = rd->ft_force0[dv], t0 = rd->ft_torque0[dv];
        if ((f0 >= 0 && f0 + 3 > rd->n_sensor) || (t0 >= 0 && t0 + 3 > rd->n_sensor))
            return fail(c, IRLOSC_ERR_ARG, "F/T sensor slice of device %d exceeds n_sensor", dv);
    }
    return IRLOSC_OK;
}

template <typename T>
static int upload_raw_t(irlosc_ctx* c, int slot, int B, const irlosc_raw_desc* rd, const void* qM, const void* qvel,
                        const void* qfrc_bias, const void* jacp, const void* jacr, const void* ee_xpos,
                        const void* ee_xquat, const void* site_xmat, const void* sensordata, const irlosc_qm_layout* qml = nullptr) {
    const size_t b = (size_t)B, nv = (size_t)rd->nv, nd = (size_t)c->cfg.ndev, ns = (size_t)rd->n_sensor, e = sizeof(T);
    const bool ft = site_xmat && sensordata && ns > 0;
    // staging layout: qM (dense nv x nv, or MuJoCo's nM-entry form) | qvel | qfrc_bias | jacp | jacr | ee_xpos | ee_xquat | site_xmat | sensordata
    const size_t sz[9] = {qml ? b * (size_t)qml->nM * e : b * nv * nv * e, b * nv * e, b * nv * e, b * nd * 3 * nv * e, b * nd * 3 * nv * e,
                          b * nd * 3 * e, b * nd * 4 * e, ft ? b * nd * 9 * e : 0, ft ? b * ns * e : 0};
    const void* src[9] = {qM, qvel, qfrc_bias, jacp, jacr, ee_xpos, ee_xquat, site_xmat, sensordata};
    size_t off[9], total = 0;
    for (int i = 0; i < 9; ++i) { off[i] = total; total += (sz[i] + 255) & ~(size_t)255; }
    if (total > c->draw_bytes) {
        if (c->draw) HIPCHK(c, hipFree(c->draw));
        c->draw = nullptr; c->draw_bytes = 0;
        HIPCHK(c, hipMalloc(&c->draw, total));
        c->draw_bytes = total;
    }
    unsigned char* base = (unsigned char*)c->draw;
    for (int i = 0; i < 9; ++i)
        if (sz[i]) HIPCHK(c, hipMemcpyAsync(base + off[i], src[i], sz[i], hipMemcpyHostToDevice, c->stream));
    const void* dptr[9];
    for (int i = 0; i < 9; ++i) dptr[i] = sz[i] ? (const void*)(base + off[i]) : nullptr;
    int rc = assemble_launch<T>(c, slot, B, rd, dptr, c->stream, qml);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return IRLOSC_OK;
}

// MuJoCo's own form of M: validate the layout (every run stays inside [0, nM), the tree is a forest numbered parents first)
static int check_qm_layout(irlosc_ctx* c, const irlosc_raw_desc* rd, const irlosc_qm_layout* q) {
    if (rd->nv > IRLOSC_MAX_NV) return fail(c, IRLOSC_ERR_ARG, "nv=%d exceeds IRLOSC_MAX_NV=%d", rd->nv, IRLOSC_MAX_NV);
    if (q->nM < rd->nv || q->nM > 32767) return fail(c, IRLOSC_ERR_ARG, "nM=%d out of range for nv=%d", q->nM, rd->nv);
    for (int i = 0; i < rd->nv; ++i) {
        if (q->dof_parentid[i] >= i || q->dof_parentid[i] < -1) return fail(c, IRLOSC_ERR_ARG, "dof_parentid[%d]=%d: a parent precedes its child (or is -1)", i, q->dof_parentid[i]);
        int len = 0;
        for (int j = i; j >= 0; j = q->dof_parentid[j]) ++len;
        if (q->dof_Madr[i] < 0 || q->dof_Madr[i] + len > q->nM) return fail(c, IRLOSC_ERR_ARG, "dof_Madr[%d]=%d + %d entries exceeds nM=%d", i, q->dof_Madr[i], len, q->nM);
    }
    return IRLOSC_OK;
}

extern "C" int irlosc_upload_raw_sparse(irlosc_ctx* c, int32_t slot, int32_t B, const irlosc_raw_desc* rd, const irlosc_qm_layout* qml,
                                        const void* qM, const void* qvel, const void* qfrc_bias, const void* jacp, const void* jacr,
                                        const void* ee_xpos, const void* ee_xquat, const void* site_xmat, const void* sensordata) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (B == 0) { c->uploaded[slot] = -1; return IRLOSC_OK; }
    if (!rd || !qml || !qM || !qvel || !qfrc_bias || !jacp || !jacr || !ee_xpos || !ee_xquat)
        return fail(c, IRLOSC_ERR_ARG, "desc, qm layout, qM, qvel, qfrc_bias, jacp, jacr, ee_xpos and ee_xquat are required");
    rc = check_raw_desc(c, rd);
    if (!rc) rc = check_qm_layout(c, rd, qml);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    rc = c->cfg.dtype == IRLOSC_F64
             ? upload_raw_t<double>(c, slot, B, rd, qM, qvel, qfrc_bias, jacp, jacr, ee_xpos, ee_xquat, site_xmat, sensordata, qml)
             : upload_raw_t<float>(c, slot, B, rd, qM, qvel, qfrc_bias, jacp, jacr, ee_xpos, ee_xquat, site_xmat, sensordata, qml);
    if (rc) return rc;
    c->has_wrench[slot] = 1;
    rc = structure_probe(c, slot, B);      // (symmetric by construction: the expansion mirrors every entry)
    if (rc) return rc;
    c->uploaded[slot] = B;
    if (!c->fused_away.empty()) c->fused_away[slot] = 0;
    return IRLOSC_OK;
}

extern "C" int irlosc_upload_raw(irlosc_ctx* c, int32_t slot, int32_t B, const irlosc_raw_desc* rd, const void* qM,
                                 const void* qvel, const void* qfrc_bias, const void* jacp, const void* jacr,
                                 const void* ee_xpos, const void* ee_xquat, const void* site_xmat,
                                 const void* sensordata) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (B == 0) { c->uploaded[slot] = -1; return IRLOSC_OK; }
    if (!rd || !qM || !qvel || !qfrc_bias || !jacp || !jacr || !ee_xpos || !ee_xquat)
        return fail(c, IRLOSC_ERR_ARG, "desc, qM, qvel, qfrc_bias, jacp, jacr, ee_xpos and ee_xquat are required");
    rc = check_raw_desc(c, rd);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    rc = c->cfg.dtype == IRLOSC_F64
             ? upload_raw_t<double>(c, slot, B, rd, qM, qvel, qfrc_bias, jacp, jacr, ee_xpos, ee_xquat, site_xmat, sensordata)
             : upload_raw_t<float>(c, slot, B, rd, qM, qvel, qfrc_bias, jacp, jacr, ee_xpos, ee_xquat, site_xmat, sensordata);
    if (rc) return rc;
    c->has_wrench[slot] = 1;
    rc = structure_probe(c, slot, B);
    if (rc) return rc;
    c->uploaded[slot] = B;
    if (!c->fused_away.empty()) c->fused_away[slot] = 0;
    return IRLOSC_OK;
}

extern "C" int irlosc_assemble_device(irlosc_ctx* c, int32_t slot, int32_t B, const irlosc_raw_desc* rd, const void* qM,
                                      const void* qvel, const void* qfrc_bias, const void* jacp, const void* jacr,
                                      const void* ee_xpos, const void* ee_xquat, const void* site_xmat,
                                      const void* sensordata, void* hip_stream) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (B == 0) { c->uploaded[slot] = -1; return IRLOSC_OK; }
    if (!rd || !qM || !qvel || !qfrc_bias || !jacp || !jacr || !ee_xpos || !ee_xquat)
        return fail(c, IRLOSC_ERR_ARG, "desc, qM, qvel, qfrc_bias, jacp, jacr, ee_xpos and ee_xquat are required");
    rc = check_raw_desc(c, rd);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    const void* dptr[9] = {qM, qvel, qfrc_bias, jacp, jacr, ee_xpos, ee_xquat, site_xmat, sensordata};
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    rc = c->cfg.dtype == IRLOSC_F64 ? assemble_launch<double>(c, slot, B, rd, dptr, st)
                                    : assemble_launch<float>(c, slot, B, rd, dptr, st);
    if (rc) return rc;
    c->has_wrench[slot] = 1;
    c->tree_ok[slot] = 0;          // enqueued on the caller's stream: no synchronous look at what it writes
    c->uploaded[slot] = B;
    if (!c->fused_away.empty()) c->fused_away[slot] = 0;
    return IRLOSC_OK;
}

extern "C" int irlosc_set_targets(irlosc_ctx* c, int32_t slot, int32_t B, const void* tgt_pose, const void* tgt_vel) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (B == 0) { c->targeted[slot] = -1; return IRLOSC_OK; }
    if (!tgt_pose) return fail(c, IRLOSC_ERR_ARG, "tgt_pose is NULL");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    const size_t b = (size_t)B, nd = (size_t)c->cfg.ndev, e = c->esz;
    HIPCHK(c, hipMemcpyAsync(c->dtgt[slot], tgt_pose, b * nd * 7 * e, hipMemcpyHostToDevice, c->stream));
    if (tgt_vel) HIPCHK(c, hipMemcpyAsync(c->dtvel[slot], tgt_vel, b * nd * 6 * e, hipMemcpyHostToDevice, c->stream));
    c->has_tvel[slot] = tgt_vel != nullptr;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->targeted[slot] = B;
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
    p.padded = c->kernel_class == IRLOSC_CLASS_ROW16_PADDED ? 1 : 0;      // decided once, at irlosc_create (IRLOSC_FORCE_PAD is read there)
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

// fp64-arithmetic path: one launch for a train of n steps (ps[i] complete with its own outputs), whatever the storage
// type T of the records.  All instances run on the row16 kernel, the truncated pseudo-inverse included; the few it gives
// up on (net of eigen-candidates full, degenerate A) are recomputed by the generic kernel (Jacobi, fp64 arithmetic) from
// the lists it leaves behind.
// The rows buffers of the task pass, steps 0 .. n - 1 of a train: allocated on first use (8 MB per step at 65 536 instances; a context
// that only runs the fused path or single ticks never pays for all eight).  Out of memory: that train computes part 1 of the task signal
// in the row16 kernel (trows = nullptr), same results.
static bool ensure_trows(irlosc_ctx* c, int n) {
    if (!c->task_pass) return false;
    for (int i = 0; i < n; ++i) {
        if (c->dtrows[i]) continue;
        if (hipMalloc((void**)&c->dtrows[i], (size_t)c->cfg.max_batch * 16 * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            c->dtrows[i] = nullptr;
            return false;
        }
    }
    return true;
}

static int ensure_xbank(irlosc_ctx* c, int which, int n, bool fused);

template <typename T>
static int row16_train(irlosc_ctx* c, const KParams<T>* ps, int n, bool tree, hipStream_t st, const int* pos = nullptr,
                       bool reset = true, const irlosc_ctx::Bank* bk = nullptr) {
    // pos[i] = step of the whole train that sub-train step i is (a train that mixes tree-form and dense slots goes out as two
    // sub-trains): give-up list and counter are those of the ORIGINAL step, and only the first sub-train zeroes the counters, so
    // that irlosc_giveup_counts reports every step of the train at its own index.
    if (n < 1 || n > R16_TRAIN) return fail(c, IRLOSC_ERR_STATE, "train of %d steps", n);
    Row16Train<T> tr;
    memset(&tr, 0, sizeof tr);
    int32_t* cnt = bk ? bk->count : c->dr16_count;      // (bk: the second bank of an overlapped train, see irlosc_ctx::Bank)
    if (reset) HIPCHK(c, hipMemsetAsync(cnt, 0, R16_TRAIN * sizeof(int32_t), st));
    c->count_cur = cnt;
    const bool rows = bk ? (c->task_pass && bk->trows[n - 1] != nullptr) : ensure_trows(c, n);
    for (int i = 0; i < n; ++i) {
        const int o = pos ? pos[i] : i;
        tr.p[i] = ps[i];
        tr.x[i] = Row16Extra{c->dzeros, bk ? bk->list[o] : c->dr16_list[o], cnt + o, nullptr, nullptr, nullptr, c->span_next,
                             rows ? (bk ? bk->trows[i] : c->dtrows[i]) : nullptr};
    }
    if (c->tev_begin) HIPCHK(c, hipEventRecord(c->tev_begin, st));
    int rc = launch_row16<T>(tr, n, tree, st);
    if (rc) return fail(c, IRLOSC_ERR_HIP, "row16 kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    HIPCHK(c, (hipError_t)launch_row16_worklist<T>(tr, n, nullptr, st));
    if (c->tev_end) HIPCHK(c, hipEventRecord(c->tev_end, st));
    return IRLOSC_OK;
}

template <typename T>
static int launch_t(irlosc_ctx* c, int B, const void* M, const void* J, const void* dq, const void* bias,
                    const void* ee, const void* tgt, const void* tvel, const void* wrench, void* u,
                    uint32_t* flags, hipStream_t st, bool tree) {
    KParams<T> p;
    fill_params<T>(c, p, B, M, J, dq, bias, ee, tgt, tvel, wrench, u, flags);
    if (c->kernel == IRLOSC_KERNEL_ROW16) return row16_train<T>(c, &p, 1, tree, st);
    HIPCHK(c, (hipError_t)launch_generic<T>(p, B, st));
    return IRLOSC_OK;
}

static int launch(irlosc_ctx* c, int B, const void* M, const void* J, const void* dq, const void* bias,
                  const void* ee, const void* tgt, const void* tvel, const void* wrench, void* u,
                  uint32_t* flags, hipStream_t st, bool tree = false) {
    if (B == 0) return IRLOSC_OK;
    if (c->gains_nb == 0) return fail(c, IRLOSC_ERR_STATE, "irlosc_set_gains has not been called");
    if (c->cfg.dtype == IRLOSC_F64) return launch_t<double>(c, B, M, J, dq, bias, ee, tgt, tvel, wrench, u, flags, st, tree);
    return launch_t<float>(c, B, M, J, dq, bias, ee, tgt, tvel, wrench, u, flags, st, tree);
}

// A step over B instances needs B instances of state AND of targets in the slot (stale or uninitialised HBM otherwise).
static int check_slot_filled(irlosc_ctx* c, int slot, int B) {
    if (!c->uploaded[slot] && !c->fused_away.empty() && c->fused_away[slot])
        return fail(c, IRLOSC_ERR_STATE, "slot %d holds no dense records: the preceding fused irlosc_step_from_q / irlosc_step_resident_from_q "
                    "invalidated them (it never writes M / J); run irlosc_frontend or an upload first", slot);
    if (!c->uploaded[slot] || !c->targeted[slot])
        return fail(c, IRLOSC_ERR_STATE, "slot %d: irlosc_upload and irlosc_set_targets must precede a step", slot);
    if (B > std::max(0, c->uploaded[slot]) || B > std::max(0, c->targeted[slot]))
        return fail(c, IRLOSC_ERR_STATE, "slot %d holds state for %d and targets for %d instances, step asked for %d", slot,
                    std::max(0, c->uploaded[slot]), std::max(0, c->targeted[slot]), B);
    return IRLOSC_OK;
}

static int launch_slot(irlosc_ctx* c, int slot, int B) {
    int rcf = check_slot_filled(c, slot, B);
    if (rcf) return rcf;
    return launch(c, B, c->dM[slot], c->dJ[slot], c->ddq[slot], c->dbias[slot], c->dee[slot], c->dtgt[slot],
                  c->has_tvel[slot] ? c->dtvel[slot] : nullptr, c->has_wrench[slot] ? c->dwrench[slot] : nullptr,
                  c->du, c->dflags, c->stream, slot_tree(c, slot));
}

extern "C" int irlosc_download(irlosc_ctx* c, int32_t B, void* u_host, uint32_t* flags_host) {
    if (!c) return IRLOSC_ERR_ARG;
    if (B < 0 || B > c->cfg.max_batch) return fail(c, IRLOSC_ERR_ARG, "B out of range");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    if (u_host && B) HIPCHK(c, hipMemcpyAsync(u_host, c->du, (size_t)B * c->cfg.n * c->esz, hipMemcpyDeviceToHost, c->stream));
    if (flags_host && B) HIPCHK(c, hipMemcpyAsync(flags_host, c->dflags, (size_t)B * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->ddbg && B >= 16) {   // debug: mean cycles per phase over all waves
        const bool r16 = c->kernel == IRLOSC_KERNEL_ROW16;
        const bool lanef = getenv("IRLOSC_PHASE_LANE") != nullptr;      // a -DIRLOSC_LANE_STAMPS build: the lane kernel of the fused path stamped
        const int tiles = lanef ? B / 64 : r16 ? B / 4 : B / 16;
        std::vector<unsigned long long> h((size_t)tiles * 10);
        HIPCHK(c, hipMemcpy(h.data(), c->ddbg, h.size() * 8, hipMemcpyDeviceToHost));
        static const char* nm_g[7] = {"vec-wait", "M-stream+Cholesky", "J+fwd-subst", "task-error", "A=YtY", "kxk", "torques+store"};
        static const char* nm_r[7] = {"loads+task-error", "J->LDS", "main-loop", "A=YtY", "kxk", "eigen", "torques+store"};
        static const char* nm_l[7] = {"first-requests+task-rows", "recursion", "kxk", "records", "Jt+sums", "torques+store", "-"};
        const char* const* nm = lanef ? nm_l : r16 ? nm_r : nm_g;
        double acc[7] = {0}, rt = 0;
        unsigned long long rmin = ~0ull, rmax = 0;
        for (int t = 0; t < tiles; ++t) {
            for (int i = 0; i < 7; ++i) acc[i] += (double)(h[(size_t)t * 10 + i + 1] - h[(size_t)t * 10 + i]);
            const unsigned long long r0 = h[(size_t)t * 10 + 8] & 0x0fffffffffffffffull;
            rt += (double)(h[(size_t)t * 10 + 9] - r0);
            rmin = std::min(rmin, r0);
            rmax = std::max(rmax, h[(size_t)t * 10 + 9]);
        }
        if (!r16) {   // where do blocks land?  XCC of tile t vs t % 8, and vs the XCC of tile t + 2048; start order of the second round
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

// `iters` steps on the row16 path, chained R16_TRAIN per launch (step i of a train writes output set i); events (if any)
// go around the launches from number `skip` on.  One kernel per launch, so a train whose slots do not ALL qualify for the
// tree-structured form is issued as two sub-trains -- the qualifying steps with the tree kernel, the others with the dense
// recursion -- instead of dropping every step to the dense recursion (one more launch, only when slots are mixed).
template <typename T>
static int row16_resident(irlosc_ctx* c, int first_slot, int B, int iters, const std::vector<hipEvent_t>* evs, int skip) {
    int done = 0, launch_no = 0;
    const hipEvent_t outer_b = c->tev_begin, outer_e = c->tev_end;      // irlosc_time_trains brackets a one-train call itself
    // more than one train, untimed: odd trains on the second bank / stream, so that their first waves fill the tail of the train before
    // (a call of exactly one full train already allocates the second bank: a caller's warm-up then pays for it, not its timed loop)
    int nx = 0;
    if (c->r16_overlap && !evs && iters >= R16_TRAIN) {
        const int want = std::min(c->r16_xbanks, std::max(1, (iters + R16_TRAIN - 1) / R16_TRAIN - 1));
        while (nx < want && ensure_xbank(c, nx, R16_TRAIN, false) == 0) ++nx;
    }
    if (nx) HIPCHK(c, hipEventRecord(c->ev_join, c->stream));
    for (int k = 0; k < nx; ++k) HIPCHK(c, hipStreamWaitEvent(c->xb[k].st, c->ev_join, 0));
    const irlosc_ctx::Bank* last_bank = nullptr;
    while (done < iters) {
        const int n = std::min((int)R16_TRAIN, iters - done);
        const int which = launch_no % (nx + 1);
        const irlosc_ctx::Bank* bk = which ? &c->xb[which - 1] : nullptr;
        KParams<T> ps[2][R16_TRAIN];        // [1]: steps whose slot qualifies for the tree form, [0]: the others
        int pos[2][R16_TRAIN];              // step of the train each sub-train step is
        int cnt[2] = {0, 0};
        for (int i = 0; i < n; ++i) {
            const int slot = (first_slot + done + i) % c->cfg.n_slots;
            int rcf = check_slot_filled(c, slot, B);
            if (rcf) return rcf;
            const int kind = slot_tree(c, slot) ? 1 : 0;
            pos[kind][cnt[kind]] = i;
            fill_params<T>(c, ps[kind][cnt[kind]++], B, c->dM[slot], c->dJ[slot], c->ddq[slot], c->dbias[slot], c->dee[slot], c->dtgt[slot],
                           c->has_tvel[slot] ? c->dtvel[slot] : nullptr, c->has_wrench[slot] ? c->dwrench[slot] : nullptr,
                           bk ? bk->u[i] : c->du_set[i], bk ? bk->flags[i] : c->dflags_set[i]);
        }
        const bool timed = evs && launch_no >= skip && 2 * (launch_no - skip) + 1 < (int)evs->size();
        hipEvent_t eb = timed ? (*evs)[2 * (launch_no - skip)] : outer_b, ee = timed ? (*evs)[2 * (launch_no - skip) + 1] : outer_e;
        const int first = cnt[1] ? 1 : 0, last = cnt[0] ? 0 : 1;      // order: tree sub-train, then dense sub-train
        for (int kind = 1; kind >= 0; --kind) {
            if (!cnt[kind]) continue;
            c->tev_begin = kind == first ? eb : nullptr;              // the event pair brackets the whole train
            c->tev_end = kind == last ? ee : nullptr;
            int rc = row16_train<T>(c, ps[kind], cnt[kind], kind == 1, bk ? bk->st : c->stream, pos[kind], kind == first, bk);
            c->tev_begin = c->tev_end = nullptr;
            if (rc) return rc;
        }
        last_bank = bk;
        c->cur = n - 1;
        done += n;
        ++launch_no;
    }
    for (int k = 0; k < nx; ++k) {
        HIPCHK(c, hipEventRecord(c->xb[k].done, c->xb[k].st));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->xb[k].done, 0));
    }
    c->du = last_bank ? last_bank->u[c->cur] : c->du_set[c->cur];
    c->dflags = last_bank ? last_bank->flags[c->cur] : c->dflags_set[c->cur];
    return IRLOSC_OK;
}

extern "C" int irlosc_step_resident(irlosc_ctx* c, int32_t first_slot, int32_t B, int32_t iters, float* ms_total,
                                    float* ms_kernel_avg) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, first_slot, B);
    if (rc) return rc;
    if (iters < 1) return fail(c, IRLOSC_ERR_ARG, "iters must be >= 1");
    if (c->gains_nb == 0) return fail(c, IRLOSC_ERR_STATE, "irlosc_set_gains has not been called");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (c->kernel == IRLOSC_KERNEL_ROW16 && B > 0) {
        rc = c->cfg.dtype == IRLOSC_F64 ? row16_resident<double>(c, first_slot, B, iters, nullptr, 0)
                                        : row16_resident<float>(c, first_slot, B, iters, nullptr, 0);
        if (rc) return rc;
    } else {
        for (int i = 0; i < iters; ++i) {
            rc = launch_slot(c, (first_slot + i) % c->cfg.n_slots, B);
            if (rc) return rc;
        }
    }
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (ms_total) *ms_total = ms;
    if (ms_kernel_avg) *ms_kernel_avg = ms / (float)iters;
    return IRLOSC_OK;
}

extern "C" int irlosc_steps_per_launch(const irlosc_ctx* c) {
    if (!c) return IRLOSC_ERR_ARG;
    return c->kernel == IRLOSC_KERNEL_GENERIC ? 1 : c->train;
}

extern "C" int irlosc_time_dominant_kernel(irlosc_ctx* c, int32_t slot, int32_t B, int32_t iters, float* ms_avg) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (iters < 1 || iters > 256 || !ms_avg) return fail(c, IRLOSC_ERR_ARG, "iters must be in [1,256] and ms_avg non-NULL");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    if (c->kernel == IRLOSC_KERNEL_GENERIC || B == 0) {        // generic path: a step IS the dominant kernel
        float tot = 0.f;
        rc = irlosc_step_resident(c, slot, B, iters, &tot, ms_avg);
        return rc;
    }
    // row16 path: the same chained launches as irlosc_step_resident, with a HIP event pair around each train (task pass + row16 kernel +
    // give-up pass) -- what a rocprofv3 kernel trace of the timed region shows, so the two averages are comparable.  The first launch is
    // not timed.
    if (c->gains_nb == 0) return fail(c, IRLOSC_ERR_STATE, "irlosc_set_gains has not been called");
    const int launches = std::max(1, iters / c->train);
    while ((int)c->tev_pool.size() < 2 * launches) {
        hipEvent_t ev;
        HIPCHK(c, hipEventCreate(&ev));
        c->tev_pool.push_back(ev);
    }
    std::vector<hipEvent_t> evs(c->tev_pool.begin(), c->tev_pool.begin() + 2 * launches);
    rc = c->cfg.dtype == IRLOSC_F64 ? row16_resident<double>(c, slot, B, (launches + 1) * c->train, &evs, 1)
                                    : row16_resident<float>(c, slot, B, (launches + 1) * c->train, &evs, 1);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (int i = 0; i < launches; ++i) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, evs[2 * i], evs[2 * i + 1]));
        tot += ms;
    }
    *ms_avg = (float)(tot / launches);
    return IRLOSC_OK;
}

// Roofline evidence without a tracer (include/irlosc.h): per train one HIP event pair AND the wall-clock stamps the train's
// main kernel takes itself (first wave's start, last wave's end).
static int fused_resident(irlosc_ctx* c, int first_slot, int B, int iters);
static int ensure_xside(irlosc_ctx* c, int n);
extern "C" int irlosc_time_trains(irlosc_ctx* c, int32_t first_slot, int32_t B, int32_t ntrains, int32_t from_q, double* out) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, first_slot, B);
    if (rc) return rc;
    if (ntrains < 1 || ntrains > 4096 || !out) return fail(c, IRLOSC_ERR_ARG, "ntrains must be in [1,4096] and out non-NULL");
    if (c->kernel != IRLOSC_KERNEL_ROW16 || B < 1) return fail(c, IRLOSC_ERR_ARG, "irlosc_time_trains needs the row16 kernel and B >= 1");
    if (c->gains_nb == 0) return fail(c, IRLOSC_ERR_STATE, "irlosc_set_gains has not been called");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    const int spl = from_q ? c->fused_train : c->train;
    if (from_q && !(c->fused && ensure_xside(c, spl) == 0)) return fail(c, IRLOSC_ERR_STATE, "the fused path from joint coordinates is not available on this context");
    while ((int)c->tev_pool.size() < 2 * ntrains) {
        hipEvent_t ev;
        HIPCHK(c, hipEventCreate(&ev));
        c->tev_pool.push_back(ev);
    }
    if (c->dspan_cap < ntrains) {
        if (c->dspan) HIPCHK(c, hipFree(c->dspan));
        c->dspan = nullptr; c->dspan_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->dspan, (size_t)ntrains * R16_SPAN_WORDS * sizeof(unsigned long long)));
        c->dspan_cap = ntrains;
    }
    std::vector<unsigned long long> h((size_t)ntrains * R16_SPAN_WORDS);
    for (int i = 0; i < ntrains; ++i) {
        unsigned long long* t = h.data() + (size_t)i * R16_SPAN_WORDS;
        for (int s2 = 0; s2 < R16_SPAN_SLOTS; ++s2) { t[2 * s2] = ~0ull; t[2 * s2 + 1] = 0ull; }
        t[2 * R16_SPAN_SLOTS] = t[2 * R16_SPAN_SLOTS + 1] = 0ull;
    }
    HIPCHK(c, hipMemcpyAsync(c->dspan, h.data(), h.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // one untimed train first (clocks, caches, nothing rides on an idle machine), then the measured ones back to back
    for (int i = -1; i < ntrains && !rc; ++i) {
        if (i >= 0) {
            c->tev_begin = c->tev_pool[2 * i]; c->tev_end = c->tev_pool[2 * i + 1];
            c->span_next = c->dspan + (size_t)R16_SPAN_WORDS * i;
        }
        const int s0 = (first_slot + (i + 1) * spl) % c->cfg.n_slots;
        if (from_q) rc = fused_resident(c, s0, B, spl);
        else rc = c->cfg.dtype == IRLOSC_F64 ? row16_resident<double>(c, s0, B, spl, nullptr, 0) : row16_resident<float>(c, s0, B, spl, nullptr, 0);
        c->tev_begin = c->tev_end = nullptr;
        c->span_next = nullptr;
    }
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(h.data(), c->dspan, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long t0 = 0;
    for (int i = 0; i < ntrains; ++i) {
        unsigned long long lo = ~0ull, hi = 0ull;                    // over the stamp pairs of the train
        const unsigned long long* t = h.data() + (size_t)i * R16_SPAN_WORDS;
        for (int s2 = 0; s2 < R16_SPAN_SLOTS; ++s2) {
            lo = std::min(lo, t[2 * s2]);
            hi = std::max(hi, t[2 * s2 + 1]);
        }
        if (i == 0) t0 = lo;
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->tev_pool[2 * i], c->tev_pool[2 * i + 1]));
        out[4 * i] = ms;
        out[4 * i + 1] = (double)(lo - t0) / 100.0;                  // s_memrealtime: 100 MHz
        out[4 * i + 2] = (double)(hi - t0) / 100.0;
        out[4 * i + 3] = t[2 * R16_SPAN_SLOTS + 1] ? (double)t[2 * R16_SPAN_SLOTS] / ((double)t[2 * R16_SPAN_SLOTS + 1] / 100.0) : 0.0;   // cycles per us = MHz
    }
    return IRLOSC_OK;
}

// Instances the most recent step / train handed from the row16 kernel's in-wave eigen stage to the generic kernel (the give-up
// lists): counters of the last train, read back after a stream synchronisation.
extern "C" int irlosc_giveup_counts(irlosc_ctx* c, int32_t* out) {
    if (!c || !out) return IRLOSC_ERR_ARG;
    for (int i = 0; i < R16_TRAIN; ++i) out[i] = 0;
    if (c->kernel != IRLOSC_KERNEL_ROW16) return IRLOSC_OK;
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    HIPCHK(c, hipMemcpyAsync(out, c->count_cur ? c->count_cur : c->dr16_count, R16_TRAIN * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return IRLOSC_OK;
}

extern "C" int irlosc_sync(irlosc_ctx* c) {
    if (!c) return IRLOSC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return IRLOSC_OK;
}

// ---- rigid-body front end ------------------------------------------------------------------------------------------
extern "C" int irlosc_set_model(irlosc_ctx* c, const irlosc_model* m) {
    if (!c) return IRLOSC_ERR_ARG;
    if (!m) return fail(c, IRLOSC_ERR_ARG, "model is NULL");
    if (m->nb < 1 || m->nb > IRLOSC_MAX_BODIES) return fail(c, IRLOSC_ERR_ARG, "nb=%d out of [1,%d]", m->nb, IRLOSC_MAX_BODIES);
    if (m->nj != c->cfg.n) return fail(c, IRLOSC_ERR_ARG, "model has %d hinges but cfg.n = %d", m->nj, c->cfg.n);
    FeModel h;
    memset(&h, 0, sizeof h);
    h.nb = m->nb; h.nj = m->nj; h.ndev = c->cfg.ndev; h.k = c->k;
    std::vector<int> seen(m->nj, 0);
    for (int b = 0; b < m->nb; ++b) {
        const int par = m->parent[b], jb = m->joint_of_body[b];
        if (par >= b || par < -1) return fail(c, IRLOSC_ERR_ARG, "body %d: parent %d must precede it (or be -1)", b, par);
        if (jb < -1 || jb >= m->nj) return fail(c, IRLOSC_ERR_ARG, "body %d: hinge index %d out of range", b, jb);
        if (jb >= 0 && seen[jb]++) return fail(c, IRLOSC_ERR_ARG, "hinge %d sits on two bodies", jb);
        if (!(m->mass[b] >= 0.0)) return fail(c, IRLOSC_ERR_ARG, "body %d: negative mass", b);
        h.parent[b] = par; h.joint_of_body[b] = jb;
        h.depth[b] = par < 0 ? 0 : h.depth[par] + 1;
        h.maxdepth = std::max(h.maxdepth, h.depth[b]);
        h.anc_mask[b] = (par < 0 ? 0u : h.anc_mask[par]) | (jb >= 0 ? 1u << jb : 0u);
        if (jb >= 0) h.body_of_joint[jb] = b;
        for (int i = 0; i < 3; ++i) { h.pos[b][i] = m->pos[b][i]; h.ipos[b][i] = m->ipos[b][i]; h.inertia[b][i] = m->inertia[b][i]; }
        for (int i = 0; i < 4; ++i) { h.quat[b][i] = m->quat[b][i]; h.iquat[b][i] = m->iquat[b][i]; }
        h.mass[b] = m->mass[b];
    }
    for (int j = 0; j < m->nj; ++j) {
        if (!seen[j]) return fail(c, IRLOSC_ERR_ARG, "hinge %d sits on no body", j);
        for (int i = 0; i < 3; ++i) { h.jaxis[j][i] = m->jaxis[j][i]; h.jpos[j][i] = m->jpos[j][i]; }
        h.armature[j] = m->armature[j];
        for (int b = 0; b < m->nb; ++b) if ((h.anc_mask[b] >> j) & 1u) h.sub_mask[j] |= 1ull << b;
    }
    for (int i = 0; i < 3; ++i) h.gravity[i] = m->gravity[i];
    {   // derived tables of the lane-per-instance kernel
        auto q2m_host = [](const double* q, double* R) {
            const double w = q[0], x = q[1], y = q[2], z = q[3];
            R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
            R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
            R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
        };
        for (int b = 0; b < m->nb; ++b) {
            double R[9];
            const int jb = h.joint_of_body[b];
            if (jb >= 0) {
                q2m_host(h.quat[b], R);
                for (int r = 0; r < 3; ++r) h.jpos_par[jb][r] = R[r * 3] * h.jpos[jb][0] + R[r * 3 + 1] * h.jpos[jb][1] + R[r * 3 + 2] * h.jpos[jb][2];
            }
            for (int j = 0; j < m->nj; ++j) if ((h.anc_mask[b] >> j) & 1u) h.cmass[j] += h.mass[b];
            q2m_host(h.iquat[b], R);
            int e = 0;
            for (int r = 0; r < 3; ++r)
                for (int cc = r; cc < 3; ++cc)
                    h.icb[b][e++] = R[r * 3] * h.inertia[b][0] * R[cc * 3] + R[r * 3 + 1] * h.inertia[b][1] * R[cc * 3 + 1] + R[r * 3 + 2] * h.inertia[b][2] * R[cc * 3 + 2];
        }
    }
    for (int b = 0; b < m->nb; ++b) {      // the per-body records of the walk (FeModel::rec)
        double* w = h.rec[b];
        const int jb = h.joint_of_body[b];
        for (int i = 0; i < 3; ++i) { w[i] = h.pos[b][i]; w[7 + i] = h.ipos[b][i]; }
        for (int i = 0; i < 4; ++i) w[3 + i] = h.quat[b][i];
        for (int i = 0; i < 6; ++i) w[10 + i] = h.icb[b][i];
        w[16] = h.mass[b];
        for (int i = 0; i < 3; ++i) { w[17 + i] = jb >= 0 ? h.jaxis[jb][i] : 0.0; w[20 + i] = jb >= 0 ? h.jpos[jb][i] : 0.0; w[23 + i] = jb >= 0 ? h.jpos_par[jb][i] : 0.0; }
    }
    int row = 0;
    for (int d = 0; d < c->cfg.ndev; ++d) {
        if (m->ee_body[d] < 0 || m->ee_body[d] >= m->nb) return fail(c, IRLOSC_ERR_ARG, "ee_body[%d]=%d out of range", d, m->ee_body[d]);
        h.ee_body[d] = m->ee_body[d];
        h.row0[d] = row; row += c->cfg.dev_rows[d];
        for (int i = 0; i < 6; ++i) if (c->cfg.ctrlr_dof[d][i]) h.dofmask[d] |= 1u << i;
    }
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    if (!c->dmodel) HIPCHK(c, hipMalloc((void**)&c->dmodel, sizeof(FeModel)));
    c->fe_smem = frontend_smem_bytes(m->nb, m->nj);
    {
        const char* e = getenv("IRLOSC_FRONTEND");           // "generic": force the wave-per-instance kernel (A/B measurements)
        c->fe_lane = frontend_lane_dual_ur5_matches(h) && !(e && !strcmp(e, "generic"));
        const char* w = getenv("IRLOSC_WALK");               // "general": the shape-only walk on the fused path (A/B measurements, tests)
        c->fe_lane_s = c->fe_lane && frontend_lane_dual_ur5_s_matches(h) && !(w && !strcmp(w, "general"));
    }
    // (the lane kernel's side buffer -- 139 MB at 65 536 robots -- is allocated by the first irlosc_frontend: a context that only
    // ever takes the fused path never needs it)
    {   // The fused path needs the compiled tree shape (lane kernel) and the fp64 row16 kernel; IRLOSC_FUSED=0 forces the
        // two-kernel path through dense records (A/B measurements).
        const char* e = getenv("IRLOSC_FUSED");
        c->fused = c->fe_lane && c->kernel == IRLOSC_KERNEL_ROW16 && !(e && !strcmp(e, "0"));
        const char* ov = getenv("IRLOSC_FQ_OVERLAP");      // "0": consecutive fused trains on one stream (A/B measurements, tests)
        c->fq_overlap = !(ov && !strcmp(ov, "0"));
        { const char* nb = getenv("IRLOSC_FQ_BANKS"); const int v = nb ? atoi(nb) : 3; c->fq_xbanks = std::max(1, std::min(v, 1 + irlosc_ctx::MAX_XBANKS)) - 1; }
        const char* t = getenv("IRLOSC_FUSED_TRAIN");      // steps per launch pair of the fused path (A/B measurements)
        if (t && atoi(t) >= 1 && atoi(t) <= R16_TRAIN) c->fused_train = atoi(t);
    }
    if (c->fused) {
        FeCompactTables t;
        memset(&t, 0, sizeof t);
        frontend_lane_dual_ur5_tables(h, &t);
        if (c->fe_xentries != t.n_entries)               // an earlier model's exchange buffers have another size
            for (int k2 = 0; k2 < R16_TRAIN; ++k2) if (c->fe_xside[k2]) { HIPCHK(c, hipFree(c->fe_xside[k2])); c->fe_xside[k2] = nullptr; }
        c->fe_xentries = t.n_entries;
        if (!c->dtables) HIPCHK(c, hipMalloc((void**)&c->dtables, sizeof t));
        HIPCHK(c, hipMemcpyAsync(c->dtables, &t, sizeof t, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));      // t lives on this stack frame
        // (the exchange buffers themselves are allocated by the first fused step: ensure_xside)
        // The OSC step behind the walk: lane-per-robot form when an instantiation holds this layout (IRLOSC_LANE=0: the row16 FROMQ
        // kernel, A/B measurements and tests); its record buffers are sized for the instantiation's entries and start over with a new layout
        const char* le = getenv("IRLOSC_LANE");
        lane::RowMap map;
        const int tier = (le && !strcmp(le, "0")) ? -1 : lane_plan(h, &map);
        if (tier != c->lane_tier || (tier >= 0 && memcmp(&map, &c->lane_map, sizeof map)))
            for (int k2 = 0; k2 < R16_TRAIN; ++k2) if (c->lane_rec[k2]) { HIPCHK(c, hipFree(c->lane_rec[k2])); c->lane_rec[k2] = nullptr; }
        c->lane_tier = tier;
        if (tier >= 0) c->lane_map = map;
    } else {
        c->lane_tier = -1;
    }
    HIPCHK(c, hipMemcpyAsync(c->dmodel, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->dqpos.empty()) {
        c->dqpos.assign(c->cfg.n_slots, nullptr);
        c->dqvel.assign(c->cfg.n_slots, nullptr);
        c->dqt.assign(c->cfg.n_slots, nullptr);
        c->has_q.assign(c->cfg.n_slots, 0);
        for (int s2 = 0; s2 < c->cfg.n_slots; ++s2) {
            HIPCHK(c, hipMalloc((void**)&c->dqpos[s2], (size_t)c->cfg.max_batch * c->cfg.n * sizeof(double)));
            HIPCHK(c, hipMalloc((void**)&c->dqvel[s2], (size_t)c->cfg.max_batch * c->cfg.n * sizeof(double)));
        }
    }
    return IRLOSC_OK;
}

extern "C" int irlosc_upload_q(irlosc_ctx* c, int32_t slot, int32_t B, const double* qpos, const double* qvel) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (!c->dmodel) return fail(c, IRLOSC_ERR_STATE, "irlosc_set_model has not been called");
    if (B == 0) { c->has_q[slot] = -1; return IRLOSC_OK; }
    if (!qpos || !qvel) return fail(c, IRLOSC_ERR_ARG, "qpos and qvel are required");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    const size_t bytes = (size_t)B * c->cfg.n * sizeof(double);
    HIPCHK(c, hipMemcpyAsync(c->dqpos[slot], qpos, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dqvel[slot], qvel, bytes, hipMemcpyHostToDevice, c->stream));
    // the fused walk reads its own layout of the same numbers ([wave][2 n][64 robots]: coalesced, hinge by hinge): one small kernel
    // behind the copies (10 us per 65 536 robots against 0.8 ms of PCIe for them)
    // (only the fused path reads this layout; its buffer is allocated by the slot's first upload while the path is on -- and, once it
    //  exists, refreshed by EVERY upload: a copy left stale while another model had the path switched off would be walked later)
    if (c->fused && !c->dqt[slot])
        HIPCHK(c, hipMalloc((void**)&c->dqt[slot], (((size_t)c->cfg.max_batch + 63) / 64) * 2 * c->cfg.n * 64 * sizeof(double)));
    if (c->dqt[slot]) HIPCHK(c, (hipError_t)launch_q_layout(c->dqpos[slot], c->dqvel[slot], c->dqt[slot], B, c->cfg.n, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->has_q[slot] = B;
    return IRLOSC_OK;
}

static int frontend_launch(irlosc_ctx* c, int slot, int B) {
    if (!c->dmodel) return fail(c, IRLOSC_ERR_STATE, "irlosc_set_model has not been called");
    if (!c->has_q[slot]) return fail(c, IRLOSC_ERR_STATE, "slot %d: irlosc_upload_q must precede irlosc_frontend", slot);
    if (B == 0) { c->uploaded[slot] = -1; return IRLOSC_OK; }
    if (B > c->has_q[slot]) return fail(c, IRLOSC_ERR_STATE, "slot %d holds joint coordinates of %d instances, front end asked for %d", slot, std::max(0, c->has_q[slot]), B);
    if (c->fe_lane && !c->fe_side) {
        const size_t waves = ((size_t)c->cfg.max_batch + 63) / 64;
        if (hipMalloc((void**)&c->fe_side, waves * frontend_lane_dual_ur5_side_doubles_per_wave() * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            c->fe_side = nullptr;
            c->fe_lane = 0;              // out of memory: the wave-per-robot kernel needs no side buffer
            c->fused = 0;
        }
    }
    int rc;
    if (c->cfg.dtype == IRLOSC_F64) {
        const FeOut<double> o{(double*)c->dM[slot], (double*)c->dJ[slot], (double*)c->ddq[slot], (double*)c->dbias[slot], (double*)c->dee[slot]};
        rc = c->fe_lane ? launch_frontend_lane_dual_ur5<double>(c->dmodel, c->dqpos[slot], c->dqvel[slot], o, B, c->fe_side, c->stream)
                        : launch_frontend_generic<double>(c->dmodel, c->dqpos[slot], c->dqvel[slot], o, B, c->fe_smem, c->stream);
    } else {
        const FeOut<float> o{(float*)c->dM[slot], (float*)c->dJ[slot], (float*)c->ddq[slot], (float*)c->dbias[slot], (float*)c->dee[slot]};
        rc = c->fe_lane ? launch_frontend_lane_dual_ur5<float>(c->dmodel, c->dqpos[slot], c->dqvel[slot], o, B, c->fe_side, c->stream)
                        : launch_frontend_generic<float>(c->dmodel, c->dqpos[slot], c->dqvel[slot], o, B, c->fe_smem, c->stream);
    }
    HIPCHK(c, (hipError_t)rc);
    // The records of this slot are now those of B robots: an earlier, larger upload must not vouch for instances the front
    // end did not write (the wrench of the slot stays what the last irlosc_upload / irlosc_upload_raw put there).
    c->uploaded[slot] = B;
    if (!c->fused_away.empty()) c->fused_away[slot] = 0;
    c->tree_ok[slot] = c->fe_lane;     // the lane kernel walks the compiled tree: its records carry the tree's zeros by construction
    return IRLOSC_OK;
}

extern "C" int irlosc_frontend(irlosc_ctx* c, int32_t slot, int32_t B) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    return frontend_launch(c, slot, B);
}

extern "C" int irlosc_download_records(irlosc_ctx* c, int32_t slot, int32_t B, void* M, void* J, void* dq, void* bias,
                                       void* ee_pose) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (B > std::max(0, c->uploaded[slot])) return fail(c, IRLOSC_ERR_STATE, "slot %d holds state for %d instances", slot, std::max(0, c->uploaded[slot]));
    if (B == 0) return IRLOSC_OK;
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    const size_t b = (size_t)B, n = (size_t)c->cfg.n, k = (size_t)c->k, nd = (size_t)c->cfg.ndev, e = c->esz;
    if (M) HIPCHK(c, hipMemcpyAsync(M, c->dM[slot], b * n * n * e, hipMemcpyDeviceToHost, c->stream));
    if (J) HIPCHK(c, hipMemcpyAsync(J, c->dJ[slot], b * k * n * e, hipMemcpyDeviceToHost, c->stream));
    if (dq) HIPCHK(c, hipMemcpyAsync(dq, c->ddq[slot], b * n * e, hipMemcpyDeviceToHost, c->stream));
    if (bias) HIPCHK(c, hipMemcpyAsync(bias, c->dbias[slot], b * n * e, hipMemcpyDeviceToHost, c->stream));
    if (ee_pose) HIPCHK(c, hipMemcpyAsync(ee_pose, c->dee[slot], b * nd * 7 * e, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return IRLOSC_OK;
}

// A step from joint coordinates over B robots needs B robots of (qpos, qvel) AND of targets in the slot.
static int check_slot_q(irlosc_ctx* c, int slot, int B) {
    if (!c->dmodel) return fail(c, IRLOSC_ERR_STATE, "irlosc_set_model has not been called");
    if (!c->has_q[slot] || !c->targeted[slot])
        return fail(c, IRLOSC_ERR_STATE, "slot %d: irlosc_upload_q and irlosc_set_targets must precede a step from joint coordinates", slot);
    if (B > std::max(0, c->has_q[slot]) || B > std::max(0, c->targeted[slot]))
        return fail(c, IRLOSC_ERR_STATE, "slot %d holds joint coordinates of %d and targets for %d instances, step asked for %d", slot,
                    std::max(0, c->has_q[slot]), std::max(0, c->targeted[slot]), B);
    return IRLOSC_OK;
}

// Exchange buffers of the fused path: allocated by the first fused step, and only as many as the longest train so far needs
// (a caller of irlosc_step_from_q uses one: 334 entries x 512 B per 64 robots = 175 MB at 65 536 robots; the benchmark form all R16_TRAIN: 1.4 GB) -- a context
// that only ever runs irlosc_frontend + irlosc_step pays nothing.  Out of memory: the fused path is switched off for this
// context and the caller continues through dense records (-> 1).
static int ensure_xside(irlosc_ctx* c, int n) {
    const size_t waves = ((size_t)c->cfg.max_batch + 63) / 64;
    for (int k2 = 0; k2 < n && k2 < R16_TRAIN; ++k2) {
        if (c->fe_xside[k2]) continue;
        if (hipMalloc((void**)&c->fe_xside[k2], waves * c->fe_xentries * 64 * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            for (int k3 = 0; k3 < R16_TRAIN; ++k3) if (c->fe_xside[k3]) { (void)hipFree(c->fe_xside[k3]); c->fe_xside[k3] = nullptr; }
            c->fused = 0;
            return 1;
        }
    }
    // lane form of the OSC step: one record per robot and step for the eigen pass (all of a batch may be flagged), in whole groups of
    // 64 (transposed records; every entry of a record is written by the lane that owns it).  Out of memory here only switches the lane form off.
    if (c->lane_tier >= 0) {
        if (!c->dlane_count && hipMalloc((void**)&c->dlane_count, R16_TRAIN * sizeof(int32_t)) != hipSuccess) { (void)hipGetLastError(); c->lane_tier = -1; }
        for (int k2 = 0; c->lane_tier >= 0 && k2 < n && k2 < R16_TRAIN; ++k2) {
            if (c->lane_rec[k2]) continue;
            const size_t bytes = (size_t)((c->cfg.max_batch + 63) / 64 * 64) * lane::REC_DOUBLES * sizeof(double);      // (whole groups of 64 records)
            if (hipMalloc((void**)&c->lane_rec[k2], bytes) != hipSuccess || hipMemsetAsync(c->lane_rec[k2], 0, bytes, c->stream) != hipSuccess) {
                (void)hipGetLastError();
                for (int k3 = 0; k3 < R16_TRAIN; ++k3) if (c->lane_rec[k3]) { (void)hipFree(c->lane_rec[k3]); c->lane_rec[k3] = nullptr; }
                c->lane_tier = -1;
            }
        }
    }
    return 0;
}

// Fused path: one train of n steps from joint coordinates (step i: slot slots[i], outputs of set i).  Three launches -- the
// lane-per-robot walk leaves the structural non-zeros of M / J, the bias forces and the EE poses in the compact exchange
// buffer of each step; the task pass (one lane per (robot, device)) adds the k gained task-error rows; the row16 kernel (FROMQ)
// stages a block's lines in LDS and gathers its operands from there -- and, fourth, the give-up pass: the few robots the
// eigen stage hands over get their dense records from the wave-per-robot front end (worklist form) and go through the
// generic kernel like on the record path.  Dense M / J exist in HBM for those robots only.
template <typename T>
static int fused_train(irlosc_ctx* c, const int* slots, int n, int B, const irlosc_ctx::Bank& bk) {
    if (n < 1 || n > R16_TRAIN) return fail(c, IRLOSC_ERR_STATE, "train of %d steps", n);
    const hipStream_t st = bk.st;
    FeLaneTrain ft;
    memset(&ft, 0, sizeof ft);
    Row16Train<T> tr;
    memset(&tr, 0, sizeof tr);
    FeGenericArgs<T> ga;
    memset(&ga, 0, sizeof ga);
    HIPCHK(c, hipMemsetAsync(bk.count, 0, R16_TRAIN * sizeof(int32_t), st));
    ft.B = ga.B = B;
    for (int i = 0; i < n; ++i) {
        const int sl = slots[i];
        ft.qpos[i] = ga.qpos[i] = c->dqpos[sl];
        ft.qvel[i] = ga.qvel[i] = c->dqvel[sl];
        if (!c->dqt[sl]) {      // coordinates uploaded before the model made the fused path available: lay them out now
            HIPCHK(c, hipMalloc((void**)&c->dqt[sl], (((size_t)c->cfg.max_batch + 63) / 64) * 2 * c->cfg.n * 64 * sizeof(double)));
            HIPCHK(c, (hipError_t)launch_q_layout(c->dqpos[sl], c->dqvel[sl], c->dqt[sl], std::max(1, c->has_q[sl]), c->cfg.n, st));
        }
        ft.qt[i] = c->dqt[sl];
        ft.side[i] = bk.xside[i];
        fill_params<T>(c, tr.p[i], B, c->dM[sl], c->dJ[sl], c->ddq[sl], c->dbias[sl], c->dee[sl], c->dtgt[sl],
                       c->has_tvel[sl] ? c->dtvel[sl] : nullptr, c->has_wrench[sl] ? c->dwrench[sl] : nullptr, bk.u[i], bk.flags[i]);
        tr.x[i] = Row16Extra{c->dzeros, bk.list[i], bk.count + i, bk.xside[i], c->dqvel[sl], c->dtables, c->span_next};
        ga.out[i] = FeOut<T>{(T*)c->dM[sl], (T*)c->dJ[sl], (T*)c->ddq[sl], (T*)c->dbias[sl], (T*)c->dee[sl]};
        ga.list[i] = bk.list[i];
        ga.count[i] = bk.count + i;
    }
    // lane form of the OSC step: not with target velocities (branch B of osc.py:173-177 reads dx between the two halves of the task
    // signal: the row16 FROMQ kernel keeps those trains)
    bool use_lane = c->lane_tier >= 0;
    lane::LaneTrain lt;
    memset(&lt, 0, sizeof lt);
    for (int i = 0; i < n && use_lane; ++i) {
        if (c->has_tvel[slots[i]] || !bk.lane_rec[i] || !bk.lane_count) use_lane = false;
        lt.qt[i] = c->dqt[slots[i]];
        lt.rec[i] = bk.lane_rec[i];
        lt.rec_count[i] = bk.lane_count + i;
    }
    if (use_lane) {
        lt.map = c->lane_map;
        HIPCHK(c, hipMemsetAsync(bk.lane_count, 0, R16_TRAIN * sizeof(int32_t), st));
    }
    if (c->tev_begin) HIPCHK(c, hipEventRecord(c->tev_begin, st));
    HIPCHK(c, (hipError_t)(c->fe_lane_s ? launch_frontend_lane_compact_dual_ur5_s(c->dmodel, ft, n, st)
                                        : launch_frontend_lane_compact_dual_ur5(c->dmodel, ft, n, st)));
    if (use_lane) {
        if (!lane_task_in_kernel()) HIPCHK(c, (hipError_t)launch_row16_fromq<T>(tr, n, st, 1));      // the task pass (A/B builds: the lane kernel computes the rows itself)
        static const int eig_blocks = [] { const char* e = getenv("IRLOSC_LANE_EIG_BLOCKS"); const int v = e ? atoi(e) : 0; return v >= 64 && v <= 65536 ? v : 1024; }();
        // flagged robots of a step from which the eigen pass runs one lane per robot (below: four records per wave, row16 form); the
        // choice is made on the device, per step, from the count the lane kernel leaves (IRLOSC_LANE_EIG_MIN: A/B measurements)
        static const int lane_min = [] { const char* e = getenv("IRLOSC_LANE_EIG_MIN"); return e ? atoi(e) : 3000; }();
        HIPCHK(c, (hipError_t)launch_lane_osc<T>(tr, lt, n, c->lane_tier, eig_blocks, lane_min, st));
    } else {
        HIPCHK(c, (hipError_t)launch_row16_fromq<T>(tr, n, st));
    }
    HIPCHK(c, (hipError_t)launch_frontend_generic_lists<T>(c->dmodel, ga, n, c->fe_smem, st));
    HIPCHK(c, (hipError_t)launch_row16_worklist<T>(tr, n, nullptr, st));
    if (c->tev_end) HIPCHK(c, hipEventRecord(c->tev_end, st));
    // The give-up pass wrote dense records of the robots on its lists into the slots (and nothing for the others): what the
    // slots held before no longer belongs to one state.  They hold no records from here on -- irlosc_step / irlosc_step_resident
    // / irlosc_download_records on them fail with IRLOSC_ERR_STATE until irlosc_frontend / irlosc_upload* fills them again.
    if (c->fused_away.empty()) c->fused_away.assign(c->cfg.n_slots, 0);
    for (int i = 0; i < n; ++i) { c->uploaded[slots[i]] = 0; c->tree_ok[slots[i]] = 0; c->fused_away[slots[i]] = 1; }
    return IRLOSC_OK;
}

// bank 0 = the context's own buffers on its stream
static irlosc_ctx::Bank bank0_of(irlosc_ctx* c) {
    irlosc_ctx::Bank b;
    b.st = c->stream;
    for (int i = 0; i < R16_TRAIN; ++i) {
        b.xside[i] = c->fe_xside[i]; b.lane_rec[i] = c->lane_rec[i]; b.list[i] = c->dr16_list[i];
        b.u[i] = c->du_set[i]; b.flags[i] = c->dflags_set[i];
    }
    b.lane_count = c->dlane_count;
    b.count = c->dr16_count;
    return b;
}

// A further bank (see irlosc_ctx::Bank): same sizes as the context's own.  -> 0, or 1: not available (out of memory: fewer banks)
static int ensure_xbank(irlosc_ctx* c, int which, int n, bool fused) {
    irlosc_ctx::Bank& b = c->xb[which];
    const size_t Bm = (size_t)c->cfg.max_batch, waves = (Bm + 63) / 64;
    auto get = [](void** p, size_t bytes) { return *p || hipMalloc(p, bytes) == hipSuccess; };
    bool ok = true;
    if (!b.st) ok = ok && hipStreamCreateWithFlags(&b.st, hipStreamNonBlocking) == hipSuccess;
    if (!b.done) ok = ok && hipEventCreateWithFlags(&b.done, hipEventDisableTiming) == hipSuccess;
    if (!c->ev_join) ok = ok && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
    ok = ok && get((void**)&b.count, R16_TRAIN * sizeof(int32_t)) && get((void**)&b.lane_count, R16_TRAIN * sizeof(int32_t));
    for (int i = 0; ok && i < n && i < R16_TRAIN; ++i) {
        ok = ok && get((void**)&b.list[i], Bm * sizeof(int32_t)) && get(&b.u[i], Bm * c->cfg.n * c->esz) && get((void**)&b.flags[i], Bm * sizeof(uint32_t));
        if (!fused) { if (ok && c->task_pass) ok = get((void**)&b.trows[i], Bm * 16 * sizeof(double)); continue; }
        ok = ok && get((void**)&b.xside[i], waves * c->fe_xentries * 64 * sizeof(double));
        if (ok && c->lane_tier >= 0 && !b.lane_rec[i]) {
            const size_t bytes = (Bm + 63) / 64 * 64 * lane::REC_DOUBLES * sizeof(double);
            ok = hipMalloc((void**)&b.lane_rec[i], bytes) == hipSuccess && hipMemsetAsync(b.lane_rec[i], 0, bytes, c->stream) == hipSuccess;
            if (!ok) b.lane_rec[i] = nullptr;
        }
    }
    if (!ok) { (void)hipGetLastError(); return 1; }
    return 0;
}

static int fused_resident(irlosc_ctx* c, int first_slot, int B, int iters) {
    int done = 0, t = 0;
    const irlosc_ctx::Bank b0 = bank0_of(c);
    // more than one train: alternate banks / streams so that a train's first waves fill the tails of the one before (irlosc_ctx::Bank)
    int nx = 0;                                                        // banks beside the context's own that this call rotates over
    if (c->fq_overlap && iters >= c->fused_train && c->fused_train > 1 && !c->tev_begin) {
        const int want = std::min(c->fq_xbanks, (iters + c->fused_train - 1) / c->fused_train - 1);
        while (nx < want && ensure_xbank(c, nx, c->fused_train, true) == 0) ++nx;
    }
    if (nx) HIPCHK(c, hipEventRecord(c->ev_join, c->stream));          // the other banks' streams start behind whatever the main stream holds
    for (int k = 0; k < nx; ++k) HIPCHK(c, hipStreamWaitEvent(c->xb[k].st, c->ev_join, 0));
    const irlosc_ctx::Bank* last = &b0;
    while (done < iters) {
        const int n = std::min(c->fused_train, iters - done);
        int slots[R16_TRAIN];
        for (int i = 0; i < n; ++i) {
            slots[i] = (first_slot + done + i) % c->cfg.n_slots;
            int rc = check_slot_q(c, slots[i], B);
            if (rc) return rc;
        }
        const int which = t % (nx + 1);
        const irlosc_ctx::Bank& bk = which ? c->xb[which - 1] : b0;
        int rc = c->cfg.dtype == IRLOSC_F64 ? fused_train<double>(c, slots, n, B, bk) : fused_train<float>(c, slots, n, B, bk);
        if (rc) return rc;
        last = &bk;
        c->cur = n - 1;
        done += n;
        ++t;
    }
    for (int k = 0; k < nx; ++k) {                                     // and the main stream continues behind all of them
        HIPCHK(c, hipEventRecord(c->xb[k].done, c->xb[k].st));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->xb[k].done, 0));
    }
    c->du = last->u[c->cur];
    c->dflags = last->flags[c->cur];
    c->count_cur = last->count;
    return IRLOSC_OK;
}

extern "C" const char* irlosc_from_q_name(const irlosc_ctx* c) {
    static thread_local std::string nm;
    if (!c || !c->dmodel) return "";
    if (c->fused && c->lane_tier >= 0) {
        char sh[64];
        snprintf(sh, sizeof sh, "osc_lane_%s_rows_%d_%d_%d + eigen pass", c->cfg.dtype == IRLOSC_F64 ? "f64" : "f32in_f64", lane::TIER_ROWS[c->lane_tier][0],
                 lane::TIER_ROWS[c->lane_tier][1], lane::TIER_ROWS[c->lane_tier][2]);
        nm = std::string(c->fe_lane_s ? "osc_frontend_lane_compact_dual_ur5_s + " : "osc_frontend_lane_compact_dual_ur5 + ") + sh +
             " (fused: compact exchange buffer, no dense M / J; OSC step one lane per robot; target velocities: " + c->kernel_name + "_fromq)";
    } else if (c->fused) nm = std::string(c->fe_lane_s ? "osc_frontend_lane_compact_dual_ur5_s + " : "osc_frontend_lane_compact_dual_ur5 + ") + c->kernel_name + "_fromq (fused: compact exchange buffer, no dense M / J)";
    else nm = std::string(irlosc_frontend_name(c)) + " + " + c->kernel_name + " (through dense records)";
    return nm.c_str();
}

extern "C" int irlosc_step_from_q(irlosc_ctx* c, int32_t slot, int32_t B, void* u_host, uint32_t* flags_host) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, slot, B);
    if (rc) return rc;
    if (c->gains_nb == 0) return fail(c, IRLOSC_ERR_STATE, "irlosc_set_gains has not been called");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    if (B > 0) {
        if (c->fused && ensure_xside(c, 1) == 0) {
            rc = fused_resident(c, slot, B, 1);
        } else {
            rc = check_slot_q(c, slot, B);
            if (!rc) rc = frontend_launch(c, slot, B);
            if (!rc) rc = launch_slot(c, slot, B);
        }
        if (rc) return rc;
    }
    if (u_host || flags_host) return irlosc_download(c, B, u_host, flags_host);
    return IRLOSC_OK;
}

extern "C" int irlosc_step_resident_from_q(irlosc_ctx* c, int32_t first_slot, int32_t B, int32_t iters, float* ms_total,
                                           float* ms_step_avg) {
    if (!c) return IRLOSC_ERR_ARG;
    int rc = check_slot(c, first_slot, B);
    if (rc) return rc;
    if (iters < 1) return fail(c, IRLOSC_ERR_ARG, "iters must be >= 1");
    if (c->gains_nb == 0) return fail(c, IRLOSC_ERR_STATE, "irlosc_set_gains has not been called");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (c->fused && B > 0 && ensure_xside(c, std::min(c->fused_train, iters)) == 0) {
        rc = fused_resident(c, first_slot, B, iters);
        if (rc) return rc;
    } else {
        for (int i = 0; i < iters; ++i) {
            const int slot = (first_slot + i) % c->cfg.n_slots;
            rc = check_slot_q(c, slot, B);
            if (!rc) rc = frontend_launch(c, slot, B);
            if (rc) return rc;
            rc = launch_slot(c, slot, B);
            if (rc) return rc;
        }
    }
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (ms_total) *ms_total = ms;
    if (ms_step_avg) *ms_step_avg = ms / (float)iters;
    return IRLOSC_OK;
}

extern "C" int irlosc_device_sync(irlosc_ctx* c) {
    if (!c) return IRLOSC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    HIPCHK(c, hipDeviceSynchronize());
    return IRLOSC_OK;
}

extern "C" int irlosc_tick(irlosc_ctx* c, int32_t B, const void* M, const void* J, const void* dq, const void* bias,
                           const void* ee_pose, const void* wrench, const void* tgt_pose, const void* tgt_vel, void* u_host,
                           uint32_t* flags_host) {
    if (!c) return IRLOSC_ERR_ARG;
    if (B < 0 || B > c->cfg.max_batch) return fail(c, IRLOSC_ERR_ARG, "B=%d out of [0,%d]", B, c->cfg.max_batch);
    if (B == 0) return IRLOSC_OK;
    if (!M || !J || !dq || !ee_pose || !tgt_pose || !u_host) return fail(c, IRLOSC_ERR_ARG, "M, J, dq, ee_pose, tgt_pose and u_host are required");
    if ((c->cfg.flags & IRLOSC_USE_G) && !bias) return fail(c, IRLOSC_ERR_ARG, "bias required with IRLOSC_USE_G");
    if (c->gains_nb == 0) return fail(c, IRLOSC_ERR_STATE, "irlosc_set_gains has not been called");
    HIPCHK(c, hipSetDevice(c->cfg.hip_device));
    const size_t b = (size_t)B, n = (size_t)c->cfg.n, k = (size_t)c->k, nd = (size_t)c->cfg.ndev, e = c->esz;
    // input block: M | J | dq | bias | ee | tgt | wrench | tvel, each piece 256-byte aligned
    const void* src[8] = {M, J, dq, bias, ee_pose, tgt_pose, wrench, tgt_vel};
    const size_t sz[8] = {b * n * n * e, b * k * n * e, b * n * e, bias ? b * n * e : 0, b * nd * 7 * e, b * nd * 7 * e,
                          wrench ? b * nd * 6 * e : 0, tgt_vel ? b * nd * 6 * e : 0};
    size_t off[8], total = 0;
    for (int i = 0; i < 8; ++i) { off[i] = total; total += (sz[i] + 255) & ~(size_t)255; }
    if (total > c->tick_in_bytes) {
        if (c->tick_hin) HIPCHK(c, hipHostFree(c->tick_hin));
        if (c->tick_din) HIPCHK(c, hipFree(c->tick_din));
        c->tick_hin = c->tick_din = nullptr; c->tick_in_bytes = 0;
        HIPCHK(c, hipHostMalloc(&c->tick_hin, total, hipHostMallocDefault));
        HIPCHK(c, hipMalloc(&c->tick_din, total));
        c->tick_in_bytes = total;
    }
    // output block: u | flags | the two words of the symmetry probe (they ride back in the one copy the tick makes anyway)
    const size_t ub = b * n * e, fl_off = (ub + 255) & ~(size_t)255, sym_off = (fl_off + b * sizeof(uint32_t) + 15) & ~(size_t)15;
    const size_t out_total = sym_off + 2 * sizeof(int32_t);
    const bool sym_dev = sym_applies(c) && B > SYM_HOST_MAX_B;
    if (sym_applies(c) && !sym_dev) {
        int rch = symmetry_host(c, M, B);
        if (rch) return rch;
    }
    if (out_total > c->tick_out_bytes) {
        if (c->tick_hout) HIPCHK(c, hipHostFree(c->tick_hout));
        if (c->tick_dout) HIPCHK(c, hipFree(c->tick_dout));
        c->tick_hout = c->tick_dout = nullptr; c->tick_out_bytes = 0;
        HIPCHK(c, hipHostMalloc(&c->tick_hout, out_total, hipHostMallocDefault));
        HIPCHK(c, hipMalloc(&c->tick_dout, out_total));
        c->tick_out_bytes = out_total;
    }
    unsigned char* hin = (unsigned char*)c->tick_hin;
    unsigned char* din = (unsigned char*)c->tick_din;
    for (int i = 0; i < 8; ++i) if (sz[i]) memcpy(hin + off[i], src[i], sz[i]);
    HIPCHK(c, hipMemcpyAsync(din, hin, total, hipMemcpyHostToDevice, c->stream));
    unsigned char* dout = (unsigned char*)c->tick_dout;
    uint32_t* dfl = (uint32_t*)(dout + fl_off);
    if (sym_dev) {
        int rcs = symmetry_probe(c, din + off[0], B, (int32_t*)(dout + sym_off), c->stream);
        if (rcs) return rcs;
    }
    int rc = launch(c, B, din + off[0], din + off[1], din + off[2], sz[3] ? din + off[3] : nullptr, din + off[4], din + off[5],
                    sz[7] ? din + off[7] : nullptr, sz[6] ? din + off[6] : nullptr, dout, dfl, c->stream);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->tick_hout, dout, out_total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (sym_dev) {                                      // nothing is handed out for an asymmetric M
        int rcs = symmetry_verdict(c, (const int32_t*)((unsigned char*)c->tick_hout + sym_off));
        if (rcs) return rcs;
    }
    memcpy(u_host, c->tick_hout, ub);
    if (flags_host) memcpy(flags_host, (unsigned char*)c->tick_hout + fl_off, b * sizeof(uint32_t));
    return IRLOSC_OK;
}

// ---- multi-GPU throughput reduction over RCCL (dlopen: a single-GPU deployment never loads librccl) ---------------
namespace {
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
thread_local std::string g_comm_error;

int comm_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_comm_error = buf;
    return code;
}

int rccl_load() {
    if (g_rccl.h) return IRLOSC_OK;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return comm_fail(IRLOSC_ERR_HIP, "cannot load librccl.so: %s", dlerror());
    RcclApi a;
    a.h = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.AllGather || !a.GetErrorString)
        return comm_fail(IRLOSC_ERR_HIP, "librccl.so lacks an expected symbol");
    g_rccl = a;
    return IRLOSC_OK;
}
}  // namespace

struct irlosc_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double* dbuf = nullptr;          // 2 doubles in, 2 doubles out, then `world` uint64 for the all-gather
    std::string err;
};

#define COMMCHK(cm, expr)                                                                                     \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            int rc_ = comm_fail(IRLOSC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));               \
            if (cm) (cm)->err = g_comm_error;                                                                 \
            return rc_;                                                                                       \
        }                                                                                                     \
    } while (0)
#define NCCLCHK(cm, expr)                                                                                     \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) {                                                                              \
            int rc_ = comm_fail(IRLOSC_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));           \
            if (cm) (cm)->err = g_comm_error;                                                                 \
            return rc_;                                                                                       \
        }                                                                                                     \
    } while (0)

extern "C" const char* irlosc_comm_last_error(const irlosc_comm* cm) { return cm ? cm->err.c_str() : g_comm_error.c_str(); }

extern "C" int irlosc_comm_unique_id(uint8_t id_out[IRLOSC_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == IRLOSC_COMM_ID_BYTES, "unique id size");
    if (!id_out) return comm_fail(IRLOSC_ERR_ARG, "id_out is NULL");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK((irlosc_comm*)nullptr, g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return IRLOSC_OK;
}

extern "C" int irlosc_comm_create(int32_t hip_device, int32_t rank, int32_t world, const uint8_t id[IRLOSC_COMM_ID_BYTES],
                                  irlosc_comm** out) {
    if (!out) return comm_fail(IRLOSC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!id) return comm_fail(IRLOSC_ERR_ARG, "id is NULL");
    if (world < 1 || rank < 0 || rank >= world) return comm_fail(IRLOSC_ERR_ARG, "rank %d outside world of %d", rank, world);
    int ndevs = 0;
    if (hipGetDeviceCount(&ndevs) != hipSuccess || ndevs < 1) return comm_fail(IRLOSC_ERR_HIP, "no HIP device available");
    if (hip_device < 0 || hip_device >= ndevs) return comm_fail(IRLOSC_ERR_ARG, "hip_device=%d but %d device(s) visible", hip_device, ndevs);
    int rc = rccl_load();
    if (rc) return rc;
    irlosc_comm* cm = new (std::nothrow) irlosc_comm();
    if (!cm) return comm_fail(IRLOSC_ERR_HIP, "out of host memory");
    cm->device = hip_device; cm->rank = rank; cm->world = world;
    auto body = [&]() -> int {
        COMMCHK(cm, hipSetDevice(hip_device));
        COMMCHK(cm, hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking));
        COMMCHK(cm, hipMalloc((void**)&cm->dbuf, (4 + (size_t)world + 1) * sizeof(double)));
        ncclUniqueId uid;
        memcpy(&uid, id, sizeof uid);
        NCCLCHK(cm, g_rccl.CommInitRank(&cm->comm, world, uid, rank));
        return IRLOSC_OK;
    };
    rc = body();
    if (rc) { irlosc_comm_destroy(cm); return rc; }
    *out = cm;
    return IRLOSC_OK;
}

extern "C" void irlosc_comm_destroy(irlosc_comm* cm) {
    if (!cm) return;
    (void)hipSetDevice(cm->device);
    if (cm->stream) (void)hipStreamSynchronize(cm->stream);
    if (cm->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(cm->comm);
    if (cm->dbuf) (void)hipFree(cm->dbuf);
    if (cm->stream) (void)hipStreamDestroy(cm->stream);
    delete cm;
}

extern "C" int irlosc_bench_allreduce(irlosc_comm* cm, double* steps_sum, double* elapsed_max) {
    if (!cm || !steps_sum || !elapsed_max) return comm_fail(IRLOSC_ERR_ARG, "NULL argument");
    COMMCHK(cm, hipSetDevice(cm->device));
    const double in[2] = {*steps_sum, *elapsed_max};
    COMMCHK(cm, hipMemcpyAsync(cm->dbuf, in, sizeof in, hipMemcpyHostToDevice, cm->stream));
    NCCLCHK(cm, g_rccl.AllReduce(cm->dbuf, cm->dbuf + 2, 1, ncclFloat64, ncclSum, cm->comm, cm->stream));
    NCCLCHK(cm, g_rccl.AllReduce(cm->dbuf + 1, cm->dbuf + 3, 1, ncclFloat64, ncclMax, cm->comm, cm->stream));
    double outv[2];
    COMMCHK(cm, hipMemcpyAsync(outv, cm->dbuf + 2, sizeof outv, hipMemcpyDeviceToHost, cm->stream));
    COMMCHK(cm, hipStreamSynchronize(cm->stream));
    *steps_sum = outv[0];
    *elapsed_max = outv[1];
    return IRLOSC_OK;
}

extern "C" int irlosc_comm_allgather_u64(irlosc_comm* cm, uint64_t mine, uint64_t* all) {
    if (!cm || !all) return comm_fail(IRLOSC_ERR_ARG, "NULL argument");
    COMMCHK(cm, hipSetDevice(cm->device));
    uint64_t* d = (uint64_t*)(cm->dbuf + 4);
    COMMCHK(cm, hipMemcpyAsync(d + cm->world, &mine, sizeof mine, hipMemcpyHostToDevice, cm->stream));
    NCCLCHK(cm, g_rccl.AllGather(d + cm->world, d, 1, ncclUint64, cm->comm, cm->stream));
    COMMCHK(cm, hipMemcpyAsync(all, d, (size_t)cm->world * sizeof(uint64_t), hipMemcpyDeviceToHost, cm->stream));
    COMMCHK(cm, hipStreamSynchronize(cm->stream));
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
