// Stage 1 of the group path: G lanes per robot instance (G = 4 or 8), n = 25, fp32.
// See the header comment of osc_group.hpp for the design; this file is the kernel itself.
//
// Lane (q, g) = (lane / G, lane % G) works on instance `tile * (64 / G) + q` and owns the joint rows
// i = G * s + g.  G = 4: 7 slots (6 paired + scalar slot 6 = row 24, real only for g == 0),
// 16 instances per wave, ~33 KB LDS, one wave per SIMD.  G = 8: 4 slots = 2 pairs (slot 3 = rows
// 24..31, real only for g == 0), 8 instances per wave, ~17 KB LDS and < 256 registers, so TWO waves
// share a SIMD: a lone wave issues at most one VALU op per quad-cycle, two interleave.
#pragma once
#include "osc_common.hpp"

namespace irlosc {

struct S2Args;
template <int K> __device__ __forceinline__ void stage2_body(const S2Args a, int blk, int32_t* list);

namespace grp {

template <int G> struct Geo {
    static constexpr int TILE = 64 / G;               // instances per wave
    static constexpr int NS = (N + G - 1) / G;        // row slots per lane
    static constexpr int P = NS / 2;                  // slot pairs (float2)
    static constexpr bool ODD = (NS & 1) != 0;        // a scalar last slot
    static constexpr int LS = NS - 1;                 // the slot that holds row 24 (= G * LS)
    static constexpr int PSTR = (G == 4) ? 25 : 26;   // 16-byte pieces per instance in a 4-row chunk image
    static constexpr int STR4 = PSTR * 4;             // float stride between instances (100 / 104)
    static constexpr int CI = TILE / 2;                  // DMA instructions per 4-row chunk: two instances each
    static constexpr int C1 = TILE / 2;                  // DMA instructions per 1-row chunk
    static constexpr int SLOT = TILE * STR4;             // floats per ring slot
    static_assert(G * LS == 24, "row 24 must be lane 0 of the last slot");
    static_assert(CI == C1, "vmcnt bookkeeping assumes equal instruction counts for both chunk kinds");
};

// broadcast lane gl (0..G-1, compile-time after unrolling) of every G-lane group to the whole group
template <int G> __device__ __forceinline__ float gbcast(float v, int gl) {
    float t = qbcast(v, gl & 3);
    if (G == 8) {
        // the value now sits in all 4 lanes of the owner's quad of every octet; copy it to the other quad:
        // row_shr:4 (lane i <- lane i-4) into the upper quads (banks 1,3), or row_shl:4 into the lower (0,2)
        const int ti = __builtin_bit_cast(int, t);
        if ((gl & 4) == 0) t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ti, ti, 0x114, 0xf, 0xA, false));
        else t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ti, ti, 0x104, 0xf, 0x5, false));
    }
    return t;
}
template <int G> __device__ __forceinline__ int gbcast_i(int v, int gl) {
    int t = qbcast_i(v, gl & 3);
    if (G == 8) {
        if ((gl & 4) == 0) t = __builtin_amdgcn_update_dpp(t, t, 0x114, 0xf, 0xA, false);
        else t = __builtin_amdgcn_update_dpp(t, t, 0x104, 0xf, 0x5, false);
    }
    return t;
}
// sum over the G lanes of a group (result in every lane)
template <int G> __device__ __forceinline__ float gsum(float v) {
    v = qsum(v);
    if (G == 8) v += dpp_f<0x141>(v);     // row_half_mirror: the two quads of an octet swap
    return v;
}
template <int G> __device__ __forceinline__ uint32_t gor(uint32_t v) {
    v = qor(v);
    if (G == 8) v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);
    return v;
}

// ---- LDS-DMA (global_load_lds), hand-counted: the compiler neither counts nor waits for these loads --------------
// Instruction issue is what bounds this kernel, so a chunk is ONE asm block with as little scalar bookkeeping as
// possible.  LDS destination of a DMA instruction = M0 + immediate offset + lane * size; the immediate offset is also
// added to the global address.  Every instruction of a chunk moves TWO instances (2 x 25 pieces = 50 lanes): its LDS
// image advances by 50 pieces (immediate offset), its source by two instance strides (one VALU add on the lane
// offset), so a single lane-dependent register addresses the whole chunk.  EXEC is narrowed to the 50 lanes inside
// the block only.
// wait until at most `younger` chunks (CI DMA instructions each) are still in flight; `younger` folds to a
// constant after unrolling
template <int CI> __device__ __forceinline__ void wait_chunks(int younger) {
    switch (younger) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<CI>(); break;
        default: wait_vm<2 * CI>(); break;
    }
}
// lane offset of piece `lane` in a two-instance group: instance lane / 25, piece lane % 25 (lanes >= 50 are masked)
template <int PIECE_BYTES> __device__ __forceinline__ uint32_t pair_offset(int lane, int stride_bytes) {
    return (uint32_t)(lane * PIECE_BYTES + (lane >= 25 ? stride_bytes - 25 * PIECE_BYTES : 0));
}
// 4-row chunk: 16 instances x 25 pieces of 16 B -> 8 instructions.  STRIDE = bytes between instances.
template <int STRIDE> __device__ __forceinline__ void dma4rows(const float* src, float* buf, int lane) {
    uint32_t t = pair_offset<16>(lane, STRIDE), km;
    unsigned long long ke;
    constexpr int D = 2 * STRIDE - 800;           // source advance per instruction, net of the immediate offset
    asm volatile(
        "s_mov_b32 %[km], m0\n\ts_mov_b64 %[ke], exec\n\ts_mov_b32 m0, %[lds]\n\t"
        "s_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, 0x3ffff\n\t"
        "global_load_lds_dwordx4 %[t], %[b]\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dwordx4 %[t], %[b] offset:800\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dwordx4 %[t], %[b] offset:1600\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dwordx4 %[t], %[b] offset:2400\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dwordx4 %[t], %[b] offset:3200\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dwordx4 %[t], %[b] offset:4000\n\tv_add_u32 %[t], %[d6], %[t]\n\t"
        "s_add_u32 m0, m0, 0x12c0\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %[t], %[b]\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dwordx4 %[t], %[b] offset:800\n\t"
        "s_mov_b64 exec, %[ke]\n\ts_mov_b32 m0, %[km]"
        : [t] "+v"(t), [km] "=&s"(km), [ke] "=&s"(ke)
        : [b] "s"(src), [lds] "s"(lds_addr(buf)), [d] "i"(D), [d6] "i"(D + 4800)
        : "memory", "scc");
}
// 1-row chunk: 16 instances x 25 dwords -> 8 instructions of 4 B per lane
template <int STRIDE> __device__ __forceinline__ void dma1row(const float* src, float* buf, int lane) {
    uint32_t t = pair_offset<4>(lane, STRIDE), km;
    unsigned long long ke;
    constexpr int D = 2 * STRIDE - 200;
    asm volatile(
        "s_mov_b32 %[km], m0\n\ts_mov_b64 %[ke], exec\n\ts_mov_b32 m0, %[lds]\n\t"
        "s_mov_b32 exec_lo, -1\n\ts_mov_b32 exec_hi, 0x3ffff\n\t"
        "global_load_lds_dword %[t], %[b]\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dword %[t], %[b] offset:200\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dword %[t], %[b] offset:400\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dword %[t], %[b] offset:600\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dword %[t], %[b] offset:800\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dword %[t], %[b] offset:1000\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dword %[t], %[b] offset:1200\n\tv_add_u32 %[t], %[d], %[t]\n\t"
        "global_load_lds_dword %[t], %[b] offset:1400\n\t"
        "s_mov_b64 exec, %[ke]\n\ts_mov_b32 m0, %[km]"
        : [t] "+v"(t), [km] "=&s"(km), [ke] "=&s"(ke)
        : [b] "s"(src), [lds] "s"(lds_addr(buf)), [d] "i"(D)
        : "memory", "scc");
}
// contiguous block of PIECES <= 128 16-byte pieces: (PIECES + 63) / 64 instructions, the last one partial
template <int PIECES> __device__ __forceinline__ void dmalinear(const float* src, float* buf, int lane) {
    static_assert(PIECES >= 1 && PIECES <= 128, "one or two instructions");
    const uint32_t t = (uint32_t)lane * 16u;
    uint32_t km;
    unsigned long long ke;
    constexpr int TAIL = PIECES > 64 ? PIECES - 64 : PIECES;                      // active lanes of the last instruction
    constexpr unsigned long long MASK = TAIL >= 64 ? ~0ull : ((1ull << (TAIL & 63)) - 1ull);
    if constexpr (PIECES > 64) {
        asm volatile(
            "s_mov_b32 %[km], m0\n\ts_mov_b64 %[ke], exec\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[t], %[b]\n\t"
            "s_mov_b32 exec_lo, %[mlo]\n\ts_mov_b32 exec_hi, %[mhi]\n\t"
            "global_load_lds_dwordx4 %[t], %[b] offset:1024\n\t"
            "s_mov_b64 exec, %[ke]\n\ts_mov_b32 m0, %[km]"
            : [km] "=&s"(km), [ke] "=&s"(ke)
            : [t] "v"(t), [b] "s"(src), [lds] "s"(lds_addr(buf)), [mlo] "i"((int)(MASK & 0xffffffffull)), [mhi] "i"((int)(MASK >> 32))
            : "memory", "scc");
    } else {
        asm volatile(
            "s_mov_b32 %[km], m0\n\ts_mov_b64 %[ke], exec\n\ts_mov_b32 m0, %[lds]\n\t"
            "s_mov_b32 exec_lo, %[mlo]\n\ts_mov_b32 exec_hi, %[mhi]\n\t"
            "global_load_lds_dwordx4 %[t], %[b]\n\t"
            "s_mov_b64 exec, %[ke]\n\ts_mov_b32 m0, %[km]"
            : [km] "=&s"(km), [ke] "=&s"(ke)
            : [t] "v"(t), [b] "s"(src), [lds] "s"(lds_addr(buf)), [mlo] "i"((int)(MASK & 0xffffffffull)), [mhi] "i"((int)(MASK >> 32))
            : "memory", "scc");
    }
}

}  // namespace grp

// One launch = a TRAIN of steps: table[s] describes step s (its resident inputs, its output set, the step whose
// stage 2 rides in front of its tiles, and the first block of its range).  Steps of a train are independent
// (different input slots, different output sets) and their riders belong to the PREVIOUS train, so nothing in a
// launch waits on anything else in it; a train of one step is the plain single-step launch.  Chaining steps in one
// grid removes the ramp-up / drain / launch gap between them (~17 us per launch against ~35 us per wave round).
// The table is written by the host before the launch and never by the kernel: it is read through the constant
// address space, i.e. with scalar loads at the point of use, exactly like kernel arguments.
struct TrainStep {
    KParams<float> p;
    S2Args prev;              // nfast == 0: nothing rides in this step
    KParams<float> prev_p;    // full parameters of that previous step (its give-up list -> generic kernel)
    float* side;              // [side_cap][NA + K] hand-off records of this step's output set
    int32_t* giveup_count;    // length of this step's give-up list (zeroed here, filled by its stage 2 later)
    int side_cap;
    int n2;                   // rider blocks in front of the tiles
    int block0;               // first block of this step's range
    int pad;
};

template <int G, int K, int NDEV, int NB>
__global__ __launch_bounds__(64, 2) void osc_group_kernel_f32(const TrainStep* __restrict__ table, const int nsteps) {
    typedef const __attribute__((address_space(4))) TrainStep* ctab_t;
    const ctab_t ct = (ctab_t)table;
    int step = 0;
    while (step + 1 < nsteps && (int)blockIdx.x >= ct[step + 1].block0) ++step;
    const __attribute__((address_space(4))) KParams<float>& p = ct[step].p;
    float* __restrict__ const side = ct[step].side;
    int32_t* __restrict__ const giveup_count = ct[step].giveup_count;
    const int n2 = ct[step].n2;
    const int blk = (int)blockIdx.x - ct[step].block0;
    using namespace grp;
    using GE = Geo<G>;
    constexpr int TILE = GE::TILE, P = GE::P, NS = GE::NS, LS = GE::LS, CI = GE::CI;
    constexpr bool ODD = GE::ODD;
    constexpr int NJ4 = K / 4;                       // J chunks of 4 rows ...
    constexpr int NJ1 = K % 4;                       // ... then the remaining rows one per chunk (k = 13: 1, k = 7: 3)
    constexpr bool LAST1 = NJ1 > 0;                  // the last J chunk is a single row
    constexpr int NCHM = 7;                          // M chunks: 6 x 4 rows + 1 row
    constexpr int NCHJ = NJ4 + NJ1;                  // J chunks
    static_assert(NJ4 >= 1 && NCHJ >= 2, "at least one 4-row chunk and two chunks of J");
    constexpr int NCH1 = NCHM + NCHJ;                // chunks of the first pass (M then J)
    constexpr int NA = K * (K + 1) / 2;
    constexpr int PDQ = TILE * N / 4, PEE = TILE * NDEV * 7 / 4;      // 16-byte pieces of the vector arrays
    constexpr int NVEC = (PDQ + 63) / 64 + 2 * ((PEE + 63) / 64);     // their DMA instruction count
    // LDS map (floats): two ring slots, then dq | ee | tgt | W (task vector exchange) | X (Mdq, dx parking).
    // Kept under 20 KB so that two waves fit per SIMD: a lone wave issues at most one VALU op per quad-cycle.
    constexpr int SLOT = GE::SLOT;
    constexpr int VEC_DQ = 0, VEC_EE = VEC_DQ + TILE * N, VEC_TGT = VEC_EE + TILE * NDEV * 7;
    constexpr int VEC_W = VEC_TGT + TILE * NDEV * 7, VEC_X = VEC_W + TILE * K, VEC_END = VEC_X + TILE * (N + K);
    static_assert(TILE * NA <= SLOT, "the A hand-off area must fit in one ring slot");
    constexpr int NT = NCH1;                         // ring chunks (one pass; the tail of J simply stays in the ring)
    // Second pass over J (u -= J^T t): rows 8..11 (chunk J2) are still in their ring slot and, for k = 13, row 12
    // (chunk J3) in the other one; only rows 0..7 are fetched again, straight into registers.
    // The last chunk of J always stays in its slot.  If it is a single row (k = 13, k = 7), the A hand-off area goes
    // BEHIND it in the same slot and the chunk before it survives in the other slot too; if it has four rows (k = 12)
    // the A area takes the other slot.  Rows [0, NDL) are the ones that have to be fetched again.
    static_assert(NB == 2, "resident-tail bookkeeping assumes two ring slots");
    constexpr int LAST_ROWS = LAST1 ? 1 : 4;
    constexpr int PREV_ROWS = NJ1 >= 2 ? 1 : 4;      // rows of the chunk before the last one
    constexpr bool PREV_RES = LAST1;                 // that chunk is still intact when the torque phase runs
    constexpr int SLOT_LAST = (NT - 1) % NB, SLOT_PREV = (NT - 2) % NB;
    constexpr int SLOT_A = LAST1 ? SLOT_LAST : SLOT_PREV;
    constexpr int A_OFF = LAST1 ? TILE * N : 0;      // A records start behind the single resident row
    constexpr int NDL = K - LAST_ROWS - (PREV_RES ? PREV_ROWS : 0);     // rows [0, NDL) by plain loads into registers
    constexpr int ROW_PREV = NDL, ROW_LAST = K - LAST_ROWS;             // first row of the two resident chunks
    constexpr int A_FIT = (SLOT - A_OFF) / NA < TILE ? (SLOT - A_OFF) / NA : TILE;   // instances whose record fits in the slot
    __shared__ __attribute__((aligned(16))) float ring[NB * SLOT];
    __shared__ __attribute__((aligned(16))) float vec[VEC_END];

    // Fused launch: the first n2 blocks run STAGE 2 OF THE PREVIOUS STEP (its own output set), the rest stage 1
    // of this step.  Stage 2 alone keeps one latency-bound wave per SIMD busy for ~28 us with the other slot
    // idle; inside this launch its waves share the SIMDs with stage-1 waves instead.
    static_assert(S2Lds<K>::WORDS <= NB * SLOT, "stage 2 borrows the ring as its LDS");
    if (blk < n2) {
        const S2Args prev = pod_copy<S2Args>(ct[step].prev);
        stage2_body<K>(prev, blk, reinterpret_cast<int32_t*>(ring));
        return;
    }
    const int lane = threadIdx.x;
    const int g = lane % G, q = lane / G;
    const int tile = blk - n2;
    const int b = tile * TILE + q;
    const size_t t0 = (size_t)tile * TILE;
    const bool has_tv = p.tvel != nullptr;
    const bool has_wr = (p.cfgflags & IRLOSC_ADMITTANCE) && p.wrench != nullptr;
    unsigned long long ts[8];
    const unsigned long long rt0 = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;   // 100 MHz wall clock
#define IRLOSC_TS(i) ts[i] = p.dbg ? __builtin_readcyclecounter() : 0ull
    IRLOSC_TS(0);
    if (tile == 0 && lane == 0) *giveup_count = 0;           // consumed by this step's stage 2, which runs later

    // Device record and gains of device g first: loads retire in order, so anything issued behind the streams
    // below would make the task-error code wait for megabytes it does not need.  (Unconditional, clamped index:
    // a select on the loaded values would make the wave wait for them right here.)
    auto gains_ptr = [&]() { return p.gains + (p.gains_per_instance ? (size_t)b * NDEV * IRLOSC_GAIN_WORDS : 0); };
    const int gd = g < NDEV ? g : 0;
    const DevMeta dm1 = pod_copy<DevMeta>(p.dev[gd]);
    float gl[IRLOSC_GAIN_WORDS];
    {
        const float* gg = gains_ptr() + gd * IRLOSC_GAIN_WORDS;
#pragma unroll
        for (int i = 0; i < IRLOSC_GAIN_WORDS; ++i) gl[i] = gg[i];
    }

    // ---------------- prologue --------------------------------------------------------------------------------
    // DMA issue order (CI instructions per chunk, retired in order):
    //   vec(NVEC) C0 C1 | C2 | C3 | ... first pass: C0..C6 = M (6 x 4 rows + row 24), then the J chunks;
    //   "| Cn" = issued right after chunk n-2 has been consumed, into the slot (n % 2) that chunk vacated.
    //   Second pass over J (for u -= J^T t; keeping J resident would cost 21 KB of LDS and the second wave
    //   per SIMD): the last chunks (rows 8..) stay where they are in the ring; rows 0..7 are loaded straight into
    //   registers at the start of the k x k phase, when Y has died, so the torque phase finds everything in place
    //   and never waits on memory.
    dmalinear<PDQ>(p.dq + t0 * N, vec + VEC_DQ, lane);
    dmalinear<PEE>(p.ee + t0 * NDEV * 7, vec + VEC_EE, lane);
    dmalinear<PEE>(p.tgt + t0 * NDEV * 7, vec + VEC_TGT, lane);
    const float* Mt = p.M + t0 * (N * N);
    const float* Jt = p.J + t0 * (K * N);
    // chunk m lives in ring slot m % NB: M (6 x 4 rows + row 24), then J
    auto issue = [&](int m) {
        if (m >= NT) return;
        float* dst = ring + (m % NB) * SLOT;
        if (m < 6) dma4rows<N * N * 4>(Mt + m * 4 * N, dst, lane);
        else if (m == 6) dma1row<N * N * 4>(Mt + 24 * N, dst, lane);
        else {
            const int jc = m - NCHM;
            if (jc < NJ4) dma4rows<K * N * 4>(Jt + jc * 4 * N, dst, lane);
            else dma1row<K * N * 4>(Jt + (4 * NJ4 + (jc - NJ4)) * N, dst, lane);
        }
    };
    // Rule: after chunk m has been consumed, chunk m + NB is issued into the slot it vacated.  Hence "younger DMAs
    // possibly in flight while consuming m" = min(NB - 1, NT - 1 - m) chunks.
#pragma unroll
    for (int m = 0; m < NB; ++m) issue(m);

    // ---------------- register state -----------------------------------------------------------------------
    // Row slots are kept as PAIRS (slots 2p, 2p+1 in one float2) so the multiply-adds are v_pk_fma_f32.
    struct Row { v2f p[P]; float o; };       // `o` = the scalar odd slot (G = 4: slot 6), unused for G = 8
    v2f Lp[P][24];         // strictly-lower rows 0..23 of L owned by this lane; upper/diagonal entries are 0
    Row dinv, mdq;         // 1/L[i][i], (M dq)[i] for the lane's own rows
    Row Y[K];              // own rows of Y = L^-1 J^T
    uint32_t flags = 0;
#pragma unroll
    for (int pp = 0; pp < P; ++pp) { dinv.p[pp] = v2f{0.f, 0.f}; mdq.p[pp] = v2f{0.f, 0.f}; }
    dinv.o = 0.f; mdq.o = 0.f;

    const bool lastpad = g != 0;                 // the last slot (row 24 + padding) is real only on lane g == 0
    const int lastcol = lastpad ? 0 : 24;        // safe in-range column for the masked slot
    float* xq = vec + VEC_X + q * (N + K);       // per-instance exchange: [0..24] Mdq, [25..25+K) dx

    // slot accessors (s is a compile-time constant wherever these are used)
    auto sget = [&](const Row& r, int s) -> float {
        if (ODD && s == LS) return r.o;
        return (s & 1) ? r.p[s >> 1].y : r.p[s >> 1].x;
    };
    // read one 25-float row of this instance: real slots into pairs (+ scalar), padding -> 0
    auto load_row = [&](const float* row, Row& d) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float v;
            if (s == LS) { const float t = row[lastcol]; v = lastpad ? 0.f : t; }
            else v = row[G * s + g];
            if (ODD && s == LS) d.o = v;
            else if (s & 1) d.p[s >> 1].y = v;
            else d.p[s >> 1].x = v;
        }
        if (!ODD) { /* all slots are in pairs */ } else { /* d.o set above */ }
    };

    wait_vm<NB * CI>();                   // the vector DMAs have landed (NB chunks still in flight)
    IRLOSC_TS(1);

    // ---------------- task-space error of device g (lane g < NDEV of the group), part 1 -------------------------
    // Done here, while the first M chunk is still in flight: quaternion -> sxyz Euler error, velocity limit and
    // gains need only ee/tgt/gains.  The rows that depend on dx (branch B) are finished after the J phase.
    // (gains pointer and null-space gain are re-derived where they are used: carrying them through the
    //  register-critical phases spilled them to scratch)
    float e6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float kv_own = 0.f;
    if (g < NDEV) {
        const DevMeta dm = dm1;
        float ee[7], tg[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            ee[i] = vec[VEC_EE + (q * NDEV + g) * 7 + i];
            tg[i] = vec[VEC_TGT + (q * NDEV + g) * 7 + i];
        }
        task_error6<float>(ee, tg, dm.calc & 1u, dm.calc & 2u, e6);
        apply_gains6<float>(gl, e6);
        // park the controlled rows in the exchange area (LDS) instead of holding six registers through the
        // register-critical phases; part 2 only touches them on the rare branch-B / admittance paths
        float* wl0 = vec + VEC_W + q * K;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (dm.dofmask & (1u << i)) { wl0[dm.row0 + cnt] = e6[i]; ++cnt; }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- stream M: Cholesky of the leading 24 x 24 block, column by column ---------------------------
    // 24 rows = six full slots on every lane, so this part is packed pairs only.  Row/column 24 is handled after
    // the loop as one more right-hand side of the forward substitution (L11 y = M[0:24][24] gives row 24 of L).
    static_assert(G == 4 && ODD && N == 25, "row 24 is treated as the bordering row of a 24 x 24 factor");
    auto load_row24 = [&](const float* row, Row& d) {          // slots 0..5 only
#pragma unroll
        for (int s = 0; s < LS; ++s) {
            const float v = row[G * s + g];
            if (s & 1) d.p[s >> 1].y = v; else d.p[s >> 1].x = v;
        }
    };
#pragma unroll
    for (int ch = 0; ch < NCHM - 1; ++ch) {
        float* buf = ring + (ch % NB) * SLOT;
        wait_chunks<CI>((NT - 1 - ch) < (NB - 1) ? (NT - 1 - ch) : (NB - 1));   // chunk ch has landed
        Row mrow[4];
        float dqj[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {    // all rows of the chunk up front: one LDS round trip per chunk
            load_row24(buf + q * GE::STR4 + rr * N, mrow[rr]);
            dqj[rr] = vec[VEC_DQ + q * N + ch * 4 + rr];
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int j = ch * 4 + rr;
            const int sj = j / G, gj = j % G;                    // row j lives in slot sj of group lane gj
            const int pj = sj >> 1;
            const v2f dq2 = v2f{dqj[rr], dqj[rr]};
#pragma unroll
            for (int pp = 0; pp < P; ++pp) mdq.p[pp] = __builtin_elementwise_fma(mrow[rr].p[pp], dq2, mdq.p[pp]);   // M symmetric
            // Order fence (no instructions): instruction selection linearises this one huge basic block as it
            // likes and, left alone, hoists the row-j broadcasts of later columns and sinks the Mdq chain, which
            // keeps hundreds of values live.  Passing the operands through an empty volatile asm pins them.
#pragma unroll
            for (int c = 0; c < j; ++c) asm volatile("" : "+v"(Lp[pj][c]));
            // left-looking column j: acc = M[j][i] - sum_{c<j} L[i][c] L[j][c] for the rows i >= j
            v2f acc[P];
#pragma unroll
            for (int pp = 0; pp < P; ++pp) acc[pp] = mrow[rr].p[pp];
#pragma unroll
            for (int c = 0; c < j; ++c) {
                const float own = (sj & 1) ? Lp[pj][c].y : Lp[pj][c].x;
                const float ljs = -gbcast<G>(own, gj);
                const v2f lj = v2f{ljs, ljs};
#pragma unroll
                for (int pp = pj; pp < P; ++pp) acc[pp] = __builtin_elementwise_fma(Lp[pp][c], lj, acc[pp]);
            }
            float d = gbcast<G>((sj & 1) ? acc[pj].y : acc[pj].x, gj);
            flags |= !(d > 0.f) ? IRLOSC_FLAG_M_NOT_PD : 0u;   // also catches NaN
            d = fmaxf(d, 1e-30f);                               // keeps the factorisation finite; the flag tells
            const float di = __builtin_amdgcn_rsqf(d);
            const v2f di2 = v2f{di, di};
            const bool own_row = (g == gj);
            const float below = (g > gj) ? 1.f : 0.f;
#pragma unroll
            for (int pp = pj + 1; pp < P; ++pp) Lp[pp][j] = acc[pp] * di2;
            {   // the pair holding slot sj: rows above / on the diagonal get exact zeros
                const v2f sc2 = acc[pj] * di2;
                if (sj & 1) Lp[pj][j] = v2f{0.f, sc2.y * below};
                else Lp[pj][j] = v2f{sc2.x * below, sc2.y};
            }
            if (sj & 1) dinv.p[pj].y = own_row ? di : dinv.p[pj].y;
            else dinv.p[pj].x = own_row ? di : dinv.p[pj].x;
            // pair pj is complete once the last row of its second slot has been the pivot: row-scale it now
            // (L'[i][c] = L[i][c] / L[i][i]: the substitutions then need no per-column multiply, y_c is the
            // running b'_c itself)
            if ((j % (2 * G)) == 2 * G - 1) {
#pragma unroll
                for (int c = 0; c <= j; ++c) Lp[pj][c] = Lp[pj][c] * dinv.p[pj];
            }
#pragma unroll
            for (int pp = 0; pp < P; ++pp) asm volatile("" : "+v"(mdq.p[pp]));
#pragma unroll
            for (int pp = pj; pp < P; ++pp) asm volatile("" : "+v"(Lp[pp][j]));
            __builtin_amdgcn_sched_barrier(0);
        }
        // recycle the ring slot just consumed
        wait_lgkm0();
        issue(ch + NB);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---------------- row 24: the bordering row -----------------------------------------------------------------
    // M's row 24 arrives laid out exactly like a right-hand side (lane g, slot s holds M[24][4s+g] = M[4s+g][24]).
    // y = L11^-1 m is row 24 of L, and it comes out DISTRIBUTED (lane g keeps L[24][4m+g]), which is the form the
    // J substitutions want: there lane g multiplies L'[24][4m+g] with ITS OWN y values, one packed FMA per pair of
    // slots and no broadcast.
    v2f Lq[P];                                  // L'[24][4(2p)+g], L'[24][4(2p+1)+g]
    {
        constexpr int ch = NCHM - 1;
        float* buf = ring + (ch % NB) * SLOT;
        wait_chunks<CI>((NT - 1 - ch) < (NB - 1) ? (NT - 1 - ch) : (NB - 1));
        Row m24, dqc;
        load_row(buf + q * N, m24);             // .o = M[24][24] on lane 0, 0 elsewhere
        load_row(vec + VEC_DQ + q * N, dqc);    // .o = dq[24] on lane 0
        const float dq24 = gbcast<G>(dqc.o, 0);
        const float m2424 = gbcast<G>(m24.o, 0);
        const v2f dq242 = v2f{dq24, dq24};
        v2f dt = v2f{0.f, 0.f};
#pragma unroll
        for (int pp = 0; pp < P; ++pp) {
            mdq.p[pp] = __builtin_elementwise_fma(m24.p[pp], dq242, mdq.p[pp]);     // column 24 into the rows < 24
            dt = __builtin_elementwise_fma(m24.p[pp], dqc.p[pp], dt);               // row 24 itself
        }
        mdq.o = fmaf(m2424, dq24, gsum<G>(dt.x + dt.y));
        v2f bq[P];
#pragma unroll
        for (int pp = 0; pp < P; ++pp) bq[pp] = m24.p[pp] * dinv.p[pp];
#pragma unroll
        for (int c = 0; c < 24; ++c) {
            const int sc = c / G, gc = c % G, pc = sc >> 1;
            const float ycs = -gbcast<G>((sc & 1) ? bq[pc].y : bq[pc].x, gc);
            const v2f yc = v2f{ycs, ycs};
#pragma unroll
            for (int pp = pc; pp < P; ++pp) bq[pp] = __builtin_elementwise_fma(Lp[pp][c], yc, bq[pp]);
        }
        v2f n2 = bq[0] * bq[0];
#pragma unroll
        for (int pp = 1; pp < P; ++pp) n2 = __builtin_elementwise_fma(bq[pp], bq[pp], n2);
        float d = m2424 - gsum<G>(n2.x + n2.y);
        flags |= !(d > 0.f) ? IRLOSC_FLAG_M_NOT_PD : 0u;
        d = fmaxf(d, 1e-30f);
        const float di = __builtin_amdgcn_rsqf(d);
        dinv.o = lastpad ? 0.f : di;
        const v2f di2 = v2f{di, di};
#pragma unroll
        for (int pp = 0; pp < P; ++pp) Lq[pp] = bq[pp] * di2;
#pragma unroll
        for (int pp = 0; pp < P; ++pp) asm volatile("" : "+v"(Lq[pp]), "+v"(mdq.p[pp]));
        asm volatile("" : "+v"(mdq.o));
        wait_lgkm0();
        issue(ch + NB);
        __builtin_amdgcn_sched_barrier(0);
    }
    IRLOSC_TS(2);
    // park Mdq in LDS (own real rows)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s == LS) { if (!lastpad) xq[24] = sget(mdq, s); }
        else xq[G * s + g] = sget(mdq, s);
    }

    // ---------------- J rows: dx and forward substitutions ----------------------------------------------------
#pragma unroll
    for (int jc = 0; jc < NCHJ; ++jc) {
        const int n = NCHM + jc;
        float* buf = ring + (n % NB) * SLOT;
        wait_chunks<CI>((NT - 1 - n) < (NB - 1) ? (NT - 1 - n) : (NB - 1));
        const int jstride = jc < NJ4 ? GE::STR4 : N;
        const int jrow0 = jc < NJ4 ? 4 * jc : 4 * NJ4 + (jc - NJ4);      // first row of J in this chunk
        Row dqc;                                   // dq of the own rows: re-read per chunk (7 registers saved)
        load_row(vec + VEC_DQ + q * N, dqc);
        const int R = jc < NJ4 ? 4 : 1;
        Row bb[4];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            load_row(buf + q * jstride + rr * N, bb[rr]);
            v2f dx2 = v2f{0.f, 0.f};
#pragma unroll
            for (int pp = 0; pp < P; ++pp) dx2 = __builtin_elementwise_fma(bb[rr].p[pp], dqc.p[pp], dx2);
            float dxs = dx2.x + dx2.y;
            if (ODD) dxs = fmaf(bb[rr].o, dqc.o, dxs);
            dxs = gsum<G>(dxs);
            if (g == 0) xq[25 + jrow0 + rr] = dxs;
#pragma unroll
            for (int pp = 0; pp < P; ++pp) bb[rr].p[pp] = bb[rr].p[pp] * dinv.p[pp];     // b' = D^-1 b
            if (ODD) bb[rr].o *= dinv.o;
        }
        // column-oriented substitution: y_c = b_c / L[c][c] (owner lane), then b_i -= L[i][c] y_c.  Two rows are
        // advanced together, column by column: a row's broadcast of y_c reads a register its own previous
        // multiply-adds have just written (a DPP source needs two wait states after a VALU write), and the other
        // row's instructions fill exactly that gap instead of an s_nop.
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 2) {
            const int RR = (R - r0) < 2 ? (R - r0) : 2;
            v2f part24[2] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};          // this lane's share of sum_c L'[24][c] y_c
#pragma unroll
            for (int c = 0; c < 24; ++c) {
                const int sc = c / G, gc = c % G, pc = sc >> 1;
#pragma unroll
                for (int h = 0; h < RR; ++h) {
                    Row& bw = bb[r0 + h];
                    const float ycs = -gbcast<G>(sget(bw, sc), gc);      // y_c = b'_c, final once columns < c are done
                    const v2f yc = v2f{ycs, ycs};
#pragma unroll
                    for (int pp = pc; pp < P; ++pp) bw.p[pp] = __builtin_elementwise_fma(Lp[pp][c], yc, bw.p[pp]);
                    if (gc == G - 1 && (sc & 1))     // pair pc is final in every lane of the group now
                        part24[h] = __builtin_elementwise_fma(Lq[pc], bw.p[pc], part24[h]);
                    asm volatile("" : "+v"(part24[h]), "+v"(bw.p[P - 1]));
                }
            }
#pragma unroll
            for (int h = 0; h < RR; ++h) {
                const int rr = r0 + h;
                if (ODD) {
                    const float y24 = bb[rr].o - gsum<G>(part24[h].x + part24[h].y);            // b'_24 sits on lane 0 only
                    bb[rr].o = lastpad ? 0.f : y24;
                }
#pragma unroll
                for (int pp = 0; pp < P; ++pp) Y[jrow0 + rr].p[pp] = bb[rr].p[pp];
                Y[jrow0 + rr].o = ODD ? bb[rr].o : 0.f;
#pragma unroll
                for (int pp = 0; pp < P; ++pp) asm volatile("" : "+v"(Y[jrow0 + rr].p[pp]));
                if (ODD) asm volatile("" : "+v"(Y[jrow0 + rr].o));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_lgkm0();
        issue(n + NB);                                // nothing beyond the last chunk: the tail of J stays resident
        __builtin_amdgcn_sched_barrier(0);
    }

    IRLOSC_TS(3);
    // ---------------- task-space signal, part 2: rows of device g into the exchange area --------------------------
    float* wls = vec + VEC_W + q * K;
    int brA_own = 1;
    __builtin_amdgcn_wave_barrier();
    float kvn = 0.f;
    if (p.cfgflags & IRLOSC_NULLSPACE) kvn = p.null_kv[p.gains_per_instance ? b : 0];
    if (g < NDEV) {
        const DevMeta dm = pod_copy<DevMeta>(p.dev[g]);
        const float* gg = gains_ptr() + g * IRLOSC_GAIN_WORDS;
        kv_own = gg[1];
        float tv[6];
        bool all_nonzero = has_tv;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            tv[i] = has_tv ? p.tvel[((size_t)b * NDEV + g) * 6 + i] : 0.f;
            all_nonzero = all_nonzero && (tv[i] != 0.f);
        }
        brA_own = all_nonzero ? 0 : 1;
        if (all_nonzero) {
            flags |= IRLOSC_FLAG_VEL_BRANCH_B;
            if (dm.jidx0 + dm.rows > K) flags |= IRLOSC_FLAG_BAD_JIDX;
        }
        if (all_nonzero || has_wr) {
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (dm.dofmask & (1u << i)) {
                    float v = wls[dm.row0 + cnt];
                    if (all_nonzero) {
                        const int row = dm.jidx0 + cnt;
                        const float dxv = xq[25 + (row < K ? row : 0)];
                        const float damp = (i < 3) ? gg[6 + i] : 1.f;
                        v += kv_own * ((row < K ? dxv : 0.f) - tv[i]) * damp;
                    }
                    if (has_wr) v += p.wrench[((size_t)b * NDEV + g) * 6 + i];
                    wls[dm.row0 + cnt] = v;
                    ++cnt;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    wait_lgkm0();
    // The k x k matrix is REPLICATED in the group (every lane does the same arithmetic), so its throughput is
    // bought with packed FP32: everything below keeps A column-major in ROW PAIRS, Ac[c][m] = (A[2m][c], A[2m+1][c])
    // for m >= c/2.  The low half of the first pair of an odd column (row c-1, upper triangle) and row 13 of the
    // k = 13 padding are don't-care / zero slots that no result ever reads.
    constexpr int KP = (K + 1) / 2;
    v2f w2[KP];
#pragma unroll
    for (int m = 0; m < KP; ++m) {
        w2[m].x = wls[2 * m] - kvn * xq[25 + 2 * m];
        w2[m].y = (2 * m + 1 < K) ? wls[2 * m + 1] - kvn * xq[25 + 2 * m + 1] : 0.f;
    }
#define IRLOSC_WE(r) (((r) & 1) ? w2[(r) / 2].y : w2[(r) / 2].x)

    IRLOSC_TS(4);
    // ---------------- A = Y^T Y (lower), replicated in the group -----------------------------------------------
    v2f Ac[K][KP];
#pragma unroll
    for (int c = 0; c < K; ++c) {
#pragma unroll
        for (int m = 0; m < KP; ++m) Ac[c][m] = v2f{0.f, 0.f};
    }
#define IRLOSC_AE(r, c) (((r) & 1) ? Ac[c][(r) / 2].y : Ac[c][(r) / 2].x)
    // Park A next to the resident tail of J (behind J3's row in its slot; the records that do not fit there go to
    // the dq | ee | tgt | W area, which is dead by now).  It is read back only by flagged instances, which hand A
    // and w to the second stage.
    static_assert((TILE - A_FIT) * NA <= VEC_X - VEC_DQ, "overflow A records must fit in the dead vector area");
    float* aq = q < A_FIT ? ring + SLOT_A * SLOT + A_OFF + q * NA : vec + VEC_DQ + (q - A_FIT) * NA;
    float nA2 = 0.f;
    // row by row: first all partial dots of the row, then the butterfly steps over the whole row, so that a
    // DPP add never has to wait on the instruction right before it (rows 0..2 are too short to hide it)
#pragma unroll
    for (int r = 0; r < K; ++r) {
        float ar[K];
#pragma unroll
        for (int s2 = 0; s2 <= r; ++s2) {
            v2f a2 = Y[r].p[0] * Y[s2].p[0];
#pragma unroll
            for (int pp = 1; pp < P; ++pp) a2 = __builtin_elementwise_fma(Y[r].p[pp], Y[s2].p[pp], a2);
            float a = a2.x + a2.y;
            if (ODD) a = fmaf(Y[r].o, Y[s2].o, a);
            ar[s2] = a;
        }
#pragma unroll
        for (int s2 = 0; s2 <= r; ++s2) asm volatile("" : "+v"(ar[s2]));
#pragma unroll
        for (int s2 = 0; s2 <= r; ++s2) ar[s2] += dpp_f<0xB1>(ar[s2]);      // quad_perm [1,0,3,2]
#pragma unroll
        for (int s2 = 0; s2 <= r; ++s2) asm volatile("" : "+v"(ar[s2]));
#pragma unroll
        for (int s2 = 0; s2 <= r; ++s2) ar[s2] += dpp_f<0x4E>(ar[s2]);      // quad_perm [2,3,0,1]
        if (G == 8) {
#pragma unroll
            for (int s2 = 0; s2 <= r; ++s2) asm volatile("" : "+v"(ar[s2]));
#pragma unroll
            for (int s2 = 0; s2 <= r; ++s2) ar[s2] += dpp_f<0x141>(ar[s2]); // row_half_mirror
        }
#pragma unroll
        for (int s2 = 0; s2 <= r; ++s2) asm volatile("" : "+v"(ar[s2]));
#pragma unroll
        for (int s2 = 0; s2 <= r; ++s2) {
            if (r & 1) Ac[s2][r / 2].y = ar[s2]; else Ac[s2][r / 2].x = ar[s2];
            aq[r * (r + 1) / 2 + s2] = ar[s2];   // every lane of the group holds the same value: plain broadcast store
            nA2 = fmaf(s2 < r ? 2.f * ar[s2] : ar[s2], ar[s2], nA2);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    IRLOSC_TS(5);
    asm volatile("" : "+v"(nA2));
    __builtin_amdgcn_sched_barrier(0);
    // second pass over J, rows 0..7: own-row elements straight from global memory (the registers Y held are free)
    // and the bias forces of the own rows.  Raw loads only: nothing here may consume a loaded value, or the wave
    // would sit out the memory latency right now (the padding slot carries junk that is never stored).
    Row jd[NDL > 0 ? NDL : 1];
    Row biasr;
    {
        const float* jg = p.J + (size_t)b * (K * N);
        auto load_row_raw = [&](const float* row, Row& d) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float v = row[s == LS ? lastcol : G * s + g];
                if (ODD && s == LS) d.o = v;
                else if (s & 1) d.p[s >> 1].y = v;
                else d.p[s >> 1].x = v;
            }
        };
#pragma unroll
        for (int r = 0; r < NDL; ++r) load_row_raw(jg + r * N, jd[r]);
#pragma unroll
        for (int pp = 0; pp < P; ++pp) biasr.p[pp] = v2f{0.f, 0.f};
        biasr.o = 0.f;
        if (p.cfgflags & IRLOSC_USE_G) load_row_raw(p.bias + (size_t)b * N, biasr);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- k x k: right-looking Cholesky of A in place (row pairs), cond certificate ---------------------
    bool pdA = true;
    float detA = 1.f;
    float dA[K];                       // 1 / L_A[j][j]
#pragma unroll
    for (int j = 0; j < K; ++j) {
        float d = IRLOSC_AE(j, j);
        const bool npd = !(d > 0.f);
        pdA = pdA && !npd;
        const float dfix = (d == d && d != 0.f) ? fabsf(d) : 1.f;
        d = npd ? dfix : d;
        detA *= d;
        const float di = __builtin_amdgcn_rsqf(d);
        dA[j] = di;
        const v2f di2 = v2f{di, di};
#pragma unroll
        for (int m = j / 2; m < KP; ++m) Ac[j][m] = Ac[j][m] * di2;     // the diagonal slot becomes sqrt(d): never read
#pragma unroll
        for (int c2 = j + 1; c2 < K; ++c2) {
            const float l = IRLOSC_AE(c2, j);
            const v2f l2 = v2f{l, l};
#pragma unroll
            for (int m = c2 / 2; m < KP; ++m) Ac[c2][m] = __builtin_elementwise_fma(-Ac[j][m], l2, Ac[c2][m]);
        }
#pragma unroll
        for (int m = j / 2; m < KP; ++m) asm volatile("" : "+v"(Ac[j][m]));
        asm volatile("" : "+v"(dA[j]), "+v"(detA));
    }
    // ||L_A^-1||_F^2 = trace(A^-1).  The columns of W = L_A^-1 are independent forward substitutions with the SAME
    // matrix, so lane g of the group takes column G*r + g in round r (a different unit right-hand side per lane,
    // identical instruction stream), and the partial sums are added over the group at the end.
    float eg[G];
#pragma unroll
    for (int m = 0; m < G; ++m) eg[m] = (g == m) ? 1.f : 0.f;
    float nW2 = 0.f;
#pragma unroll
    for (int c0 = 0; c0 < K; c0 += G) {
        v2f wp[KP];
#pragma unroll
        for (int m = c0 / 2; m < KP; ++m) {
            wp[m].x = (2 * m - c0 < G && 2 * m < K) ? eg[(2 * m - c0) % G] : 0.f;
            wp[m].y = (2 * m + 1 - c0 < G && 2 * m + 1 < K) ? eg[(2 * m + 1 - c0) % G] : 0.f;
        }
#pragma unroll
        for (int c = c0; c < K; ++c) {
            const float wc = ((c & 1) ? wp[c / 2].y : wp[c / 2].x) * dA[c];
            nW2 = fmaf(wc, wc, nW2);
            const v2f wc2 = v2f{wc, wc};
#pragma unroll
            for (int m = (c + 1) / 2; m < KP; ++m) {
                const bool has = (2 * m > c && 2 * m < K) || (2 * m + 1 > c && 2 * m + 1 < K);
                if (has) wp[m] = __builtin_elementwise_fma(-Ac[c][m], wc2, wp[m]);
            }
        }
        asm volatile("" : "+v"(nW2));
    }
    nW2 = gsum<G>(nW2);
    __builtin_amdgcn_sched_barrier(0);
    const bool small_det = !(fabsf(detA) >= 1e-4f);
    const float cond_bound = sqrtf(nA2) * nW2;        // >= cond_2(A) for SPD A
    const bool plain = pdA && t_finite(cond_bound) && (!small_det || cond_bound < 0.99e5f);
    flags |= small_det ? IRLOSC_FLAG_PINV_BRANCH : 0u;
    // Flagged instances hand A and w to the second stage now: one contiguous record of NA + K floats per instance
    // (indexed by instance: no atomics; stage-2 blocks compact their span from the flag words).  The group writes
    // 16 contiguous bytes per store instruction, so a record costs its own bytes in write traffic and no more.
    if (!plain) {
        flags |= IRLOSC_FLAG_EIGEN_PATH;
        float* rec = side + (size_t)b * (NA + K);
        for (int e = g; e < NA; e += G) rec[e] = aq[e];
#pragma unroll
        for (int r = 0; r < K; ++r)
            if ((r % G) == g) rec[NA + r] = IRLOSC_WE(r);
    }
    __builtin_amdgcn_sched_barrier(0);
    float t[K];
    // forward, column-oriented on row pairs: z = L_A^-1 w
#pragma unroll
    for (int c = 0; c < K; ++c) {
        t[c] = IRLOSC_WE(c) * dA[c];
        const v2f t2 = v2f{t[c], t[c]};
#pragma unroll
        for (int m = (c + 1) / 2; m < KP; ++m) {
            const bool has = (2 * m > c && 2 * m < K) || (2 * m + 1 > c && 2 * m + 1 < K);
            if (has) w2[m] = __builtin_elementwise_fma(-Ac[c][m], t2, w2[m]);
        }
    }
    // backward, dot form over the row pairs of column i: t = L_A^-T z
    v2f tp[KP];
#pragma unroll
    for (int m = 0; m < KP; ++m) tp[m] = v2f{0.f, 0.f};
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
        v2f acc = v2f{0.f, 0.f};
#pragma unroll
        for (int m = (i + 2) / 2; m < KP; ++m) acc = __builtin_elementwise_fma(Ac[i][m], tp[m], acc);
        float s2 = t[i] - (acc.x + acc.y);
        if (!(i & 1) && i + 1 < K) s2 = fmaf(-IRLOSC_AE(i + 1, i), t[i + 1], s2);
        t[i] = s2 * dA[i];
        if (i & 1) tp[i / 2].y = t[i]; else tp[i / 2].x = t[i];
    }
#undef IRLOSC_AE
#undef IRLOSC_WE

    IRLOSC_TS(6);
    // keep the scheduler from hoisting the ~120 LDS reads of the torque phase above the k x k work
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- joint torques for the own rows: u = u0 + bias - kvn*Mdq - J^T t ----------------------------
    Row jt;                                   // (J^T t) for the own rows
#pragma unroll
    for (int pp = 0; pp < P; ++pp) jt.p[pp] = v2f{0.f, 0.f};
    jt.o = 0.f;
    wait_vm<0>();                             // the register rows were issued a whole k x k phase ago
    {
#pragma unroll
        for (int r = 0; r < NDL; ++r) {
            const v2f t2 = v2f{t[r], t[r]};
#pragma unroll
            for (int pp = 0; pp < P; ++pp) jt.p[pp] = __builtin_elementwise_fma(jd[r].p[pp], t2, jt.p[pp]);
            if (ODD) jt.o = fmaf(jd[r].o, t[r], jt.o);      // padding lanes: junk in, never stored
        }
#pragma unroll
        for (int r = NDL; r < K; ++r) {                     // the resident tail: the chunk before the last (if intact), the last
            Row jr;
            if (r < ROW_LAST) load_row(ring + SLOT_PREV * SLOT + q * (PREV_ROWS == 4 ? GE::STR4 : N) + (r - ROW_PREV) * N, jr);
            else load_row(ring + SLOT_LAST * SLOT + q * (LAST_ROWS == 4 ? GE::STR4 : N) + (r - ROW_LAST) * N, jr);
            const v2f t2 = v2f{t[r], t[r]};
#pragma unroll
            for (int pp = 0; pp < P; ++pp) jt.p[pp] = __builtin_elementwise_fma(jr.p[pp], t2, jt.p[pp]);
            if (ODD) jt.o = fmaf(jr.o, t[r], jt.o);
        }
        wait_lgkm0();
        __builtin_amdgcn_sched_barrier(0);
    }
    bool bad = false;
    float* ubuf = ring;                       // both ring slots have been drained by now (LDS ops of a wave are in order)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = G * s + g;
        const bool valid = (s < LS) || !lastpad;
        const int icol = valid ? i : 0;
        const float mdq_i = xq[icol];
        float uu = 0.f;
#pragma unroll
        for (int d = 0; d < NDEV; ++d) {
            const int brA_d = gbcast_i<G>(brA_own, d);
            const float kv_d = gbcast<G>(kv_own, d);
            if (brA_d && (p.dev[d].joint_mask & (1u << (i & 31)))) uu = -kv_d * mdq_i;
        }
        uu -= plain ? sget(jt, s) : 0.f;     // flagged instances keep u_base; stage 2 subtracts J^T t
        uu += sget(biasr, s);                // zeros unless IRLOSC_USE_G
        uu -= kvn * mdq_i;
        if (valid) {
            ubuf[q * N + i] = uu;
            bad = bad || !t_finite(uu);
        }
    }
    // u leaves through LDS as whole 16-byte pieces of the tile's contiguous [TILE][25] block: per-lane
    // stores at a 100-byte instance stride were partial-line writes (WRITE_SIZE 3.6x the payload)
    __builtin_amdgcn_wave_barrier();
    wait_lgkm0();
    {
        const float4* src4 = reinterpret_cast<const float4*>(ubuf);
        float4* dst4 = reinterpret_cast<float4*>(p.u + t0 * N);
#pragma unroll
        for (int j = 0; j < (TILE * N / 4 + 63) / 64; ++j) {
            const int x = j * 64 + lane;
            if (x < TILE * N / 4) dst4[x] = src4[x];
        }
    }
    flags |= bad ? IRLOSC_FLAG_NONFINITE : 0u;
    flags = gor<G>(flags);
    if (g == 0) p.flags[b] = flags;
    IRLOSC_TS(7);
    if (p.dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p.dbg[(size_t)tile * 10 + i] = ts[i];
        p.dbg[(size_t)tile * 10 + 8] = rt0 | ((unsigned long long)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 0xf) << 60);   // XCC_ID in the top bits
        p.dbg[(size_t)tile * 10 + 9] = __builtin_amdgcn_s_memrealtime();
    }
#undef IRLOSC_TS
}

}  // namespace irlosc
