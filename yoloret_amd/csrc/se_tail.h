// Squeeze-excite finished by the kernel that produces the map ("SE tail", ABI 7).
//
// The SE block (reference code/yolo3/efficientnet.py:406-438: Mean over H, W -> 1x1 + bias -> Swish -> 1x1 + bias ->
// sigmoid) needs the channel means of the COMPLETE depthwise map; the kernels that produce that map already leave per-workgroup
// channel sums (rows of a small float32 buffer).  Through round 4 a separate launch (se_fc_kernel, one workgroup per image)
// added the rows up and ran the two tiny FCs: 6 launches of 11-22 us in every detection head, 21-30 in the SE EfficientNets, at
// 0.003 of any pipe - a latency chain.  Here the workgroup that COMPLETES an image's rows does it on the spot:
//   * every workgroup stores its sums with AGENT-scope stores (yr_st_agent: write-through to the device's coherence point), waits
//     for them (vmcnt(0) + the workgroup barrier) and then adds its share to the image's arrival counter (agent-scope atomic);
//   * the one that brings the counter to `arrivals` (all others have arrived before it) resets the counter, reads all rows
//     with agent-scope loads (they were written on other CUs / XCDs), and computes mean -> FC1 -> swish -> FC2 -> sigmoid in a
//     FIXED order - rows added in index order, hidden units and channels as sequential fma chains - so the gate does not depend
//     on which workgroup happens to be last (a batch still equals its images run one by one).
// What this deliberately does NOT use is __threadfence(): on gfx950 an agent-scope release is an L2 write-back (buffer_wbl2) of
// everything the XCD holds dirty - i.e. of the map the kernel is busy writing - and measured 5 x the kernel's time (dw_kernel on
// 52 x 52 x 128 @64: 46 -> 250 us with a fence per workgroup).  Only the sums need to be visible device-wide, so only they are
// written through; the map reaches memory at the kernel boundary like every other output.
// 131 k multiply-adds per image at most (F = 512): a few microseconds on one CU, spread over as many CUs as there are images,
// under the tail of the producing kernel.  No grid-wide barrier (round 3 measured that dead end), no co-residency assumption.
//
// Counters: one unsigned per image, zero before the launch and zero again after it (yr_forward clears the plan's counters at the
// start of every pass as well; yr_op_run callers hand in zeroed memory once).
#pragma once
#include "yr_common.h"

struct SeTail {
    const float* sums;   // [B][rows][ld_sums] partial channel sums written by this launch (nullptr: no tail)
    int rows, ld_sums;
    float count;         // pixels per image (what the summed rows are divided by)
    const float* w;      // packed FC parameters: W1 [ldc][R4] | W2 [R][ldc] | b1 [R4] | b2 [ldc], R4 = round_up(R, 4), ldc = round_up(C, 4)
    float* gate;         // [B][ld_gate] out
    int ld_gate, C, R, ldc;
    unsigned* sync;      // [B] arrival counters
    unsigned arrivals;   // what an image's counter reaches when all of its rows are stored (set by the launcher)
};

static inline size_t yr_se_tail_floats(int C, int R, int nth = 256) { return (size_t)yr_round_up(C, 4) + (size_t)yr_round_up(R, 4) + 4 * (size_t)nth; }   // LDS floats the tail needs
#define YR_SE_TAIL_LDS 4608   // floats of LDS the kernels with a tail set aside for it (C + R + 1024 <= 4608: every EfficientNet up to B6)

#ifdef __HIPCC__
// host: fill a SeTail from an op that carries the ABI-7 fields (gate = the sums buffer it writes, gate_out, se_w, sync)
static inline int yr_make_se_tail(const yr_op& op, int rows, SeTail* t) {
    t->sums = nullptr;
    if (op.gate_out == nullptr) return YR_OK;
    YR_REQUIRE(op.gate != nullptr && op.se_w != nullptr && op.sync != nullptr && op.se_hidden >= 1,
               "SE tail: gate_out needs the partial-sum buffer (gate), se_w, sync and se_hidden");
    YR_REQUIRE(op.gate_out_ld >= yr_round_up(op.cout, 4) && ((uintptr_t)op.gate_out % 16) == 0 && ((uintptr_t)op.se_w % 16) == 0, "SE tail: bad gate_out / se_w");
    YR_REQUIRE(yr_se_tail_floats(op.cout, op.se_hidden) <= YR_SE_TAIL_LDS, "SE tail: %d channels + %d hidden units exceed the tail's LDS scratch", op.cout, op.se_hidden);
    t->sums = op.gate; t->rows = rows; t->ld_sums = op.gate_ld;
    t->count = (float)(op.h * op.w);
    t->w = op.se_w; t->gate = op.gate_out; t->ld_gate = op.gate_out_ld;
    t->C = op.cout; t->R = op.se_hidden; t->ldc = yr_round_up(op.cout, 4);
    t->sync = op.sync; t->arrivals = (unsigned)rows;
    return YR_OK;
}

__device__ __forceinline__ float yr_ld_agent(const float* p) {   // a load that sees what other CUs / XCDs have written through
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the store of a partial sum: through to the device's coherence point (no L2 write-back fence needed later)
__device__ __forceinline__ void yr_st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void yr_st_agent4(float* p, float x, float y, float z, float w) {
    yr_st_agent(p, x); yr_st_agent(p + 1, y); yr_st_agent(p + 2, z); yr_st_agent(p + 3, w);
}

// The FC pair for image b by NTH cooperating threads (NTH = 64: one wave, no barriers; otherwise the whole workgroup).
// lds: yr_se_tail_floats(C, R, NTH) floats.  The tail runs when the kernel is about to end - nothing hides it - so every stage
// is cut for LATENCY: 16-byte loads, all NTH threads busy (a thread owns a quad of outputs and a SEGMENT of the inputs; the
// segments' partial sums meet in LDS in segment order), a handful of load batches per stage instead of one load per multiply-add
// (the first version - a thread per output, scalar loads - took 8-17 us per head block, as long as the se_fc launch it replaces).
// Fixed orders throughout: the gate is a function of the sums alone.
typedef float se_f4 __attribute__((ext_vector_type(4)));
typedef unsigned se_u4 __attribute__((ext_vector_type(4)));
template <int NTH>
__device__ __forceinline__ void yr_se_tail_fc(const SeTail& t, const int b, float* lds, const int tid) {
    const int R4 = (t.R + 3) & ~3, QP = t.ldc >> 2, JQ = R4 >> 2;
    float* mean = lds;                               // [ldc]
    float* hid = lds + t.ldc;                        // [R4]
    se_f4* part = reinterpret_cast<se_f4*>(hid + R4);   // [NTH] quads
    auto sync = [&]() { if constexpr (NTH > 64) __syncthreads(); };
    const se_f4 zero = (se_f4){0.f, 0.f, 0.f, 0.f};
    // ---- the channel means: the rows written through by the other workgroups, read with agent-scope (sc1) 16-byte loads
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t.sums) + (size_t)b * t.rows * t.ld_sums, 0, (unsigned)(t.rows * t.ld_sums) * 4u, 0x00020000);
        for (int q0 = 0; q0 < QP; q0 += NTH) {       // (one pass unless C > 4 NTH)
            const int qp = QP - q0 < NTH ? QP - q0 : NTH;          // quads of this pass
            const int nrs = NTH / qp, rps = (t.rows + nrs - 1) / nrs;   // row segments, rows per segment
            const int q = tid % qp, seg = tid / qp;
            se_f4 s = zero;
            if (seg < nrs) {
                const int ra = seg * rps, rb = ra + rps < t.rows ? ra + rps : t.rows;
#pragma unroll 4
                for (int r = ra; r < rb; ++r)
                    s += __builtin_bit_cast(se_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(r * t.ld_sums + 4 * (q0 + q)) * 4u, 0, 16));   // aux 16: sc1
            }
            part[tid] = s;
            sync();
            if (tid < qp) {
                se_f4 m = part[tid];
                for (int sg = 1; sg < nrs; ++sg) m += part[sg * qp + tid];
#pragma unroll
                for (int i = 0; i < 4; ++i) mean[4 * (q0 + tid) + i] = 4 * (q0 + tid) + i < t.C ? m[i] / t.count : 0.f;
            }
            sync();
        }
    }
    const float* w1 = t.w;                               // [ldc][R4]
    const float* w2 = t.w + (size_t)t.ldc * R4;          // [R][ldc]
    const float* b1 = w2 + (size_t)t.R * t.ldc;          // [R4]
    const float* b2 = b1 + R4;                           // [ldc]
    // ---- FC1 + swish: a thread = (channel segment, quad of hidden units)
    for (int j0 = 0; j0 < JQ; j0 += NTH) {
        const int jq = JQ - j0 < NTH ? JQ - j0 : NTH;
        const int nseg = NTH / jq, cps = (t.C + nseg - 1) / nseg;
        const int j = tid % jq, seg = tid / jq;
        se_f4 s = zero;
        if (seg < nseg) {
            const int ca = seg * cps, cb = ca + cps < t.C ? ca + cps : t.C;
            const float* wp = w1 + 4 * (j0 + j);
#pragma unroll 8
            for (int c = ca; c < cb; ++c) {
                const se_f4 w = *reinterpret_cast<const se_f4*>(wp + (size_t)c * R4);
                const float m = mean[c];
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] = __builtin_fmaf(w[i], m, s[i]);
            }
        }
        part[tid] = s;
        sync();
        if (tid < jq) {
            se_f4 v = *reinterpret_cast<const se_f4*>(b1 + 4 * (j0 + tid));
            for (int sg = 0; sg < nseg; ++sg) v += part[sg * jq + tid];
#pragma unroll
            for (int i = 0; i < 4; ++i) hid[4 * (j0 + tid) + i] = 4 * (j0 + tid) + i < t.R ? yr_apply_act(v[i], YR_ACT_SWISH) : 0.f;
        }
        sync();
    }
    // ---- FC2 + sigmoid: a thread = (hidden-unit segment, channel quad)
    for (int q0 = 0; q0 < QP; q0 += NTH) {
        const int qp = QP - q0 < NTH ? QP - q0 : NTH;
        const int nseg = NTH / qp, jps = (t.R + nseg - 1) / nseg;
        const int q = tid % qp, seg = tid / qp;
        se_f4 s = zero;
        if (seg < nseg) {
            const int ja = seg * jps, jb = ja + jps < t.R ? ja + jps : t.R;
            const float* wp = w2 + 4 * (q0 + q);
#pragma unroll 8
            for (int j = ja; j < jb; ++j) {
                const se_f4 w = *reinterpret_cast<const se_f4*>(wp + (size_t)j * t.ldc);
                const float hj = hid[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] = __builtin_fmaf(hj, w[i], s[i]);
            }
        }
        part[tid] = s;
        sync();
        if (tid < qp) {
            se_f4 v = *reinterpret_cast<const se_f4*>(b2 + 4 * (q0 + tid));
            for (int sg = 0; sg < nseg; ++sg) v += part[sg * qp + tid];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = 4 * (q0 + tid) + i < t.C ? yr_sigmoid(v[i]) : 0.f;
            *reinterpret_cast<se_f4*>(t.gate + (size_t)b * t.ld_gate + 4 * (q0 + tid)) = v;
        }
        sync();
    }
}

// Workgroup form: call by ALL NTH threads of the workgroup once its share of image b's sums is stored (uniform arguments); n: what it adds to the image's counter.
// flag: one LDS word; lds: yr_se_tail_floats() floats that are free by now.
template <int NTH>
__device__ __forceinline__ void yr_se_tail_arrive(const SeTail& t, const int b, const unsigned n, unsigned* flag, float* lds) {
    if (t.sums == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's write-through stores of the sums have been acknowledged
    __syncthreads();                                   // ... and every other thread's
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(t.sync + b, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old + n == t.arrivals;
        if (last) __hip_atomic_store(t.sync + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (nobody else touches it before the next launch)
        *flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (*flag == 0u || t.w == nullptr) return;   // (w == nullptr: probing - the arrival without the FC pair)
    yr_se_tail_fc<NTH>(t, b, lds, (int)threadIdx.x);
}

// Wave form, for kernels whose waves work on their own (no workgroup barrier may be used: the other waves may have left).
// lds: this WAVE's private yr_se_tail_floats() floats.
__device__ __forceinline__ void yr_se_tail_arrive_wave(const SeTail& t, const int b, const unsigned n, float* lds) {
    if (t.sums == nullptr) return;
    const int lane = threadIdx.x & 63;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores of the sums have been acknowledged
    unsigned last = 0u;
    if (lane == 0) {
        const unsigned old = __hip_atomic_fetch_add(t.sync + b, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = old + n == t.arrivals ? 1u : 0u;
        if (last) __hip_atomic_store(t.sync + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (last == 0u) return;
    yr_se_tail_fc<64>(t, b, lds, lane);
}
#endif
