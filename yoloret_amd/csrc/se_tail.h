// Squeeze-excite finished by the kernel that produces the map ("SE tail", ABI 7) - and the FC pair itself (yr_se_fc_pair), which
// se_fc_kernel shares.
//
// STATUS (round 5): the tail is OPT-IN (YOLORET_SE_TAIL=1), not what a plan gets by default.  It is bit-exact and deterministic in one
// stream (tests/test_gpu_head.py), but with three steps in flight on three streams 5-27 of 64 images per step got a gate computed from
// sums that were not yet the ones the other workgroups had published - whatever the publication used (write-through stores + sc1
// loads, returning atomics at agent scope, at system scope; tools/plan_diff.py located it: the sums buffers end up right, the gates
// differ in the fourth digit).  In a single stream the workgroups of one image run on ONE XCD (yr_xcd_swizzle) and share its L2;
// beside other kernels the dispatcher spreads them over XCDs whose L2s are coherent only through a full write-back / invalidate
// (__threadfence: measured 5 x the kernel's time with a fence per workgroup).  Default plans keep the SE_FC launch - made 2-3 x
// faster by the FC pair below - and the kernel boundary as the only cross-XCD hand-over.
//
// The SE block (reference code/yolo3/efficientnet.py:406-438: Mean over H, W -> 1x1 + bias -> Swish -> 1x1 + bias ->
// sigmoid) needs the channel means of the COMPLETE depthwise map; the kernels that produce that map already leave per-workgroup
// channel sums (rows of a small float32 buffer).  Through round 4 a separate launch (se_fc_kernel, one workgroup per image)
// added the rows up and ran the two tiny FCs: 6 launches of 11-22 us in every detection head, 21-30 in the SE EfficientNets, at
// 0.003 of any pipe - a latency chain.  Here the workgroup that COMPLETES an image's rows does it on the spot:
//   * every workgroup publishes its sums with returning atomic exchanges (yr_st_agent: performed at the device's coherence point),
//     waits for their return (vmcnt(0) + the workgroup barrier) and then adds its share to the image's arrival counter (atomic);
//   * the one that brings the counter to `arrivals` (all others have arrived before it) resets the counter, reads all rows
//     with returning atomics as well (yr_ld_agent), and computes mean -> FC1 -> swish -> FC2 -> sigmoid in a
//     FIXED order - rows added in index order, hidden units and channels as sequential fma chains - so the gate does not depend
//     on which workgroup happens to be last (a batch still equals its images run one by one).
// What this deliberately does NOT use is __threadfence(): on gfx950 an agent-scope release is an L2 write-back (buffer_wbl2) of
// everything the XCD holds dirty - i.e. of the map the kernel is busy writing - and measured 5 x the kernel's time (dw_kernel on
// 52 x 52 x 128 @64: 46 -> 250 us with a fence per workgroup).  Only the sums need to be visible device-wide, so only they are
// published that way; the map reaches memory at the kernel boundary like every other output.
// 131 k multiply-adds per image at most (F = 512): a few microseconds on one CU, spread over as many CUs as there are images,
// under the tail of the producing kernel.  No grid-wide barrier (round 3 measured that dead end), no co-residency assumption.
//
// Counters: one unsigned per image, zero before the launch and zero again after it (yr_forward clears the plan's counters at the
// start of every pass as well; yr_op_run callers hand in zeroed memory once).
#pragma once
#include "yr_common.h"

#ifndef YR_SE_SCOPE
#define YR_SE_SCOPE __HIP_MEMORY_SCOPE_SYSTEM
#endif

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

static inline size_t yr_se_tail_floats(int C, int R, int nth = 256) { return (size_t)yr_round_up(C, 4) + (size_t)yr_round_up(R, 4) + 4 * (size_t)nth; }   // LDS floats the tail needs (== yr_se_fc_floats)
#define YR_SE_TAIL_LDS 4608   // floats of LDS the kernels with a tail set aside for it (C + R + 1024 <= 4608: every EfficientNet up to B6)

#ifdef __HIPCC__
typedef float se_f4 __attribute__((ext_vector_type(4)));
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

// The partial sums travel between workgroups (CUs, XCDs) as RETURNING ATOMICS: an atomic read-modify-write is performed at the
// device's coherence point and its returned value (waited for with vmcnt) proves it has been - no fence, no cache state involved.
// (Round 5 tried write-through stores + sc1 loads first: correct in a strictly serial stream, WRONG with three steps in flight
// on three streams - 5-27 images of 64 per step differed: a store acknowledged to the wave is not yet a store another XCD reads.)
__device__ __forceinline__ float yr_ld_agent(const float* p) {   // the current value at the coherence point (fetch-or with 0: bit-exact)
    return __uint_as_float(__hip_atomic_fetch_or(reinterpret_cast<unsigned*>(const_cast<float*>(p)), 0u, __ATOMIC_RELAXED, YR_SE_SCOPE));
}
__device__ __forceinline__ void yr_st_agent(float* p, float v) {  // exchange; the old value is consumed so that the instruction returns (and counts in vmcnt until it has)
    const unsigned old = __hip_atomic_exchange(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, YR_SE_SCOPE);
    asm volatile("" ::"v"(old));
}
__device__ __forceinline__ void yr_st_agent4(float* p, float x, float y, float z, float w) {
    yr_st_agent(p, x); yr_st_agent(p + 1, y); yr_st_agent(p + 2, z); yr_st_agent(p + 3, w);
}

// The store of a quad of partial sums: a plain 16-byte store when a later LAUNCH reads them (se_fc_kernel: the kernel boundary is the
// hand-over - and the only form that was right with three steps in flight, tools/inflight_diff.py), atomics only under a tail.
__device__ __forceinline__ void yr_st_sums4(float* p, const bool tail, float x, float y, float z, float w) {
    if (tail) yr_st_agent4(p, x, y, z, w);
    else *reinterpret_cast<se_f4*>(p) = (se_f4){x, y, z, w};
}

// The SE block's FC pair (efficientnet.py:419-434: 1x1 + bias -> Swish -> 1x1 + bias -> sigmoid) by NTH cooperating threads (64: one
// wave, no barriers; otherwise the whole workgroup), shared by the SE tail below and se_fc_kernel (elementwise.hip).  Cut for
// LATENCY - the launch / tail that runs it has nothing else to do: 16-byte weight loads, all NTH threads busy (a thread owns a quad
// of outputs and a SEGMENT of the inputs; the segments' partial sums meet in LDS in segment order), a handful of load batches per
// stage instead of one load per multiply-add (round 4's se_fc_kernel: one wave per hidden unit + butterflies, one thread per channel
// with scalar loads: 11-22 us per launch).  Fixed orders throughout: the gate is a function of the means alone.
struct SeFc {
    const float* w1;   // [ldc][R4]  (the Keras kernel [1,1,C,R], rows padded to R4 = round_up(R, 4))
    const float* w2;   // [R][ldc]
    const float* b1;   // [R4]
    const float* b2;   // [ldc]
    int C, R, ldc;
};
__host__ __device__ static inline size_t yr_se_fc_floats(int C, int R, int nth) { return (size_t)((C + 3) & ~3) + (size_t)((R + 3) & ~3) + 4 * (size_t)nth; }   // LDS: mean | hid | partial quads

// mean: [ldc] floats in LDS (pad channels zero), filled and visible to all threads (a barrier behind the writes); scratch: R4 + 4 NTH
// floats behind it; gate: [>= ldc] floats of this image.
// Every batch of weight loads is waited for COMPLETELY (vmcnt(0)) before its first use.  Round 5 measured why: the plain loop
// (#pragma unroll 8; the compiler rotates the 16-byte loads through registers and consumes them under partial waits, vmcnt(5) ..
// vmcnt(0)) gave a WRONG hidden unit now and then - one register of one quarter wave (lanes 48-63) stale - but only beside other
// kernels of the plan on other streams (mbe / mbr / the walking head block / split pointwise: tools/sefc_probe2.py, 60-80 % of the
// launches differ; beside GEMMs or alone: none).  Not a race in the code (LDS pre-zeroed, canaries, atomics for the sums: no change);
// one load in flight, or a full wait per batch: 0 of 384.  This is what made the gates of the 512-channel head blocks differ in the
// fourth digit with three steps in flight - and what round 5 first blamed on the SE tail's cross-XCD publication.
__device__ __forceinline__ void yr_se_wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Segments a stage cuts its n inputs into: as many as give a thread two batches of 8 rows, not as many as there are threads.  The
// segments' partial sums meet in LDS one after the other (fixed order), so every segment is a dependent LDS read of the thread that
// owns the output: with NTH / quads segments a squeeze-excite block of 4 .. 48 hidden units (EfficientNet: 1 .. 12 quads) summed
// 85 .. 1024 partials serially - 3 .. 27 us of a launch that should be three round trips (round 5, found at batch 1: 21 launches of
// 12 us were 27 % of an EfficientNet-B0 pass).  The head blocks (512 channels, 128 hidden units: 32 and 8 segments) are unchanged.
__device__ __forceinline__ int yr_se_segments(const int n, const int most) {
    const int want = (n + 15) >> 4;
    return want < 1 ? 1 : (want < most ? want : most);
}

template <int NTH>
__device__ __forceinline__ void yr_se_fc_pair(const SeFc& f, const float* mean, float* scratch, float* gate, const int tid) {
    const int R4 = (f.R + 3) & ~3, QP = f.ldc >> 2, JQ = R4 >> 2;
    float* hid = scratch;                                  // [R4]
    se_f4* part = reinterpret_cast<se_f4*>(scratch + R4);  // [NTH] quads
    auto sync = [&]() { if constexpr (NTH > 64) __syncthreads(); };
    const se_f4 zero = (se_f4){0.f, 0.f, 0.f, 0.f};
    // ---- FC1 + swish: a thread = (channel segment, quad of hidden units)
    for (int j0 = 0; j0 < JQ; j0 += NTH) {
        const int jq = JQ - j0 < NTH ? JQ - j0 : NTH;
        const int nseg = yr_se_segments(f.C, NTH / jq), cps = (f.C + nseg - 1) / nseg;
        const int j = tid % jq, seg = tid / jq;
        se_f4 s = zero;
        if (seg < nseg) {
            const int ca = seg * cps, cb = ca + cps < f.C ? ca + cps : f.C;
            const float* wp = f.w1 + 4 * (j0 + j);
            for (int c = ca; c < cb; c += 8) {      // batches of 8 rows: all loads issued, ALL waited for, then used (see yr_se_wait_loads)
                se_f4 w8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w8[u] = *reinterpret_cast<const se_f4*>(wp + (size_t)(c + u < cb ? c + u : cb - 1) * R4);
                yr_se_wait_loads();
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float m = c + u < cb ? mean[c + u] : 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) s[i] = __builtin_fmaf(w8[u][i], m, s[i]);
                }
            }
        }
        part[tid] = s;
        sync();
        if (tid < jq) {
            se_f4 v = *reinterpret_cast<const se_f4*>(f.b1 + 4 * (j0 + tid));
            for (int sg = 0; sg < nseg; ++sg) v += part[sg * jq + tid];
#pragma unroll
            for (int i = 0; i < 4; ++i) hid[4 * (j0 + tid) + i] = 4 * (j0 + tid) + i < f.R ? yr_apply_act(v[i], YR_ACT_SWISH) : 0.f;
        }
        sync();
    }
    // ---- FC2 + sigmoid: a thread = (hidden-unit segment, channel quad)
    for (int q0 = 0; q0 < QP; q0 += NTH) {
        const int qp = QP - q0 < NTH ? QP - q0 : NTH;
        const int nseg = yr_se_segments(f.R, NTH / qp), jps = (f.R + nseg - 1) / nseg;
        const int q = tid % qp, seg = tid / qp;
        se_f4 s = zero;
        if (seg < nseg) {
            const int ja = seg * jps, jb = ja + jps < f.R ? ja + jps : f.R;
            const float* wp = f.w2 + 4 * (q0 + q);
            for (int j = ja; j < jb; j += 8) {
                se_f4 w8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w8[u] = *reinterpret_cast<const se_f4*>(wp + (size_t)(j + u < jb ? j + u : jb - 1) * f.ldc);
                yr_se_wait_loads();
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float hj = j + u < jb ? hid[j + u] : 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) s[i] = __builtin_fmaf(hj, w8[u][i], s[i]);
                }
            }
        }
        part[tid] = s;
        sync();
        if (tid < qp) {
            se_f4 v = *reinterpret_cast<const se_f4*>(f.b2 + 4 * (q0 + tid));
            for (int sg = 0; sg < nseg; ++sg) v += part[sg * qp + tid];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = 4 * (q0 + tid) + i < f.C ? yr_sigmoid(v[i]) : 0.f;
            *reinterpret_cast<se_f4*>(gate + 4 * (q0 + tid)) = v;
        }
        sync();
    }
}

// The channel means of image b from rows of partial sums [rows][ld] (a thread owns a channel and a segment of the rows; the segments
// meet in LDS in order), into mean[ldc].  ATOMIC: read every float with a returning atomic (the SE tail: the rows were published by
// other workgroups of the same launch); otherwise plain loads (a separate launch: se_fc_kernel).  scratch: NTH floats.
template <int NTH, bool ATOMIC>
__device__ __forceinline__ void yr_se_mean_rows(const float* rows, const int nrows, const int ld, const int C, const int ldc, const float count, float* mean, float* scratch, const int tid) {
    auto sync = [&]() { if constexpr (NTH > 64) __syncthreads(); };
    for (int c0 = 0; c0 < ldc; c0 += NTH) {       // (one pass unless C > NTH)
        const int cn = ldc - c0 < NTH ? ldc - c0 : NTH;          // channels of this pass
        const int nrs = NTH / cn, rps = (nrows + nrs - 1) / nrs;   // row segments, rows per segment
        const int c = tid % cn, seg = tid / cn;
        float s = 0.f;
        if (seg < nrs && c0 + c < C) {
            const int ra = seg * rps, rb = ra + rps < nrows ? ra + rps : nrows;
            if constexpr (ATOMIC) {
#pragma unroll 4
                for (int r = ra; r < rb; ++r) s += yr_ld_agent(rows + (size_t)r * ld + c0 + c);
            } else {
                // batches of 16 rows: all loads issued, all waited for, then added in row order (the maps of the first stages hand over
                // hundreds of rows: at 4 loads per round trip the pooling was most of the launch there)
                for (int r = ra; r < rb; r += 16) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = rows[(size_t)(r + u < rb ? r + u : rb - 1) * ld + c0 + c];
                    yr_se_wait_loads();
#pragma unroll
                    for (int u = 0; u < 16; ++u) if (r + u < rb) s += v[u];
                }
            }
        }
        scratch[tid] = s;
        sync();
        if (tid < cn) {
            float m = scratch[tid];
            for (int sg = 1; sg < nrs; ++sg) m += scratch[sg * cn + tid];
            mean[c0 + tid] = c0 + tid < C ? m / count : 0.f;
        }
        sync();
    }
}

template <int NTH>
__device__ __forceinline__ void yr_se_tail_fc(const SeTail& t, const int b, float* lds, const int tid) {
    const int R4 = (t.R + 3) & ~3;
    yr_se_mean_rows<NTH, true>(t.sums + (size_t)b * t.rows * t.ld_sums, t.rows, t.ld_sums, t.C, t.ldc, t.count, lds, lds + t.ldc + R4, tid);
    SeFc f;
    f.w1 = t.w; f.w2 = t.w + (size_t)t.ldc * R4; f.b1 = f.w2 + (size_t)t.R * t.ldc; f.b2 = f.b1 + R4;
    f.C = t.C; f.R = t.R; f.ldc = t.ldc;
    yr_se_fc_pair<NTH>(f, lds, lds + t.ldc, t.gate + (size_t)b * t.ld_gate, tid);
}

// Workgroup form: call by ALL NTH threads of the workgroup once its share of image b's sums is stored (uniform arguments); n: what it adds to the image's counter.
// flag: one LDS word; lds: yr_se_tail_floats() floats that are free by now.
template <int NTH>
__device__ __forceinline__ void yr_se_tail_arrive(const SeTail& t, const int b, const unsigned n, unsigned* flag, float* lds) {
    if (t.sums == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's write-through stores of the sums have been acknowledged
    __syncthreads();                                   // ... and every other thread's
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(t.sync + b, n, __ATOMIC_RELAXED, YR_SE_SCOPE);
        const bool last = old + n == t.arrivals;
        if (last) __hip_atomic_store(t.sync + b, 0u, __ATOMIC_RELAXED, YR_SE_SCOPE);   // (nobody else touches it before the next launch)
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
        const unsigned old = __hip_atomic_fetch_add(t.sync + b, n, __ATOMIC_RELAXED, YR_SE_SCOPE);
        last = old + n == t.arrivals ? 1u : 0u;
        if (last) __hip_atomic_store(t.sync + b, 0u, __ATOMIC_RELAXED, YR_SE_SCOPE);
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (last == 0u) return;
    yr_se_tail_fc<64>(t, b, lds, lane);
}
#endif
