// Plan runtime: owns the device copy of the parameter blob and the static op list produced
// by yoloret_amd.compiler; replays it on the caller's stream.  This is what stands in for
// tf.keras.Model.__call__ on the graph built by yolov3_body (reference
// code/yolo3/model.py:170-342, called at code/yolo.py:152 and code/yolo3/map.py:111).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <map>
#include <utility>
#include <vector>

#include "yr_common.h"

static thread_local char g_err[512] = "";

void yr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local const char* g_kernel = "";
void yr_note_kernel(const char* name) { g_kernel = name; }

extern "C" const char* yr_last_error(void) { return g_err; }
extern "C" int yr_abi_version(void) { return YR_ABI_VERSION; }
extern "C" int yr_abi_sizeof(int which) {
    return which == 0 ? (int)sizeof(yr_src) : which == 1 ? (int)sizeof(yr_op) : which == 2 ? (int)sizeof(yr_buf) : 0;
}

static int dispatch(const yr_op& op, int batch, hipStream_t s) {
    switch (op.kind) {
        case YR_OP_STEM: return yr_launch_stem(op, batch, s);
        case YR_OP_POINTWISE: return yr_launch_pointwise(op, batch, s);
        case YR_OP_DEPTHWISE: return yr_launch_depthwise(op, batch, s);
        case YR_OP_SE_MEAN: return yr_launch_se_mean(op, batch, s);
        case YR_OP_SE_FC: return yr_launch_se_fc(op, batch, s);
        case YR_OP_WSUM: return yr_launch_wsum(op, batch, s);
        case YR_OP_GATHER: return yr_launch_gather(op, batch, s);
        case YR_OP_MBCONV: yr_set_error("YR_OP_MBCONV (8) was removed in ABI 5: superseded by YR_OP_MBR / YR_OP_MBE (matrix pipe) and YR_OP_MBLANE"); return YR_ERR_ARG;
        case YR_OP_STEMBLOCK: return yr_launch_stemblock(op, batch, s);
        case YR_OP_MBLANE: return yr_launch_mblane(op, batch, s);
        case YR_OP_MBH: case YR_OP_MBX: return yr_launch_mbh(op, batch, s);
        case YR_OP_MBR: return (op.k & 0x40) ? yr_launch_mbk(op, batch, s) : yr_launch_mbr(op, batch, s);
        case YR_OP_MBE: return yr_launch_mbe(op, batch, s);
        case YR_OP_HEAD: return yr_launch_head(op, batch, s);
        default: yr_set_error("unknown op kind %d", op.kind); return YR_ERR_ARG;
    }
}

extern "C" int yr_op_run(const yr_op* op, int batch, void* stream) {
    YR_REQUIRE(op != nullptr && batch > 0, "yr_op_run: null op or empty batch");
    return dispatch(*op, batch, (hipStream_t)stream);
}

struct yr_handle {
    std::vector<yr_op> ops;
    std::vector<yr_buf> bufs;
    int64_t arena_per_image = 0;  // bytes
    std::vector<int> sync_slot;   // per op: index of its SE-tail arrival counters [batch] behind the arena, or -1
    int n_sync = 0;
    float* weights = nullptr;
    size_t n_weights = 0;
    std::map<int, std::vector<int>> tuned;  // batch -> per-op pointwise tile choice (1-based, 0 = heuristic)
    int device = 0;
    int32_t in_hw[2] = {0, 0};              // yr_create_from_blob only
    int32_t out_hwc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
};

extern "C" int yr_create(const yr_op* ops, int n_ops, const yr_buf* bufs, int n_bufs, yr_handle** out) {
    YR_REQUIRE(ops && bufs && out && n_ops > 0 && n_bufs > 0, "yr_create: bad arguments");
    yr_handle* h = new yr_handle();
    h->ops.assign(ops, ops + n_ops);
    h->bufs.assign(bufs, bufs + n_bufs);
    if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
    for (const yr_buf& b : h->bufs) {
        if (b.external_slot < 0) {
            if (b.arena_off_per_image < 0 || b.arena_off_per_image % 16 != 0 || b.bytes_per_image <= 0 || !yr_dtype_ok(b.dtype)) {
                delete h;
                yr_set_error("yr_create: arena buffer with bad offset/size/dtype");
                return YR_ERR_ARG;
            }
            const int64_t end = b.arena_off_per_image + b.bytes_per_image;
            if (end > h->arena_per_image) h->arena_per_image = end;
        } else if (b.external_slot > 3 || (b.dtype != YR_F32 && !(b.external_slot == 0 && b.dtype == YR_U8))) {   // (the image may be uint8)
            delete h;
            yr_set_error("yr_create: external slot %d out of range (or not float32)", b.external_slot);
            return YR_ERR_ARG;
        }
    }
    auto buf_ok = [&](int32_t b) { return b >= 0 && b < n_bufs; };
    for (const yr_op& op : h->ops) {
        bool ok = buf_ok(op.out_buf) && op.nsrc >= 1 && op.nsrc <= YR_MAX_SRC;
        for (int i = 0; ok && i < op.nsrc; ++i) ok = buf_ok(op.src[i].buf) && op.src[i].dtype == h->bufs[op.src[i].buf].dtype;
        if (ok) ok = op.out_dtype == h->bufs[op.out_buf].dtype;
        if (ok && op.res_buf >= 0) ok = buf_ok(op.res_buf);
        if (ok && op.gate_buf >= 0) ok = buf_ok(op.gate_buf);
        if (ok && op.gate_out_buf >= 0) ok = buf_ok(op.gate_out_buf) && h->bufs[op.gate_out_buf].dtype == YR_F32;
        if (!ok) {
            delete h;
            yr_set_error("yr_create: op references a buffer outside the table, or with a dtype other than the buffer's");
            return YR_ERR_ARG;
        }
        // extents: what the op addresses in each buffer must lie inside it (a corrupt or truncated plan must not turn into
        // out-of-bounds device accesses).  Sources: their own h x w x ld; output / residual: the op's rows x pitch.
        const char* why = nullptr;
        auto fits = [&](int32_t b, int64_t elems, int dtype) {
            const yr_buf& d = h->bufs[b];
            return elems > 0 && (d.external_slot >= 0 || elems * (int64_t)(dtype == YR_F32 ? 4 : dtype == YR_U8 ? 1 : 2) <= d.bytes_per_image);
        };
        if (op.h <= 0 || op.w <= 0 || op.h > (1 << 15) || op.w > (1 << 15) || op.out_ld <= 0 || op.out_ld > (1 << 20) || op.cout <= 0 || op.cin <= 0) why = "dims";
        for (int i = 0; !why && i < op.nsrc; ++i) {
            const yr_src& sr = op.src[i];
            if (sr.h <= 0 || sr.w <= 0 || sr.h > (1 << 15) || sr.w > (1 << 15) || sr.c <= 0 || sr.ld < sr.c || sr.ld > (1 << 20) ||
                !fits(sr.buf, (int64_t)sr.h * sr.w * sr.ld, sr.dtype)) why = "a source";
        }
        if (!why) {
            // rows the op writes: its h x w (already the pooled dims when a pointwise op stores the 2x2-pooled map); the
            // squeeze-excite ops write vectors; a depthwise / fused-block op with per-tile channel sums writes them to `gate`
            int64_t rows = (int64_t)op.h * op.w;
            if (op.kind == YR_OP_SE_MEAN || op.kind == YR_OP_SE_FC) rows = 1;
            if (op.out_ld < (op.kind == YR_OP_SE_MEAN || op.kind == YR_OP_SE_FC ? op.cin : op.cout) || !fits(op.out_buf, rows * op.out_ld, op.out_dtype)) why = "the output";
            else if (op.kind == YR_OP_HEAD && op.res_buf >= 0 && (op.res_ld < op.cin || !fits(op.res_buf, op.res_ld, YR_F32))) why = "the source's gate";   // (HEAD: res = the SE gate vector of its single source)
            else if (op.kind != YR_OP_HEAD && op.res_buf >= 0 && (op.res_ld < op.cout || !fits(op.res_buf, (int64_t)op.h * op.w * op.res_ld, op.dtype))) why = "the residual";
            else if (op.gate_buf >= 0 && op.gate_ld <= 0) why = "the gate";
            else if (op.kind == YR_OP_POINTWISE && op.gate_out_buf >= 0) {
                // the second output of a two-output conv (se_reduced bits 18 + 19): se_hidden couts, its own pooling in reserved0
                const int64_t ch = (int64_t)op.h * (op.stride == 2 ? 2 : 1), cw = (int64_t)op.w * (op.stride == 2 ? 2 : 1), p2 = ((op.reserved0 >> 8) & 1) ? 2 : 1;
                if ((op.se_reduced & 0xc0000) != 0xc0000 || op.dtype != YR_F32 || op.se_hidden < 1 || op.gate_out_ld < op.se_hidden || !fits(op.gate_out_buf, (ch / p2) * (cw / p2) * op.gate_out_ld, YR_F32))
                    why = "the second output";
            }
            else if (op.gate_out_buf >= 0 && (op.gate_buf < 0 || op.se_hidden < 1 || op.se_w_off < 0 || op.gate_out_ld < op.cout || !fits(op.gate_out_buf, op.gate_out_ld, YR_F32) ||
                                              op.se_reduced < 1 || !fits(op.gate_buf, (int64_t)op.se_reduced * op.gate_ld, YR_F32))) why = "the squeeze-excite tail";
        }
        if (why) {
            delete h;
            yr_set_error("yr_create: %s of an op (kind %d, %d x %d) does not fit the buffer it names", why, op.kind, op.h, op.w);
            return YR_ERR_ARG;
        }
    }
    h->sync_slot.assign(h->ops.size(), -1);
    for (size_t i = 0; i < h->ops.size(); ++i)
        if (h->ops[i].gate_out_buf >= 0 && h->ops[i].kind != YR_OP_POINTWISE) h->sync_slot[i] = h->n_sync++;      // (a POINTWISE op's gate_out is its second output)
    *out = h;
    return YR_OK;
}

// Header of a serialised plan (include/yoloret_hip.h: yr_create_from_blob).
struct yr_blob_header {
    char magic[8];
    uint32_t abi, n_ops, n_bufs, sizeof_op, sizeof_buf, n_tables;
    uint64_t n_weight_floats;
    int32_t in_h, in_w;
    int32_t out_hwc[9];
    int32_t reserved[3];
};
static_assert(sizeof(yr_blob_header) == 96, "serialised plan header is 96 bytes");

// Floats of the blob a parameter of `op` occupies from its offset on (0: the role is not used by this kind, or its size is
// not modelled here - the offset alone is then range-checked).  Mirrors the layouts documented in include/yoloret_hip.h.
static int64_t param_floats(const yr_op& op, int role) {   // role: 0 wgt, 1 scale, 2 shift, 3 wgt2, 4 b1, 5 b2, 6 se_w
    const int V = yr_vec_of(op.dtype);
    auto ru = [](int64_t v, int64_t m) { return (v + m - 1) / m * m; };
    if (role == 6) {   // the SE tail's FC pair: W1t [R][ldc] | W2 [R][ldc] | b1 [round_up(R, 4)] | b2 [ldc]
        if (op.se_hidden < 1) return -1;
        const int64_t ldc = ru(op.cout, 4), r4 = ru(op.se_hidden, 4);
        return ldc * r4 + (int64_t)op.se_hidden * ldc + r4 + ldc;
    }
    switch (op.kind) {
        case YR_OP_HEAD:
            if (role == 3) return (op.k & 0x40) ? (int64_t)(op.cout / 16) * 176 : 10 * ru(op.cout, 4);
            if (role == 0 && (op.k & 0xc0)) {   // float16 planes in fragment order, 32-channel chunks per source
                int64_t nk = 0;
                for (int i = 0; i < op.nsrc; ++i)
                    if (op.src[i].xform != YR_X_UP2_ADD) nk += (op.src[i].c + 31) / 32;
                return ru(op.cout, 16) / 16 * nk * (op.dtype == YR_F32 ? 512 : 256);   // (16-bit plans, walking form: one 16-bit fragment set, headwalk_h.hip)
            }
            [[fallthrough]];   // (the convolution's parameters as POINTWISE)
        case YR_OP_POINTWISE: {
            int64_t kp = 0;
            for (int i = 0; i < op.nsrc; ++i)
                if (op.src[i].xform != YR_X_UP2_ADD) kp += ru(op.src[i].c, V);
            if (op.kind == YR_OP_POINTWISE && op.dtype == YR_F32 && (op.se_reduced & 0x40000)) {     // the pixel-stationary form: float16 planes (pointwise_stream.hip)
                const int64_t tiles = ru(op.cout, 16) / 16 + ((op.se_reduced & 0x80000) ? ru(op.se_hidden, 16) / 16 : 0);      // (bit 19: the second output's tiles behind the first's)
                if (role == 0) return tiles * (int64_t)yr_pwt_chunks((int)kp) * 512;
                if ((role == 1 || role == 2) && (op.se_reduced & 0x80000)) return 16 * tiles;
            }
            if (role == 0) return op.dtype == YR_F32 ? (int64_t)op.cout * kp : ((int64_t)op.cout * kp + 1) / 2;
            if (role == 1 || role == 2) return op.cout;
            return 0;
        }
        case YR_OP_DEPTHWISE:
            if (role == 0) return (int64_t)op.k * op.k * ru(op.cin, V);
            if (role == 1 || role == 2) return ru(op.cin, V);
            return 0;
        case YR_OP_STEM:
            if (role == 0) return 27 * ru(op.cout, 4);
            if (role == 1 || role == 2) return ru(op.cout, 4);
            return 0;
        case YR_OP_SE_FC: {
            const int64_t ldc = ru(op.cin, 4), r4 = ru(op.se_reduced, 4);
            if (role == 0) return ldc * r4;
            if (role == 3) return (int64_t)op.se_reduced * ldc;
            if (role == 4) return r4;
            if (role == 5) return ldc;
            return 0;
        }
        case YR_OP_WSUM: return role == 0 ? 4 : 0;
        case YR_OP_MBH: case YR_OP_MBX: {
            const int64_t cexp = ru(op.kind == YR_OP_MBH ? op.se_reduced : op.cout, 32), kp = ru(op.cin, 32), kk = (op.k & 0xff) * (op.k & 0xff);
            if (role == 0) return cexp * kp / 2;
            if (role == 3) return (kk + 4) * cexp;
            if (op.kind == YR_OP_MBH && role == 4) return (int64_t)op.cout * cexp / 2;
            if (op.kind == YR_OP_MBH && role == 5) return 2 * ru(op.cout, 8);
            return 0;
        }
        case YR_OP_STEMBLOCK: {
            if (op.scale_off >= 0) {   // the matrix-pipe layout of the 16-bit plans (stemblock_h.hip)
                const int64_t c1p = ru(op.se_reduced, 32), cop = ru(op.cout, 16);
                const int64_t ext[6] = {c1p * 32 / 2, c1p, c1p, 10 * c1p, cop * c1p / 2, 2 * cop};
                return ext[role];
            }
            const int64_t cp = ru(op.se_reduced, 4) / 2, cop = ru(op.cout, 8);   // pair-packed, float32 pipe (stemblock.hip)
            if (role == 0) return cp * 58;
            if (role == 3) return cp * 22;
            if (role == 4) return 2 * cp * cop;
            if (role == 5) return 2 * cop;
            return 0;
        }
        case YR_OP_MBLANE: {
            const int64_t pr = ru((op.se_reduced + 1) / 2, 8), cinp = ru(op.cin, 4), cop = ru(op.cout, 8);
            if (role == 0) return pr * (cinp * 2 + 4);
            if (role == 3) return pr * 22;
            if (role == 4) return 2 * pr * cop;
            if (role == 5) return 2 * cop;
            return 0;
        }
        case YR_OP_MBE: {
            const int64_t t = op.cout / 16, ke = op.cin / 4;
            if (role == 0 && (op.k & 0x80)) return t * ((op.cin + 31) / 32) * 2 * 64 * 4;   // the split form: float16 planes
            if (role == 0) return t * ke * 64;
            if (role == 3) return t * 176;
            return 0;
        }
        case YR_OP_MBR: {
            const int64_t t = op.se_reduced / 16, to = ru(op.cout, 16) / 16, ke = op.cin / 4;
            if (op.k & 0x40) {   // the weight-streaming form (mbk.hip): one chunk per pair of expanded tiles, no wgt2
                if (role == 0) return (t + 1) / 2 * ((4 * ((op.cin + 31) / 32) + 2 * to) * 1024 + 2048) / 4;
                if (role == 5) return 16 * to;
                return 0;
            }
            if (role == 0 && (op.k & 0x80)) {   // the split form: float16 planes, projection fragments per tile pair of the nw waves
                const int64_t nw = (op.k >> 8) & 0xff, nke = (op.cin + 31) / 32;
                if (nw < 1 || nw > t) return -1;
                const int64_t ntl = t / nw, r = t % nw;
                const int64_t pairs = r * ((ntl + 2) / 2) + (nw - r) * ((ntl + 1) / 2);
                return (t * nke + pairs * to) * 2 * 64 * 4;
            }
            if (role == 0) return t * (ke + 4 * to) * 64;
            if (role == 3) return t * 176;
            if (role == 5) return 16 * to;
            return 0;
        }
        default: return 0;
    }
}

extern "C" int yr_create_from_blob(const void* blob, size_t bytes, yr_handle** out) {
    YR_REQUIRE(blob && out && bytes >= sizeof(yr_blob_header), "yr_create_from_blob: bad arguments");
    yr_blob_header hd;
    memcpy(&hd, blob, sizeof(hd));
    YR_REQUIRE(memcmp(hd.magic, "YRPLAN\0\0", 8) == 0, "yr_create_from_blob: not a serialised plan (bad magic)");
    YR_REQUIRE(hd.abi == YR_ABI_VERSION && hd.sizeof_op == sizeof(yr_op) && hd.sizeof_buf == sizeof(yr_buf),
               "yr_create_from_blob: plan written for ABI %u (yr_op %u bytes, yr_buf %u bytes); this library is ABI %d (%zu, %zu)",
               hd.abi, hd.sizeof_op, hd.sizeof_buf, YR_ABI_VERSION, sizeof(yr_op), sizeof(yr_buf));
    // every count is bounded by what `bytes` could hold BEFORE anything is multiplied: the sum below cannot wrap
    YR_REQUIRE(hd.n_ops > 0 && hd.n_bufs > 0 && hd.n_weight_floats > 0 && hd.n_ops <= bytes / sizeof(yr_op) && hd.n_bufs <= bytes / sizeof(yr_buf) &&
               hd.n_weight_floats <= bytes / sizeof(float) && hd.n_tables <= bytes / sizeof(int32_t),
               "yr_create_from_blob: the header's counts (%u ops, %u buffers, %llu weights, %u tables) exceed the %zu bytes given",
               hd.n_ops, hd.n_bufs, (unsigned long long)hd.n_weight_floats, hd.n_tables, bytes);
    const size_t need = sizeof(hd) + (size_t)hd.n_ops * sizeof(yr_op) + (size_t)hd.n_bufs * sizeof(yr_buf) +
                        (size_t)hd.n_weight_floats * sizeof(float) + (size_t)hd.n_tables * (1 + (size_t)hd.n_ops) * sizeof(int32_t);
    YR_REQUIRE(bytes == need, "yr_create_from_blob: %zu bytes, the header describes %zu", bytes, need);
    yr_handle* h = nullptr;
    try {   // no exception may cross the C boundary (std::bad_alloc / std::length_error from the vectors below)
        const char* p = static_cast<const char*>(blob) + sizeof(hd);
        std::vector<yr_op> ops(hd.n_ops);
        std::vector<yr_buf> bufs(hd.n_bufs);
        memcpy(ops.data(), p, ops.size() * sizeof(yr_op));
        p += ops.size() * sizeof(yr_op);
        memcpy(bufs.data(), p, bufs.size() * sizeof(yr_buf));
        p += bufs.size() * sizeof(yr_buf);
        for (yr_op& op : ops) {   // a file must not smuggle pointers in
            for (int i = 0; i < YR_MAX_SRC; ++i) op.src[i].ptr = nullptr;
            op.out = nullptr; op.res = nullptr; op.gate = nullptr; op.gate_out = nullptr; op.sync = nullptr;
            op.wgt = op.scale = op.shift = op.wgt2 = op.b1 = op.b2 = op.se_w = nullptr;
            YR_REQUIRE(op.nsrc >= 1 && op.nsrc <= YR_MAX_SRC && yr_dtype_ok(op.dtype), "yr_create_from_blob: op with %d sources / dtype %d", op.nsrc, op.dtype);
            const int64_t offs[7] = {op.wgt_off, op.scale_off, op.shift_off, op.wgt2_off, op.b1_off, op.b2_off, op.se_w_off};
            for (int role = 0; role < 7; ++role) {
                if (offs[role] < 0) continue;   // role not used
                const int64_t ext = param_floats(op, role);
                YR_REQUIRE(ext >= 0 && offs[role] < (int64_t)hd.n_weight_floats && ext <= (int64_t)hd.n_weight_floats - offs[role],
                           "yr_create_from_blob: a parameter of an op (kind %d, role %d: %lld floats at %lld) lies outside the weight blob of %llu",
                           op.kind, role, (long long)ext, (long long)offs[role], (unsigned long long)hd.n_weight_floats);
            }
        }
        int rc = yr_create(ops.data(), (int)ops.size(), bufs.data(), (int)bufs.size(), &h);
        if (rc) return rc;
        std::vector<float> w(hd.n_weight_floats);
        memcpy(w.data(), p, w.size() * sizeof(float));
        p += w.size() * sizeof(float);
        rc = yr_load_weights(h, w.data(), w.size());
        for (uint32_t t = 0; rc == YR_OK && t < hd.n_tables; ++t) {
            std::vector<int32_t> tab(1 + hd.n_ops);
            memcpy(tab.data(), p, tab.size() * sizeof(int32_t));
            p += tab.size() * sizeof(int32_t);
            rc = yr_set_tuning(h, tab[0], tab.data() + 1, (int)hd.n_ops);
        }
        if (rc) { yr_destroy(h); return rc; }
    } catch (const std::exception& e) {
        if (h) yr_destroy(h);
        yr_set_error("yr_create_from_blob: %s", e.what());
        return YR_ERR_ARG;
    }
    h->in_hw[0] = hd.in_h; h->in_hw[1] = hd.in_w;
    memcpy(h->out_hwc, hd.out_hwc, sizeof(h->out_hwc));
    *out = h;
    return YR_OK;
}

extern "C" int yr_plan_io_dims(const yr_handle* h, int32_t* in_hw, int32_t* out_hwc) {
    YR_REQUIRE(h && in_hw && out_hwc, "yr_plan_io_dims: null argument");
    if (h->in_hw[0] == 0) { yr_set_error("yr_plan_io_dims: the handle was not created from a serialised plan"); return YR_ERR_STATE; }
    memcpy(in_hw, h->in_hw, sizeof(h->in_hw));
    memcpy(out_hwc, h->out_hwc, sizeof(h->out_hwc));
    return YR_OK;
}

extern "C" void yr_destroy(yr_handle* h) {
    if (!h) return;
    if (h->weights) (void)hipFree(h->weights);
    delete h;
}

extern "C" int yr_load_weights(yr_handle* h, const float* host_blob, size_t n_floats) {
    YR_REQUIRE(h && host_blob && n_floats > 0, "yr_load_weights: bad arguments");
    if (h->weights) { (void)hipFree(h->weights); h->weights = nullptr; }
    YR_CHECK_HIP(hipMalloc((void**)&h->weights, n_floats * sizeof(float)));
    YR_CHECK_HIP(hipMemcpy(h->weights, host_blob, n_floats * sizeof(float), hipMemcpyHostToDevice));
    h->n_weights = n_floats;
    return YR_OK;
}

// the SE-tail arrival counters sit behind the arena: [n_sync][batch] words
static size_t yr_sync_offset(const yr_handle* h, int batch) { return ((size_t)h->arena_per_image * (size_t)batch + 15) & ~(size_t)15; }

extern "C" size_t yr_workspace_bytes(const yr_handle* h, int batch) {
    if (!h || batch <= 0) return 0;
    return yr_sync_offset(h, batch) + (size_t)h->n_sync * (size_t)batch * sizeof(uint32_t);
}

extern "C" int yr_plan_num_launches(const yr_handle* h) { return h ? (int)h->ops.size() : 0; }

extern "C" int yr_get_tuning(const yr_handle* h, int batch, int32_t* cfg, int n) {
    YR_REQUIRE(h && cfg && n == (int)h->ops.size(), "yr_get_tuning: bad arguments (n must equal yr_plan_num_launches)");
    auto it = h->tuned.find(batch);
    if (it == h->tuned.end()) { yr_set_error("yr_get_tuning: batch %d has not been tuned", batch); return YR_ERR_STATE; }
    for (int i = 0; i < n; ++i) cfg[i] = it->second[i];
    return YR_OK;
}

extern "C" int yr_set_tuning(yr_handle* h, int batch, const int32_t* cfg, int n) {
    YR_REQUIRE(h && cfg && batch > 0 && n == (int)h->ops.size(), "yr_set_tuning: bad arguments (n must equal yr_plan_num_launches)");
    std::vector<int> t(n, 0);
    for (int i = 0; i < n; ++i) {
        const int ncfg = yr_pointwise_num_cfgs(h->ops[i].dtype);
        const bool pw_ok = cfg[i] >= 0 && cfg[i] <= ncfg && (cfg[i] == 0 || h->ops[i].kind == YR_OP_POINTWISE);
        const bool mbh_ok = (h->ops[i].kind == YR_OP_MBH || h->ops[i].kind == YR_OP_MBX || h->ops[i].kind == YR_OP_MBR || h->ops[i].kind == YR_OP_MBE) && cfg[i] >= 0 && (cfg[i] & 0xff) == 0 && cfg[i] < (1 << 24);   // th << 8 | tw << 16
        YR_REQUIRE(pw_ok || mbh_ok, "yr_set_tuning: entry %d = %d is not a valid tile shape for that op", i, cfg[i]);
        // a split-form block's fragments are packed for ONE nw: a table tuned for another form of the plan (YOLORET_MBR_SPLIT=0) is refused
        YR_REQUIRE(!(h->ops[i].kind == YR_OP_MBR && (h->ops[i].k & 0x40)) || cfg[i] == 0, "yr_set_tuning: entry %d: the weight-streaming block form has nothing to tune (must be 0)", i);
        const bool split_block = (h->ops[i].kind == YR_OP_MBR || h->ops[i].kind == YR_OP_MBE) && (h->ops[i].k & 0x80);
        YR_REQUIRE(!split_block || (cfg[i] & 0xff00) == 0 || (cfg[i] & 0xff00) == (h->ops[i].k & 0xff00),
                   "yr_set_tuning: entry %d asks for %d waves per workgroup, the split-form fragments of that op are packed for %d", i, (cfg[i] >> 8) & 0xff, (h->ops[i].k >> 8) & 0xff);
        t[i] = cfg[i];
    }
    h->tuned[batch] = t;
    return YR_OK;
}

static int resolve_op(const yr_handle* h, size_t i, int batch, float* const ext[4], char* ws, yr_op* out) {
    yr_op op = h->ops[i];
    auto bufptr = [&](int32_t b) -> void* {
        const yr_buf& d = h->bufs[b];
        if (d.external_slot >= 0) return ext[d.external_slot];
        return ws + (size_t)d.arena_off_per_image * (size_t)batch;
    };
    auto wptr = [&](int64_t off) -> const float* { return off >= 0 ? h->weights + off : nullptr; };
    for (int k = 0; k < op.nsrc; ++k) op.src[k].ptr = bufptr(op.src[k].buf);
    op.out = bufptr(op.out_buf);
    op.res = op.res_buf >= 0 ? bufptr(op.res_buf) : nullptr;
    op.gate = op.gate_buf >= 0 ? (const float*)bufptr(op.gate_buf) : nullptr;
    op.gate_out = op.gate_out_buf >= 0 ? (float*)bufptr(op.gate_out_buf) : nullptr;
    op.se_w = wptr(op.se_w_off);
    op.sync = h->sync_slot[i] >= 0 ? reinterpret_cast<uint32_t*>(ws + yr_sync_offset(h, batch)) + (size_t)h->sync_slot[i] * (size_t)batch : nullptr;
    op.wgt = wptr(op.wgt_off); op.scale = wptr(op.scale_off); op.shift = wptr(op.shift_off);
    op.wgt2 = wptr(op.wgt2_off); op.b1 = wptr(op.b1_off); op.b2 = wptr(op.b2_off);
    if (op.out == nullptr) { yr_set_error("op %zu writes a null external buffer", i); return YR_ERR_ARG; }
    if (op.kind == YR_OP_POINTWISE) {
        auto it = h->tuned.find(batch);
        op.k = it != h->tuned.end() ? it->second[i] : 0;
    } else if (op.kind == YR_OP_MBH || op.kind == YR_OP_MBX) {   // the tuned output tile (th << 8 | tw << 16) rides in the upper bytes of k
        auto it = h->tuned.find(batch);
        if (it != h->tuned.end()) op.k = (op.k & 0xff) | it->second[i];
    } else if (op.kind == YR_OP_MBR || op.kind == YR_OP_MBE) {   // (waves per workgroup << 8 | row segments << 16: the tuned walk geometry)
        auto it = h->tuned.find(batch);
        if (it != h->tuned.end() && it->second[i] != 0 && !(op.k & 0x40)) {   // (k bit 6, the weight-streaming form: its geometry is the plan's)
            // the SPLIT form's fragments are packed for the plan's nw (compiler.mbs_pack): only the row segments are tunable there
            if (op.k & 0x80) op.k = (op.k & 0xffff) | (it->second[i] & 0xff0000);
            else op.k = (op.k & 0xff) | it->second[i];
        }
    }
    *out = op;
    return YR_OK;
}

static int check_forward_args(yr_handle* h, const float* images, int batch, void* workspace, size_t workspace_bytes) {
    YR_REQUIRE(h && images && batch > 0, "yr_forward: bad arguments");
    if (!h->weights) { yr_set_error("yr_forward: weights not loaded"); return YR_ERR_STATE; }
    YR_REQUIRE(workspace_bytes >= yr_workspace_bytes(h, batch) && (workspace || h->arena_per_image == 0),
               "yr_forward: workspace too small (%zu < %zu)", workspace_bytes, yr_workspace_bytes(h, batch));
    YR_REQUIRE(((uintptr_t)workspace % 16) == 0, "yr_forward: workspace must be 16-byte aligned");
    return YR_OK;
}

// The arrival counters of the SE tails (se_tail.h) must be zero when a pass starts.  Every launch leaves them zero again, so this
// matters for a workspace's first pass (the memory is the caller's, its contents unknown) and after an aborted one - cleared on
// every pass all the same: one 4 * n_sync * batch byte fill on the stream.
static int clear_sync(const yr_handle* h, int batch, void* workspace, hipStream_t s) {
    if (h->n_sync == 0) return YR_OK;
#ifdef YR_DEBUG_HOOKS      // (python tools/relink.py runtime.hip -DYR_DEBUG_HOOKS: never in the shipped library)
    static const bool skip = getenv("YR_NO_SYNC_CLEAR") != nullptr;
    if (skip) return YR_OK;
#endif
    YR_CHECK_HIP(hipMemsetAsync(static_cast<char*>(workspace) + yr_sync_offset(h, batch), 0, (size_t)h->n_sync * (size_t)batch * sizeof(uint32_t), s));
    return YR_OK;
}

static int fail_op(size_t i, int kind, int rc) {
    char tmp[400];
    strncpy(tmp, g_err, sizeof(tmp) - 1);
    tmp[sizeof(tmp) - 1] = 0;
    yr_set_error("op %zu (kind %d): %s", i, kind, tmp);
    return rc;
}

extern "C" int yr_forward(yr_handle* h, const float* images, int batch, float* y1, float* y2, float* y3,
                          void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_forward_args(h, images, batch, workspace, workspace_bytes);
    if (rc) return rc;
    float* ext[4] = {const_cast<float*>(images), y1, y2, y3};
    hipStream_t s = (hipStream_t)stream;
    rc = clear_sync(h, batch, workspace, s);
    if (rc) return rc;
#ifdef YR_DEBUG_HOOKS      // YR_ONLY_OPS="lo-hi" launches just those ops of the plan (tools/sefc_probe2.py); a malformed value is refused
    static const char* only = getenv("YR_ONLY_OPS");
    static int lo = 0, hi = 1 << 30;
    static const bool parsed = !only || (sscanf(only, "%d-%d", &lo, &hi) == 2 && lo >= 0 && hi >= lo);
    YR_REQUIRE(parsed, "YR_ONLY_OPS=%s: want lo-hi", only);
#endif
    for (size_t i = 0; i < h->ops.size(); ++i) {
#ifdef YR_DEBUG_HOOKS
        if ((int)i < lo || (int)i > hi) continue;
#endif
        yr_op op;
        rc = resolve_op(h, i, batch, ext, static_cast<char*>(workspace), &op);
        if (rc == YR_OK) rc = dispatch(op, batch, s);
        if (rc != YR_OK) return fail_op(i, h->ops[i].kind, rc);
    }
    return YR_OK;
}

// The forward pass with the largest |value| every op READS (its float32 k-space sources, a gated source's SE gate excluded):
// max_abs_per_op[n_ops] on the host, +inf where a NaN was seen, 0 for ops without float32 sources.  What a float32 plan's SPLIT-form
// ops need to know (their operands travel as two float16 planes: |x| must stay below 65504 - yoloret_amd Model.check_ranges runs this
// once per set of weights and moves the ops beyond the bound to the float32-MFMA forms).  Synchronises the stream.
extern "C" int yr_forward_ranges(yr_handle* h, const float* images, int batch, float* y1, float* y2, float* y3,
                                 void* workspace, size_t workspace_bytes, void* stream, float* max_abs_per_op) {
    int rc = check_forward_args(h, images, batch, workspace, workspace_bytes);
    if (rc) return rc;
    YR_REQUIRE(max_abs_per_op, "yr_forward_ranges: null result array");
    float* ext[4] = {const_cast<float*>(images), y1, y2, y3};
    hipStream_t s = (hipStream_t)stream;
    const size_t n = h->ops.size();
    unsigned* dmax = nullptr;
    YR_CHECK_HIP(hipMalloc((void**)&dmax, n * sizeof(unsigned)));
    rc = hipMemsetAsync(dmax, 0, n * sizeof(unsigned), s) == hipSuccess ? YR_OK : YR_ERR_HIP;
    if (rc == YR_OK) rc = clear_sync(h, batch, workspace, s);
    for (size_t i = 0; i < n && rc == YR_OK; ++i) {
        yr_op op;
        rc = resolve_op(h, i, batch, ext, static_cast<char*>(workspace), &op);
        for (int k = 0; rc == YR_OK && k < op.nsrc; ++k) {
            const yr_src& sr = op.src[k];
            if (sr.dtype != YR_F32 || sr.ptr == nullptr || (i == 0 && h->bufs[h->ops[i].src[k].buf].external_slot == 0 && h->bufs[h->ops[i].src[k].buf].dtype != YR_F32)) continue;
            rc = yr_launch_absmax((const float*)sr.ptr, (long long)batch * sr.h * sr.w, sr.c, sr.ld, dmax + i, s);
        }
        if (rc == YR_OK) rc = dispatch(op, batch, s);
        if (rc != YR_OK) rc = fail_op(i, h->ops[i].kind, rc);
    }
    if (rc == YR_OK) {
        std::vector<unsigned> bits(n);
        rc = hipMemcpyAsync(bits.data(), dmax, n * sizeof(unsigned), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess ? YR_OK : YR_ERR_HIP;
        if (rc == YR_OK)
            for (size_t i = 0; i < n; ++i) memcpy(&max_abs_per_op[i], &bits[i], sizeof(float));
        else yr_set_error("yr_forward_ranges: copying the result failed");
    }
    (void)hipFree(dmax);
    return rc;
}

// Same replay with a hipEvent pair around every op, `iters` times; ms_per_op[i] receives the
// average duration of op i and kernel_names[i] (if non-null) a pointer to a static string naming
// the kernel symbol it dispatched to.  Synchronises the stream; for measurement only.
extern "C" int yr_forward_profile(yr_handle* h, const float* images, int batch, float* y1, float* y2, float* y3,
                                  void* workspace, size_t workspace_bytes, void* stream, int iters,
                                  float* ms_per_op, const char** kernel_names) {
    int rc = check_forward_args(h, images, batch, workspace, workspace_bytes);
    if (rc) return rc;
    YR_REQUIRE(iters > 0 && ms_per_op, "yr_forward_profile: bad arguments");
    float* ext[4] = {const_cast<float*>(images), y1, y2, y3};
    hipStream_t s = (hipStream_t)stream;
    const size_t n = h->ops.size();
    std::vector<hipEvent_t> ev(n + 1);
    for (auto& e : ev) YR_CHECK_HIP(hipEventCreate(&e));
    std::vector<double> acc(n, 0.0);
    for (int it = 0; it < iters && rc == YR_OK; ++it) {
        rc = clear_sync(h, batch, workspace, s);
        if (rc) break;
        YR_CHECK_HIP(hipEventRecord(ev[0], s));
        for (size_t i = 0; i < n; ++i) {
            yr_op op;
            rc = resolve_op(h, i, batch, ext, static_cast<char*>(workspace), &op);
            if (rc == YR_OK) rc = dispatch(op, batch, s);
            if (rc != YR_OK) { rc = fail_op(i, h->ops[i].kind, rc); break; }
            if (kernel_names) kernel_names[i] = g_kernel;
            YR_CHECK_HIP(hipEventRecord(ev[i + 1], s));
        }
        if (rc != YR_OK) break;
        YR_CHECK_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < n; ++i) {
            float ms = 0.f;
            YR_CHECK_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            acc[i] += ms;
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (rc == YR_OK)
        for (size_t i = 0; i < n; ++i) ms_per_op[i] = (float)(acc[i] / iters);
    return rc;
}

// Does what op `i` writes (output, squeeze-excite sums / gate) overlap, in the arena, anything op `j` reads?  (Arena offsets are per
// image and scale with the batch alike: the per-image intervals decide.)
static bool output_aliases_inputs_of(const yr_handle* h, size_t i, size_t j) {
    auto span = [&](int32_t b, int64_t* lo, int64_t* hi) {
        if (b < 0 || h->bufs[b].external_slot >= 0) return false;
        *lo = h->bufs[b].arena_off_per_image; *hi = *lo + h->bufs[b].bytes_per_image;
        return true;
    };
    const yr_op& w = h->ops[i];
    const yr_op& r = h->ops[j];
    const int32_t writes[3] = {w.out_buf, (w.kind == YR_OP_SE_FC || w.kind == YR_OP_POINTWISE) ? -1 : w.gate_buf, w.gate_out_buf};
    int32_t reads[YR_MAX_SRC + 2];
    int nr = 0;
    for (int k = 0; k < r.nsrc; ++k) reads[nr++] = r.src[k].buf;
    reads[nr++] = r.res_buf;
    reads[nr++] = (r.kind == YR_OP_POINTWISE) ? r.gate_buf : -1;
    for (int32_t wb : writes)
        for (int k = 0; k < nr; ++k) {
            int64_t a0, a1, b0, b1;
            if (span(wb, &a0, &a1) && span(reads[k], &b0, &b1) && a0 < b1 && b0 < a1) return true;
        }
    return false;
}

// Per-op tile autotuning for one batch size: runs the forward once (so every buffer holds real data), then
// times every pointwise tile shape on every pointwise op (hipEvents, `iters` launches each) and remembers
// the fastest; later yr_forward calls with the same batch use those shapes.  Results do not depend on the
// tile shape (the k-summation order is the same for every shape), only the speed does.  Synchronises.
extern "C" int yr_autotune(yr_handle* h, const float* images, int batch, float* y1, float* y2, float* y3,
                           void* workspace, size_t workspace_bytes, void* stream, int iters) {
    int rc = check_forward_args(h, images, batch, workspace, workspace_bytes);
    if (rc) return rc;
    YR_REQUIRE(iters > 0, "yr_autotune: iters must be positive");
    h->tuned.erase(batch);
    rc = yr_forward(h, images, batch, y1, y2, y3, workspace, workspace_bytes, stream);
    if (rc) return rc;
    float* ext[4] = {const_cast<float*>(images), y1, y2, y3};
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    YR_CHECK_HIP(hipEventCreate(&e0));
    YR_CHECK_HIP(hipEventCreate(&e1));
    std::vector<int> best(h->ops.size(), 0);
    for (size_t i = 0; i < h->ops.size() && rc == YR_OK; ++i) {
        if (h->ops[i].kind == YR_OP_MBH || h->ops[i].kind == YR_OP_MBX) {
            // fused 16-bit block: time a fixed list of output tiles (runs of 4 along x: tw % 4 == 0); tiles the op
            // cannot take (too many pixels for its accumulators, LDS footprint) are refused by the launcher and skipped
            static const int tiles[][2] = {{4, 8}, {8, 4}, {7, 4}, {7, 8}, {8, 8}, {13, 4}, {4, 16}, {8, 16}, {7, 16}, {13, 8},
                                           {16, 8}, {13, 16}, {8, 12}, {7, 12}, {13, 12}, {16, 12}, {16, 16}, {4, 12}, {6, 8}};
            // ops that run the row-walking register-chained form (mbxr_h.hip; chosen by shape): 1 .. 6 row segments per strip
            static const int chained[][2] = {{255, 1}, {255, 2}, {255, 3}, {255, 4}, {255, 6}};
            yr_op op;
            rc = resolve_op(h, i, batch, ext, static_cast<char*>(workspace), &op);
            if (rc) break;
            const int kk = op.k & 0xff;
            float best_ms = 1e30f;
            int best_cfg = 0;
            const bool ch = yr_mbh_prefers_chained(op);
            const int (*cand)[2] = ch ? chained : tiles;
            const int ncand = ch ? (int)(sizeof(chained) / sizeof(chained[0])) : (int)(sizeof(tiles) / sizeof(tiles[0]));
            for (int pass = 0; pass < 2; ++pass)                 // two passes, minimum: one noisy sample must not decide
                for (int c = -1; c < ncand; ++c) {
                    const int cfg = c < 0 ? 0 : (cand[c][0] << 8) | (cand[c][1] << 16);
                    op.k = kk | cfg;
                    if (dispatch(op, batch, s) != YR_OK) continue;          // warm-up / validity
                    YR_CHECK_HIP(hipEventRecord(e0, s));
                    for (int it = 0; it < iters; ++it) (void)dispatch(op, batch, s);
                    YR_CHECK_HIP(hipEventRecord(e1, s));
                    YR_CHECK_HIP(hipEventSynchronize(e1));
                    float ms = 0.f;
                    YR_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best_ms * 0.98f) { best_ms = ms; best_cfg = cfg; }
                }
            best[i] = best_cfg;
            continue;
        }
        if (h->ops[i].kind == YR_OP_MBR && (h->ops[i].k & 0x40)) continue;   // the weight-streaming form (mbk.hip): nothing to tune
        if (h->ops[i].kind == YR_OP_MBR || h->ops[i].kind == YR_OP_MBE) {
            // register-chained float32 blocks: row segments per strip (how many waves the walk is cut into; each segment
            // recomputes two halo rows), IN CONTEXT - right behind the predecessor, whose output is what the caches hold (in
            // isolation block_2 measured 203 us, in the pipeline 240).  Nothing an op writes depends on the choice (the squeeze-excite sums of a
            // chained MBX op leave per quantum of output rows fixed by the map's shape; a segment is a whole number of quanta: mbxr_h.hip).
            yr_op op, prev;
            rc = resolve_op(h, i, batch, ext, static_cast<char*>(workspace), &op);
            if (rc) break;
            // ... unless op i's output shares arena memory with an input of the predecessor (their lifetimes do not overlap, so the
            // planner may have placed them on top of each other): re-running the predecessor would then read what op i just wrote
            bool have_prev = i > 0 && !output_aliases_inputs_of(h, i, i - 1);
            if (have_prev) {
                rc = resolve_op(h, i - 1, batch, ext, static_cast<char*>(workspace), &prev);
                if (rc) break;
                if (h->ops[i - 1].kind == YR_OP_POINTWISE) prev.k = best[i - 1];
                else if (prev.kind == YR_OP_MBR && (prev.k & 0x40)) {}
                else if (best[i - 1] != 0 && (prev.k & 0x80)) prev.k = (prev.k & 0xffff) | (best[i - 1] & 0xff0000);
                else if (best[i - 1] != 0) prev.k = (prev.k & 0xff) | best[i - 1];
            }
            static const int segs_list[] = {0, 1, 2, 3, 4, 6, 8, 13, 18, 26};   // (13 .. 26: the few-image passes, where a strip's walk is the whole launch)
            const int base = op.k & 0xffff;     // 3 | nw << 8: the workgroup shape the plan asks for
            float best_ms = 1e30f;
            int best_cfg = 0;
            for (int c = 0; c < (int)(sizeof(segs_list) / sizeof(segs_list[0])); ++c) {
                if (segs_list[c] > h->ops[i].h) continue;
                op.k = base | (segs_list[c] << 16);
                if (dispatch(op, batch, s) != YR_OK) continue;
                float ms = 1e30f;
                for (int it = 0; it < 2 * iters + 1; ++it) {
                    if (have_prev && dispatch(prev, batch, s) != YR_OK) break;
                    YR_CHECK_HIP(hipEventRecord(e0, s));
                    (void)dispatch(op, batch, s);
                    YR_CHECK_HIP(hipEventRecord(e1, s));
                    YR_CHECK_HIP(hipEventSynchronize(e1));
                    float t = 0.f;
                    YR_CHECK_HIP(hipEventElapsedTime(&t, e0, e1));
                    if (t < ms) ms = t;
                }
                if (ms < best_ms * 0.98f) { best_ms = ms; best_cfg = (base & 0xff00) | (segs_list[c] << 16); }
            }
            best[i] = best_cfg;
            continue;
        }
        if (h->ops[i].kind != YR_OP_POINTWISE) continue;
        if (h->ops[i].dtype == YR_F32 && (h->ops[i].se_reduced & 0x40000)) continue;   // the pixel-stationary form (pointwise_stream.hip): nothing to tune
        const int ncfg = yr_pointwise_num_cfgs(h->ops[i].dtype);
        yr_op op;
        rc = resolve_op(h, i, batch, ext, static_cast<char*>(workspace), &op);
        if (rc) break;
        auto time_cfg = [&](int c, float* ms) -> int {
            op.k = c;
            int r = dispatch(op, batch, s);                 // warm-up
            if (r) return r;
            YR_CHECK_HIP(hipEventRecord(e0, s));
            for (int it = 0; it < iters && r == YR_OK; ++it) r = dispatch(op, batch, s);
            YR_CHECK_HIP(hipEventRecord(e1, s));
            YR_CHECK_HIP(hipEventSynchronize(e1));
            YR_CHECK_HIP(hipEventElapsedTime(ms, e0, e1));
            return r;
        };
        // The launch as the forward pass sees it: right behind its predecessor (whose tail it overlaps and whose
        // output is what the caches hold), one launch per sample.  Back-to-back repeats of one op flatter some
        // shapes by up to 30 % (measured: td2_conv, block_7_expand).
        yr_op prev;
        bool have_prev = false;
        if (i > 0 && !output_aliases_inputs_of(h, i, i - 1)) {
            rc = resolve_op(h, i - 1, batch, ext, static_cast<char*>(workspace), &prev);
            if (rc) break;
            if (h->ops[i - 1].kind == YR_OP_POINTWISE) prev.k = best[i - 1];
            have_prev = true;
        }
        auto time_in_context = [&](int c, float* ms) -> int {
            op.k = c;
            *ms = 1e30f;
            for (int it = 0; it < iters; ++it) {
                int r = have_prev ? dispatch(prev, batch, s) : YR_OK;
                if (r) return r;
                YR_CHECK_HIP(hipEventRecord(e0, s));
                r = dispatch(op, batch, s);
                if (r) return r;
                YR_CHECK_HIP(hipEventRecord(e1, s));
                YR_CHECK_HIP(hipEventSynchronize(e1));
                float t = 0.f;
                YR_CHECK_HIP(hipEventElapsedTime(&t, e0, e1));
                if (t < *ms) *ms = t;
            }
            return YR_OK;
        };
        // round 1: every shape, repeated launches (c == 0: the heuristic's own pick); round 2: the FINALISTS in
        // context, interleaved, each scored by its minimum - one noisy sample must not crown (or sink) a shape
        constexpr int FINALISTS = 5, ROUNDS = 3;
        std::vector<std::pair<float, int>> first;
        for (int c = 0; c <= ncfg && rc == YR_OK; ++c) {
            float ms = 0.f;
            rc = time_cfg(c, &ms);
            first.push_back({ms, c});
        }
        if (rc == YR_OK) {
            std::stable_sort(first.begin(), first.end());
            if ((int)first.size() > FINALISTS) first.resize(FINALISTS);
            for (auto& f : first) f.first = 1e30f;
            for (int r = 0; r < ROUNDS && rc == YR_OK; ++r)
                for (auto& f : first) {
                    float ms = 0.f;
                    rc = time_in_context(f.second, &ms);
                    if (rc) break;
                    if (ms < f.first) f.first = ms;
                }
            float best_ms = 1e30f;
            for (auto& f : first)   // earlier in round-1 order wins ties within 1 %
                if (f.first < best_ms * 0.99f) { best_ms = f.first; best[i] = f.second; }
        }
        if (rc != YR_OK) rc = fail_op(i, h->ops[i].kind, rc);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc == YR_OK) h->tuned[batch] = best;
    return rc;
}
