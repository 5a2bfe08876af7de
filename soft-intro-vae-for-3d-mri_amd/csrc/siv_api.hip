// C-ABI entry points that dispatch between the convolution kernel families, plus the
// library-wide helpers (error string, version, workspace size, test overrides).
#include "siv_common.h"
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include <stdlib.h>
#include <string.h>

namespace siv {

static thread_local char g_err[512] = "";
static std::atomic<int> g_conv_path{0};
// kernel family the last 3x3x3 convolution entry point of this thread dispatched to (for tests)
static thread_local const char* g_conv_kernel = "";
#define SIV_PICK(name) (g_conv_kernel = (name))
// (a launcher that chooses between forms of its family names the one it took)
void note_conv_kernel(const char* name) { g_conv_kernel = name; }

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
// Path 0 ("automatic") resolves to the value of the environment variable SIV_CONV_PATH when that is
// set (e.g. SIV_CONV_PATH=5 runs a whole training job on the split-bf16 kernels), else to 0.
int conv_path() {
    const int forced = g_conv_path.load(std::memory_order_relaxed);
    if (forced != 0) return forced;
    static const int env_path = [] {
        const char* e = getenv("SIV_CONV_PATH");
        const int v = e ? atoi(e) : 0;
        return (v >= 0 && v <= 6) ? v : 0;
    }();
    return env_path;
}

// ---- packed-weight cache --------------------------------------------------------------------------
// Host-side bookkeeping only: the memory is donated by the caller (siv_pack_cache_configure), one cache
// per device.  Entries are keyed by (weight pointer, family, variant, flip, Cin, Cout) and live until
// siv_pack_cache_invalidate.  A slot is filled by the stream of the call that created it; later calls on
// the same stream read it in order.
namespace {
struct PackSlot {
    size_t off;
    bool valid;       // false: the weights changed since the slot was filled (the slot keeps its memory)
};
struct PackCache {
    char* base = nullptr;
    size_t size = 0, used = 0;
    std::vector<std::pair<const char*, const char*>> stable;      // [lo, hi) ranges of cacheable weights
    std::map<std::tuple<const void*, int, int, int, int, int>, PackSlot> slots;
    long hits = 0, misses = 0;
};
constexpr int kMaxDevices = 64;
PackCache g_pack[kMaxDevices];
std::mutex g_pack_mu;
PackCache* current_cache() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    return &g_pack[dev];
}
}  // namespace

void* pack_slot(const void* w, int family, int variant, int flip, int Cin, int Cout, size_t bytes, void* ws,
                bool* need_pack) {
    *need_pack = true;
    std::lock_guard<std::mutex> lock(g_pack_mu);
    PackCache* c = current_cache();
    if (!c || !c->base) return ws;
    bool stable = false;
    for (const auto& r : c->stable) stable = stable || ((const char*)w >= r.first && (const char*)w < r.second);
    if (!stable) return ws;
    const auto key = std::make_tuple(w, family, variant, flip, Cin, Cout);
    auto it = c->slots.find(key);
    if (it != c->slots.end()) {
        if (it->second.valid) {
            *need_pack = false;
            ++c->hits;
        } else {
            it->second.valid = true;       // refilled in place by the caller (same key = same size)
            ++c->misses;
        }
        return c->base + it->second.off;
    }
    const size_t need = align256(bytes);
    if (c->used + need > c->size) return ws;                      // cache full: pack into the workspace as before
    const size_t off = c->used;
    c->used += need;
    c->slots.emplace(key, PackSlot{off, true});
    ++c->misses;
    return c->base + off;
}

void pack_slot_failed(const void* slot) {
    std::lock_guard<std::mutex> lock(g_pack_mu);
    PackCache* c = current_cache();
    if (!c || !c->base) return;
    for (auto& kv : c->slots)
        if (c->base + kv.second.off == (const char*)slot) {
            kv.second.valid = false;
            return;
        }
}

// conv_generic.hip
int k3_fwd_generic_launch(const float*, const float*, const float*, float*, int, int, int, int, int,
                          int, int, hipStream_t);
int k3_dgrad_generic_launch(const float*, const float*, const float*, float*, int, int, int, int, int,
                            int, hipStream_t);
int k3_wgrad_generic_launch(const float*, const float*, float*, float*, int, int, int, int, int, int,
                            int, int, void*, size_t, hipStream_t);
int colsum_launch(const float*, float*, long, int, int, void*, size_t, hipStream_t);
// conv_mfma.hip
bool k3_fwd_mfma_supported(int Cin, int Cout);
int k3_fwd_mfma_launch(const float*, const float*, const float*, float*, int, int, int, int, int, int,
                       int, int, void*, size_t, hipStream_t);
bool k3_wgrad_mfma_supported(int Cin, int Cout);
int k3_wgrad_mfma_launch(const float*, const float*, float*, int, int, int, int, int, int, int, int,
                         void*, size_t, hipStream_t);
// conv_wino.hip
bool k3_fwd_wino_supported(int Cin, int Cout);
int k3_fwd_wino_launch(const float*, const float*, const float*, float*, int, int, int, int, int, int,
                       int, int, void*, size_t, hipStream_t, float* stat_part);
long k3_fwd_wino_bricks(int N, int D, int H, int W);
size_t k3_fwd_wino_ws_bytes(int Cin, int Cout);
// conv_wino2d.hip
bool k3_fwd_wino2d_supported(int Cin, int Cout);
size_t k3_fwd_wino2d_ws_bytes(int Cin, int Cout);
long k3_fwd_wino2d_bricks(int N, int D, int H, int W);
int k3_fwd_wino2d_launch(const float*, const float*, const float*, float*, int, int, int, int, int, int, int,
                         void*, size_t, hipStream_t, float*, void*, size_t);
long k3_fwd_wino2d_stat_rows(int N, int D, int H, int W, int Cin, int Cout);
size_t k3_fwd_wino2d_ksplit_bytes(int N, int D, int H, int W, int Cin, int Cout);
// conv_wgrad_wino.hip
bool k3_wgrad_wino2d_supported(int Cin, int Cout);
int k3_wgrad_wino2d_launch(const float*, const float*, float*, int, int, int, int, int, int, int, void*, size_t,
                           hipStream_t);
int k2_up_wgrad_wino_launch(const float*, const float*, float*, int, int, int, int, int, int, int, void*, size_t,
                            hipStream_t);
// conv_wino_b3.hip
bool k3_fwd_wino_b3_supported(int Cin, int Cout);
size_t k3_fwd_wino_b3_ws_bytes(int Cin, int Cout);
int k3_fwd_wino_b3_launch(const float*, const float*, const float*, float*, int, int, int, int, int, int, int,
                          void*, size_t, hipStream_t, float*);
// conv_up.hip
bool k2_up_fwd_supported(int Cin, int Cout);
size_t k2_up_fwd_ws_bytes(int Cin, int Cout);
long k2_up_fwd_blocks(int N, int D, int H, int W, bool split, bool force_prod);
int k2_up_fwd_launch(const float*, const float*, const float*, float*, int, int, int, int, int, int, void*,
                     size_t, hipStream_t, float*, bool, bool);
bool k2_up_dgrad_supported(int Cin, int Cout);
int k2_up_dgrad_launch(const float*, const float*, float*, int, int, int, int, int, int, void*, size_t,
                       hipStream_t, bool, bool);
bool k2_up_wgrad_supported(int Cin, int Cout);
int k2_up_wgrad_launch(const float*, const float*, float*, int, int, int, int, int, int, int, void*, size_t,
                       hipStream_t);
// bn_act.hip
int bn_finalize_launch(const float*, const float*, int, long, int, float*, float*, float*, float*,
                       int64_t*, float, hipStream_t);
// conv_thin.hip
bool k3_thin_supported(int Cin, int Cout);
long k3_thin_fwd_stat_blocks(int N, int D, int H, int W, int Ca, int Cb);
int k3_thin_fwd_launch(const float*, const float*, const float*, float*, int, int, int, int, int, int,
                       int flip, hipStream_t, float* stat_part);
int k3_thin_wgrad_launch(const float*, const float*, float*, float*, int, int, int, int, int, int,
                         int, void*, size_t, hipStream_t, const BnPrologue* bn, bool x_bf16 = false);
int k3_thin_bn_dgrad_launch(const float* g1, const float* w, const float* x, const BnPrologue& bn, float* dx,
                            float* dgamma, float* dbeta, int accumulate, int N, int D, int H, int W, int Cin,
                            void* ws, size_t ws_bytes, hipStream_t s, bool x_bf16 = false);
int stem_bnact_fwd_launch(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                          const uint8_t* keep, const PhiloxMask& rk, float p, float slope, float eps, int train,
                          float* mean, float* var, float* running_mean, float* running_var, int64_t* nbt,
                          float momentum, float* out, int N, int D, int H, int W, int C, void* ws, size_t ws_bytes,
                          hipStream_t s, void* out_bf16 = nullptr, void* bits = nullptr);
int stem_bnact_wgrad_launch(const float* dout, const float* x, const float* w, const float* bias, const float* mean,
                            const float* var, const float* gamma, const void* bits, float p, float slope, float eps,
                            float* dw, float* dbias, float* dgamma, float* dbeta, int accumulate, int N, int D,
                            int H, int W, int C, void* ws, size_t ws_bytes, hipStream_t s);
int stem_bnact_bwd_launch(const float* dout, const float* x, const float* w, const float* bias, const float* mean,
                          const float* var, const float* gamma, const float* beta, const uint8_t* keep,
                          const PhiloxMask& rk, float p, float slope, float eps, float* dy, float* dgamma,
                          float* dbeta, int accumulate, int N, int D, int H, int W, int C, void* ws, size_t ws_bytes,
                          hipStream_t s, const void* bits = nullptr);
bool k3_thin_bn_fwd_supported(int N, int D, int H, int W, int Cin);
int k3_thin_bn_fwd_launch(const float* x, const BnPrologue& bn, const float* w, const float* bias, float* out,
                          int N, int D, int H, int W, int Cin, hipStream_t s, void* x_bf16 = nullptr);

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static int check_conv_shape(const char* what, int N, int D, int H, int W, int Cin, int Cout, int ups) {
    SIV_CHECK_ARG(N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0,
                  "%s: extents must be positive (N=%d D=%d H=%d W=%d Cin=%d Cout=%d)", what, N, D, H,
                  W, Cin, Cout);
    SIV_CHECK_ARG(ups == 1 || ups == 2, "%s: ups must be 1 or 2 (got %d)", what, ups);
    SIV_CHECK_ARG(ups == 1 || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0),
                  "%s: ups=2 needs even output extents (got %d,%d,%d)", what, D, H, W);
    return SIV_OK;
}

// A volume is "big" when its 4x8x8 bricks times its 64-channel output tiles fill the chip twice over;
// the Winograd kernels (one block per brick and tile) only pay off from there.
static bool big_volume(int N, int D, int H, int W, int Cout) {
    return (long)N * ((D + 3) / 4) * ((H + 7) / 8) * ((W + 7) / 8) * (Cout / 64) >= 512;
}

// Winograd F(2,3) path: on request (path 3), or as the fallback of the automatic paths for big volumes
// that the two-dimensional kernel cannot take.  Path 4 = matrix-core kernels without Winograd.
static bool use_wino(int path, bool aligned, int N, int D, int H, int W, int Cin, int Cout) {
    if (!aligned || !k3_fwd_wino_supported(Cin, Cout)) return false;
    if (path == 3) return true;
    if (path != 0 && path < 5) return false;
    return big_volume(N, D, H, W, Cout);
}

// Two-dimensional Winograd F(2x2,3x3) kernel for the non-upsampled forward / input gradient: on request
// (path 6) and automatically (path 0) once the volume gives it enough blocks (measured 13-18 % faster than
// the one-dimensional kernel on big volumes; the 10- and 20-wide levels run on 6x10-voxel brick planes).
static bool use_w2d(int path, bool aligned, int ups, int N, int D, int H, int W, int Cin, int Cout) {
    if (ups != 1 || !aligned || !k3_fwd_wino2d_supported(Cin, Cout)) return false;
    if ((long)D * H * W * Cin >= (1L << 29)) return false;      // 32-bit byte offsets inside a sample
    // from 160 blocks (bricks of the geometry the kernel picks x 64-channel output tiles): measured 1.4x the
    // direct kernel at 192 blocks (256->256 @10x12x10, batch 8)
    return path == 6 || (path == 0 && k3_fwd_wino2d_bricks(N, D, H, W) * (Cout / 64) >= 160);
}

// Winograd F(3x3,2x2) weight gradient: on request (path 6) and automatically where it was measured
// faster than the direct kernel (1.9-2.2x from 10x12x10 upwards at batch 8): at least 128 column steps
// for its 256 persistent blocks, planes that fill at least 45 % of their columns
static bool use_gw(int path, bool aligned, int ups, int N, int D, int H, int W, int Cin, int Cout) {
    if (!aligned || !k3_wgrad_wino2d_supported(Cin, Cout)) return false;
    if ((long)(D + 3) * H * W * (Cin > Cout ? Cin : Cout) >= (1L << 29)) return false;   // 32-bit byte offsets inside a sample
    if (path == 6) return true;
    if (path != 0 && path != 5) return false;
    // columns of 8 x 8 voxels, or (non-upsampled kernel) of 6 x 10 where those pad less
    long nrh = (H + 7) / 8, nrw = (W + 7) / 8, colv = 64;
    if (ups == 1 && cdiv(H, 6) * cdiv(W, 10) * 60 * 16 < nrh * nrw * 64 * 15) nrh = cdiv(H, 6), nrw = cdiv(W, 10), colv = 60;
    return (long)N * nrh * nrw * (D / ups) >= 128 && 100L * H * W >= 45L * nrh * nrw * colv;
}

// Path 5: the Winograd kernel whose fp32 products are formed from bf16 triples on the bf16 matrix cores
// (non-upsampled forward / input gradient; everything else as path 0)
static bool use_b3(int path, bool aligned, int ups, int Cin, int Cout) {
    return path == 5 && ups == 1 && aligned && k3_fwd_wino_b3_supported(Cin, Cout);
}

// the collapsed-tap kernel for convolutions of an upsampled input (auto and "specialised" paths)
static bool use_up(int path, bool aligned, int ups, int Cin, int Cout) {
    return ups == 2 && (path == 0 || path == 2 || path >= 5) && aligned && k2_up_fwd_supported(Cin, Cout);
}

}  // namespace siv

using namespace siv;

extern "C" const char* siv_last_error(void) { return g_err; }
extern "C" const char* siv_last_conv_kernel(void) { return g_conv_kernel; }
extern "C" int siv_version(void) { return 107; }
extern "C" size_t siv_workspace_bytes(void) { return (size_t)512 << 20; }
extern "C" int siv_pack_cache_configure(void* mem, size_t bytes) {
    SIV_CHECK_ARG(((uintptr_t)mem & 255) == 0, "pack_cache_configure: memory must be 256-byte aligned");
    std::lock_guard<std::mutex> lock(g_pack_mu);
    PackCache* c = current_cache();
    SIV_CHECK_ARG(c != nullptr, "pack_cache_configure: no current HIP device");
    *c = PackCache();
    c->base = (char*)mem;
    c->size = mem ? bytes : 0;
    return SIV_OK;
}
extern "C" int siv_pack_cache_add_weights(const void* lo, size_t bytes) {
    SIV_CHECK_ARG(lo && bytes > 0, "pack_cache_add_weights: bad range");
    std::lock_guard<std::mutex> lock(g_pack_mu);
    PackCache* c = current_cache();
    SIV_CHECK_ARG(c != nullptr, "pack_cache_add_weights: no current HIP device");
    c->stable.emplace_back((const char*)lo, (const char*)lo + bytes);
    c->slots.clear();            // the range may reuse addresses an older range covered
    c->used = 0;
    return SIV_OK;
}
extern "C" int siv_pack_cache_remove_weights(const void* lo) {
    std::lock_guard<std::mutex> lock(g_pack_mu);
    PackCache* c = current_cache();
    if (!c) return SIV_OK;
    for (size_t i = 0; i < c->stable.size();)
        if (c->stable[i].first == (const char*)lo) c->stable.erase(c->stable.begin() + i);
        else ++i;
    c->slots.clear();
    c->used = 0;
    return SIV_OK;
}
extern "C" int siv_pack_cache_invalidate(void) {
    std::lock_guard<std::mutex> lock(g_pack_mu);
    PackCache* c = current_cache();
    if (c) c->slots.clear(), c->used = 0;
    return SIV_OK;
}
extern "C" int siv_pack_cache_invalidate_range(const void* lo, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_pack_mu);
    PackCache* c = current_cache();
    if (!c) return SIV_OK;
    for (auto& kv : c->slots) {
        const char* w = (const char*)std::get<0>(kv.first);
        if (w >= (const char*)lo && w < (const char*)lo + bytes) kv.second.valid = false;
    }
    return SIV_OK;
}
extern "C" long siv_pack_cache_hits(void) {
    std::lock_guard<std::mutex> lock(g_pack_mu);
    PackCache* c = current_cache();
    return c ? c->hits : 0;
}
extern "C" int siv_set_conv_path(int path) {
    SIV_CHECK_ARG(path >= 0 && path <= 6, "set_conv_path: path must be 0..6");
    g_conv_path.store(path, std::memory_order_relaxed);
    return SIV_OK;
}

extern "C" int siv_conv3d_k3_fwd(const float* x, const float* w, const float* bias, float* y, int N,
                                 int D, int H, int W, int Cin, int Cout, int ups, void* ws,
                                 size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(x && w && y, "conv3d_k3_fwd: null pointer");
    int rc = check_conv_shape("conv3d_k3_fwd", N, D, H, W, Cin, Cout, ups);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int path = conv_path();
    const bool al = aligned16(x) && aligned16(w) && aligned16(y) && aligned16(ws);
    if (path != 1 && al && ups == 1 && k3_thin_supported(Cin, Cout))
        return SIV_PICK("thin"), k3_thin_fwd_launch(x, w, bias, y, N, D, H, W, Cin, Cout, 0, s, nullptr);
    if (use_up(path, al, ups, Cin, Cout))
        return SIV_PICK("up"), k2_up_fwd_launch(x, w, bias, y, N, D, H, W, Cin, Cout, ws, ws_bytes, s, nullptr, path == 5 && Cout % 64 == 0, path == 6);
    if (use_b3(path, al, ups, Cin, Cout))
        return SIV_PICK("wino_b3"), k3_fwd_wino_b3_launch(x, w, bias, y, N, D, H, W, Cin, Cout, 0, ws, ws_bytes, s, nullptr);
    if (use_w2d(path, al, ups, N, D, H, W, Cin, Cout))
        return SIV_PICK("wino2d"), k3_fwd_wino2d_launch(x, w, bias, y, N, D, H, W, Cin, Cout, 0, ws, ws_bytes, s, nullptr,
                                                        ws_tail(ws, ws_bytes, k3_fwd_wino2d_ws_bytes(Cin, Cout)),
                                                        ws_tail_bytes(ws_bytes, k3_fwd_wino2d_ws_bytes(Cin, Cout)));
    if (use_wino(path, al, N, D, H, W, Cin, Cout))
        return SIV_PICK("wino"), k3_fwd_wino_launch(x, w, bias, y, N, D, H, W, Cin, Cout, ups, 0, ws, ws_bytes, s, nullptr);
    if (path != 1 && al && k3_fwd_mfma_supported(Cin, Cout))
        return SIV_PICK("mfma"), k3_fwd_mfma_launch(x, w, bias, y, N, D, H, W, Cin, Cout, ups, 0, ws, ws_bytes, s);
    SIV_CHECK_ARG(path != 2, "conv3d_k3_fwd: no specialised kernel for Cin=%d Cout=%d ups=%d", Cin,
                  Cout, ups);
    return SIV_PICK("generic"), k3_fwd_generic_launch(x, w, bias, y, N, D, H, W, Cin, Cout, ups, s);
}

extern "C" int siv_conv3d_k3_fwd_bnstats(const float* x, const float* w, const float* bias, float* y,
                                         int N, int D, int H, int W, int Cin, int Cout, int ups,
                                         float* mean, float* var, float* running_mean,
                                         float* running_var, int64_t* num_batches_tracked,
                                         float momentum, void* ws, size_t ws_bytes,
                                         siv_stream_t stream) {
    SIV_CHECK_ARG(x && w && y && mean && var, "conv3d_k3_fwd_bnstats: null pointer");
    SIV_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr),
                  "conv3d_k3_fwd_bnstats: running_mean and running_var must both be given or both be NULL");
    int rc = check_conv_shape("conv3d_k3_fwd_bnstats", N, D, H, W, Cin, Cout, ups);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)N * D * H * W;
    const bool al = aligned16(x) && aligned16(w) && aligned16(y) && aligned16(ws);
    // Kernels whose epilogue emits per-block sums of their bias-free output: the statistics then cost
    // one small finalize launch instead of a pass over y.  The partials sit behind the packed weights.
    const int path = conv_path();
    enum { NONE, UP, B3, W2D, WINO, THIN } fused = NONE;
    size_t wbytes = 0;
    long nblk = 0;
    if (path != 1 && al && ups == 1 && k3_thin_supported(Cin, Cout)) {
        // the single-channel stem: its matrix-pipe kernel sums what it stores; the other thin kernels do not
        nblk = k3_thin_fwd_stat_blocks(N, D, H, W, Cin, Cout);
        if (nblk > 0) fused = THIN;
    } else if (use_up(path, al, ups, Cin, Cout)) {
        fused = UP, wbytes = k2_up_fwd_ws_bytes(Cin, Cout), nblk = k2_up_fwd_blocks(N, D, H, W, path == 5 && Cout % 64 == 0, path == 6);
    } else if (use_b3(path, al, ups, Cin, Cout)) {
        fused = B3, wbytes = k3_fwd_wino_b3_ws_bytes(Cin, Cout), nblk = k3_fwd_wino_bricks(N, D, H, W);
    } else if (use_w2d(path, al, ups, N, D, H, W, Cin, Cout)) {
        fused = W2D, wbytes = k3_fwd_wino2d_ws_bytes(Cin, Cout), nblk = k3_fwd_wino2d_stat_rows(N, D, H, W, Cin, Cout);
    } else if (use_wino(path, al, N, D, H, W, Cin, Cout)) {
        fused = WINO, wbytes = k3_fwd_wino_ws_bytes(Cin, Cout), nblk = k3_fwd_wino_bricks(N, D, H, W);
    }
    if (fused != NONE) {
        wbytes = align256(wbytes);
        SIV_CHECK_WS(wbytes + (size_t)nblk * 2 * Cout * sizeof(float), ws_bytes, "conv3d_k3_fwd_bnstats");
        float* part = (float*)((char*)ws + wbytes);
        SIV_PICK(fused == UP ? "up" : fused == B3 ? "wino_b3" : fused == W2D ? "wino2d" : fused == THIN ? "thin" : "wino");
        switch (fused) {
        case THIN: rc = k3_thin_fwd_launch(x, w, bias, y, N, D, H, W, Cin, Cout, 0, s, part); break;
        case UP: rc = k2_up_fwd_launch(x, w, bias, y, N, D, H, W, Cin, Cout, ws, ws_bytes, s, part, path == 5 && Cout % 64 == 0, path == 6); break;
        case B3: rc = k3_fwd_wino_b3_launch(x, w, bias, y, N, D, H, W, Cin, Cout, 0, ws, ws_bytes, s, part); break;
        case W2D: {
            // the partial outputs of a split channel contraction go behind the statistics rows; a workspace
            // without room for them runs unsplit, and the unsplit launch writes one statistics row per brick
            // (which can be more rows than the split launch's: checked again)
            const size_t used = wbytes + (size_t)nblk * 2 * Cout * sizeof(float);
            void* kws = ws_tail(ws, ws_bytes, used);
            size_t kbytes = ws_tail_bytes(ws_bytes, used);
            if (kbytes < k3_fwd_wino2d_ksplit_bytes(N, D, H, W, Cin, Cout)) {
                kws = nullptr, kbytes = 0;
                nblk = k3_fwd_wino2d_bricks(N, D, H, W);
                SIV_CHECK_WS(wbytes + (size_t)nblk * 2 * Cout * sizeof(float), ws_bytes, "conv3d_k3_fwd_bnstats");
            }
            rc = k3_fwd_wino2d_launch(x, w, bias, y, N, D, H, W, Cin, Cout, 0, ws, ws_bytes, s, part, kws, kbytes);
            break;
        }
        default: rc = k3_fwd_wino_launch(x, w, bias, y, N, D, H, W, Cin, Cout, ups, 0, ws, ws_bytes, s, part); break;
        }
        if (rc) return rc;
        return bn_finalize_launch(bias, part, (int)nblk, M, Cout, mean, var, running_mean, running_var,
                                  num_batches_tracked, momentum, s);
    }
    rc = siv_conv3d_k3_fwd(x, w, bias, y, N, D, H, W, Cin, Cout, ups, ws, ws_bytes, stream);
    if (rc) return rc;
    return siv_bn_stats(y, M, Cout, mean, var, running_mean, running_var, num_batches_tracked, momentum, ws,
                        ws_bytes, stream);
}

// Conv3d(1, Cout, 3) -> BatchNorm -> LeakyReLU -> Dropout of the encoder stem with the convolution output
// recomputed by every pass that needs it instead of stored.
extern "C" int siv_conv1_bnact_supported(int Cout) {
    return conv_path() != 1 && Cout % 4 == 0 && Cout >= 4 && Cout <= 64 ? 1 : 0;
}
extern "C" int siv_conv1_bnact_fwd_bits(const float* x, const float* w, const float* bias, const float* gamma,
                                        const float* beta, const uint8_t* keep, uint64_t mask_seed,
                                        uint64_t mask_stream_id, const uint64_t* mask_epoch, float p, float slope,
                                        float eps, int train, float* mean, float* var, float* running_mean,
                                        float* running_var, int64_t* num_batches_tracked, float momentum,
                                        float* out, void* out_bf16, void* bits, int N, int D, int H, int W, int Cout,
                                        void* ws, size_t ws_bytes, siv_stream_t stream);
extern "C" int siv_conv1_bnact_fwd_bf16(const float* x, const float* w, const float* bias, const float* gamma,
                                        const float* beta, const uint8_t* keep, uint64_t mask_seed,
                                        uint64_t mask_stream_id, const uint64_t* mask_epoch, float p, float slope,
                                        float eps, int train, float* mean, float* var, float* running_mean,
                                        float* running_var, int64_t* num_batches_tracked, float momentum,
                                        float* out, void* out_bf16, int N, int D, int H, int W, int Cout, void* ws,
                                        size_t ws_bytes, siv_stream_t stream);
extern "C" int siv_conv1_bnact_fwd(const float* x, const float* w, const float* bias, const float* gamma,
                                   const float* beta, const uint8_t* keep, uint64_t mask_seed,
                                   uint64_t mask_stream_id, const uint64_t* mask_epoch, float p, float slope,
                                   float eps, int train, float* mean, float* var, float* running_mean,
                                   float* running_var, int64_t* num_batches_tracked, float momentum,
                                   float* out, int N, int D, int H, int W, int Cout, void* ws, size_t ws_bytes,
                                   siv_stream_t stream) {
    return siv_conv1_bnact_fwd_bf16(x, w, bias, gamma, beta, keep, mask_seed, mask_stream_id, mask_epoch, p, slope, eps,
                                    train, mean, var, running_mean, running_var, num_batches_tracked, momentum, out,
                                    nullptr, N, D, H, W, Cout, ws, ws_bytes, stream);
}
extern "C" int siv_conv1_bnact_fwd_bf16(const float* x, const float* w, const float* bias, const float* gamma,
                                        const float* beta, const uint8_t* keep, uint64_t mask_seed,
                                        uint64_t mask_stream_id, const uint64_t* mask_epoch, float p, float slope,
                                        float eps, int train, float* mean, float* var, float* running_mean,
                                        float* running_var, int64_t* num_batches_tracked, float momentum,
                                        float* out, void* out_bf16, int N, int D, int H, int W, int Cout, void* ws,
                                        size_t ws_bytes, siv_stream_t stream) {
    return siv_conv1_bnact_fwd_bits(x, w, bias, gamma, beta, keep, mask_seed, mask_stream_id, mask_epoch, p, slope, eps,
                                    train, mean, var, running_mean, running_var, num_batches_tracked, momentum, out,
                                    out_bf16, nullptr, N, D, H, W, Cout, ws, ws_bytes, stream);
}
extern "C" size_t siv_conv1_bnact_bits_bytes(int N, int D, int H, int W, int Cout) {
    return (size_t)N * D * H * W * (size_t)(Cout / 4);
}
extern "C" int siv_conv1_bnact_fwd_bits(const float* x, const float* w, const float* bias, const float* gamma,
                                        const float* beta, const uint8_t* keep, uint64_t mask_seed,
                                        uint64_t mask_stream_id, const uint64_t* mask_epoch, float p, float slope,
                                        float eps, int train, float* mean, float* var, float* running_mean,
                                        float* running_var, int64_t* num_batches_tracked, float momentum,
                                        float* out, void* out_bf16, void* bits, int N, int D, int H, int W, int Cout,
                                        void* ws, size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(((uintptr_t)out_bf16 & 7) == 0, "conv1_bnact_fwd: the bf16 copy must be 8-byte aligned");
    SIV_CHECK_ARG(x && w && gamma && beta && mean && var && out, "conv1_bnact_fwd: null pointer");
    int rc = check_conv_shape("conv1_bnact_fwd", N, D, H, W, 1, Cout, 1);
    if (rc) return rc;
    SIV_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr),
                  "conv1_bnact_fwd: running_mean and running_var must both be given or both be NULL");
    SIV_CHECK_ARG(p >= 0.f && p < 1.f, "conv1_bnact_fwd: dropout probability %g outside [0, 1)", (double)p);
    SIV_CHECK_ARG(aligned16(out) && aligned16(ws) && ((uintptr_t)keep & 3) == 0,
                  "conv1_bnact_fwd: out and the workspace must be 16-byte aligned, keep 4-byte aligned");
    SIV_PICK("thin");
    // no mask tensor and p > 0: the mask is drawn inside the kernels from (seed, stream id [, epoch])
    const PhiloxMask rk{mask_seed, mask_stream_id, (const unsigned long long*)mask_epoch, keep == nullptr && p > 0.f};
    return stem_bnact_fwd_launch(x, w, bias, gamma, beta, keep, rk, p, slope, eps, train, mean, var, running_mean,
                                 running_var, num_batches_tracked, momentum, out, N, D, H, W, Cout, ws, ws_bytes,
                                 (hipStream_t)stream, out_bf16, bits);
}
extern "C" int siv_conv1_bnact_bwd(const float* dout, const float* x, const float* w, const float* bias,
                                   const float* mean, const float* var, const float* gamma, const float* beta,
                                   const uint8_t* keep, uint64_t mask_seed, uint64_t mask_stream_id,
                                   const uint64_t* mask_epoch, float p, float slope, float eps, float* dy,
                                   float* dgamma, float* dbeta, int accumulate, int N, int D, int H, int W,
                                   int Cout, void* ws, size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(dout && x && w && mean && var && gamma && beta && dy, "conv1_bnact_bwd: null pointer");
    int rc = check_conv_shape("conv1_bnact_bwd", N, D, H, W, 1, Cout, 1);
    if (rc) return rc;
    SIV_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr),
                  "conv1_bnact_bwd: dgamma and dbeta must both be given or both be NULL");
    SIV_CHECK_ARG(aligned16(dout) && aligned16(dy) && aligned16(ws) && ((uintptr_t)keep & 3) == 0,
                  "conv1_bnact_bwd: dout, dy and the workspace must be 16-byte aligned, keep 4-byte aligned");
    SIV_PICK("thin");
    const PhiloxMask rk{mask_seed, mask_stream_id, (const unsigned long long*)mask_epoch, keep == nullptr && p > 0.f};
    return stem_bnact_bwd_launch(dout, x, w, bias, mean, var, gamma, beta, keep, rk, p, slope, eps, dy, dgamma, dbeta,
                                 accumulate, N, D, H, W, Cout, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int siv_conv1_bnact_bwd_bits(const float* dout, const float* x, const float* w, const float* bias,
                                        const float* mean, const float* var, const float* gamma, const float* beta,
                                        const void* bits, float p, float slope, float eps, float* dy, float* dgamma,
                                        float* dbeta, int accumulate, int N, int D, int H, int W, int Cout, void* ws,
                                        size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(dout && x && w && mean && var && gamma && beta && dy && bits, "conv1_bnact_bwd_bits: null pointer");
    int rc = check_conv_shape("conv1_bnact_bwd_bits", N, D, H, W, 1, Cout, 1);
    if (rc) return rc;
    SIV_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr),
                  "conv1_bnact_bwd_bits: dgamma and dbeta must both be given or both be NULL");
    SIV_CHECK_ARG(p >= 0.f && p < 1.f, "conv1_bnact_bwd_bits: dropout probability %g outside [0, 1)", (double)p);
    SIV_CHECK_ARG(aligned16(dout) && aligned16(dy) && aligned16(ws),
                  "conv1_bnact_bwd_bits: dout, dy and the workspace must be 16-byte aligned");
    SIV_PICK("thin");
    return stem_bnact_bwd_launch(dout, x, w, bias, mean, var, gamma, beta, nullptr, PhiloxMask{}, p, slope, eps, dy,
                                 dgamma, dbeta, accumulate, N, D, H, W, Cout, ws, ws_bytes, (hipStream_t)stream, bits);
}
extern "C" int siv_conv1_bnact_wgrad_bits(const float* dout, const float* x, const float* w, const float* bias,
                                          const float* mean, const float* var, const float* gamma, const void* bits,
                                          float p, float slope, float eps, float* dw, float* dbias, float* dgamma,
                                          float* dbeta, int accumulate, int N, int D, int H, int W, int Cout,
                                          void* ws, size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(dout && x && w && mean && var && gamma && bits, "conv1_bnact_wgrad_bits: null pointer");
    SIV_CHECK_ARG(dw || dbias || dgamma || dbeta, "conv1_bnact_wgrad_bits: no output requested");
    int rc = check_conv_shape("conv1_bnact_wgrad_bits", N, D, H, W, 1, Cout, 1);
    if (rc) return rc;
    SIV_CHECK_ARG(p >= 0.f && p < 1.f, "conv1_bnact_wgrad_bits: dropout probability %g outside [0, 1)", (double)p);
    SIV_CHECK_ARG(aligned16(dout) && aligned16(ws), "conv1_bnact_wgrad_bits: dout and the workspace must be 16-byte aligned");
    SIV_PICK("thin");
    return stem_bnact_wgrad_launch(dout, x, w, bias, mean, var, gamma, bits, p, slope, eps, dw, dbias, dgamma, dbeta,
                                   accumulate, N, D, H, W, Cout, ws, ws_bytes, (hipStream_t)stream);
}

// BatchNorm + LeakyReLU of the layer in front folded into a Cin -> 1 convolution (the decoder's output layer
// behind its last block): x is that block's raw convolution output, the activated tensor is never formed.
extern "C" int siv_bnact_conv3d_k3_to1_supported(int N, int D, int H, int W, int Cin) {
    return conv_path() != 1 && k3_thin_bn_fwd_supported(N, D, H, W, Cin) ? 1 : 0;
}
extern "C" int siv_bnact_conv3d_k3_to1_fwd_bf16(const float* x, const float* mean, const float* var,
                                                const float* gamma, const float* beta, float eps, float slope,
                                                const float* w, const float* bias, float* y, void* x_bf16, int N,
                                                int D, int H, int W, int Cin, siv_stream_t stream) {
    SIV_CHECK_ARG(x && mean && var && gamma && beta && w && y, "bnact_conv3d_k3_to1_fwd: null pointer");
    int rc = check_conv_shape("bnact_conv3d_k3_to1_fwd", N, D, H, W, Cin, 1, 1);
    if (rc) return rc;
    SIV_CHECK_ARG(aligned16(x) && aligned16(w) && ((uintptr_t)x_bf16 & 7) == 0,
                  "bnact_conv3d_k3_to1_fwd: x and w must be 16-byte aligned, the bf16 copy 8-byte aligned");
    SIV_PICK("thin");
    return k3_thin_bn_fwd_launch(x, BnPrologue{mean, var, gamma, beta, eps, slope}, w, bias, y, N, D, H, W, Cin,
                                 (hipStream_t)stream, x_bf16);
}
extern "C" int siv_bnact_conv3d_k3_to1_fwd(const float* x, const float* mean, const float* var,
                                           const float* gamma, const float* beta, float eps, float slope,
                                           const float* w, const float* bias, float* y, int N, int D, int H,
                                           int W, int Cin, siv_stream_t stream) {
    return siv_bnact_conv3d_k3_to1_fwd_bf16(x, mean, var, gamma, beta, eps, slope, w, bias, y, nullptr, N, D, H, W, Cin,
                                            stream);
}
static int to1_wgrad_impl(const float* x, bool x_bf16, const float* mean, const float* var,
                          const float* gamma, const float* beta, float eps, float slope,
                          const float* dy, float* dw, float* dbias, int N, int D, int H,
                          int W, int Cin, int accumulate, void* ws, size_t ws_bytes,
                          siv_stream_t stream) {
    SIV_CHECK_ARG(x && mean && var && gamma && beta && dy && dw, "bnact_conv3d_k3_to1_wgrad: null pointer");
    int rc = check_conv_shape("bnact_conv3d_k3_to1_wgrad", N, D, H, W, Cin, 1, 1);
    if (rc) return rc;
    SIV_CHECK_ARG(Cin % 4 == 0 && Cin <= 64 && aligned16(x),
                  "bnact_conv3d_k3_to1_wgrad: needs Cin %% 4 == 0, Cin <= 64 and a 16-byte aligned x");
    const BnPrologue bn{mean, var, gamma, beta, eps, slope};
    SIV_PICK("thin");
    return k3_thin_wgrad_launch(x, dy, dw, dbias, N, D, H, W, Cin, 1, accumulate, ws, ws_bytes,
                                (hipStream_t)stream, &bn, x_bf16);
}
extern "C" int siv_bnact_conv3d_k3_to1_wgrad(const float* x, const float* mean, const float* var,
                                             const float* gamma, const float* beta, float eps, float slope,
                                             const float* dy, float* dw, float* dbias, int N, int D, int H,
                                             int W, int Cin, int accumulate, void* ws, size_t ws_bytes,
                                             siv_stream_t stream) {
    return to1_wgrad_impl(x, false, mean, var, gamma, beta, eps, slope, dy, dw, dbias, N, D, H, W, Cin, accumulate, ws,
                          ws_bytes, stream);
}
extern "C" int siv_bnact_conv3d_k3_to1_wgrad_bf16(const void* x_bf16, const float* mean, const float* var,
                                                  const float* gamma, const float* beta, float eps, float slope,
                                                  const float* dy, float* dw, float* dbias, int N, int D, int H,
                                                  int W, int Cin, int accumulate, void* ws, size_t ws_bytes,
                                                  siv_stream_t stream) {
    return to1_wgrad_impl((const float*)x_bf16, true, mean, var, gamma, beta, eps, slope, dy, dw, dbias, N, D, H, W, Cin,
                          accumulate, ws, ws_bytes, stream);
}

static int to1_dgrad_impl(const float* dy, const float* w, const float* x, bool x_bf16,
                          const float* mean, const float* var, const float* gamma,
                          const float* beta, float eps, float slope, float* dx,
                          float* dgamma, float* dbeta, int accumulate, int N, int D, int H,
                          int W, int Cin, void* ws, size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(dy && w && x && mean && var && gamma && beta && dx, "bnact_conv3d_k3_to1_dgrad: null pointer");
    int rc = check_conv_shape("bnact_conv3d_k3_to1_dgrad", N, D, H, W, Cin, 1, 1);
    if (rc) return rc;
    SIV_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr),
                  "bnact_conv3d_k3_to1_dgrad: dgamma and dbeta must both be given or both be NULL");
    SIV_CHECK_ARG(aligned16(x) && aligned16(dx) && aligned16(ws),
                  "bnact_conv3d_k3_to1_dgrad: x, dx and the workspace must be 16-byte aligned");
    SIV_PICK("thin");
    return k3_thin_bn_dgrad_launch(dy, w, x, BnPrologue{mean, var, gamma, beta, eps, slope}, dx, dgamma, dbeta,
                                   accumulate, N, D, H, W, Cin, ws, ws_bytes, (hipStream_t)stream, x_bf16);
}
extern "C" int siv_bnact_conv3d_k3_to1_dgrad(const float* dy, const float* w, const float* x,
                                             const float* mean, const float* var, const float* gamma,
                                             const float* beta, float eps, float slope, float* dx,
                                             float* dgamma, float* dbeta, int accumulate, int N, int D, int H,
                                             int W, int Cin, void* ws, size_t ws_bytes, siv_stream_t stream) {
    return to1_dgrad_impl(dy, w, x, false, mean, var, gamma, beta, eps, slope, dx, dgamma, dbeta, accumulate, N, D, H, W,
                          Cin, ws, ws_bytes, stream);
}
extern "C" int siv_bnact_conv3d_k3_to1_dgrad_bf16(const float* dy, const float* w, const void* x_bf16,
                                                  const float* mean, const float* var, const float* gamma,
                                                  const float* beta, float eps, float slope, float* dx,
                                                  float* dgamma, float* dbeta, int accumulate, int N, int D, int H,
                                                  int W, int Cin, void* ws, size_t ws_bytes, siv_stream_t stream) {
    return to1_dgrad_impl(dy, w, (const float*)x_bf16, true, mean, var, gamma, beta, eps, slope, dx, dgamma, dbeta,
                          accumulate, N, D, H, W, Cin, ws, ws_bytes, stream);
}

// Input gradient of the 3x3x3 convolution = forward convolution of dy with the flipped, transposed
// weights; `bias` (per dx channel, may be NULL) makes the same call a transposed convolution.
static int dgrad_dispatch(const char* what, const float* dy, const float* w, const float* bias, float* dx,
                          int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes,
                          hipStream_t s) {
    const int path = conv_path();
    const bool al = aligned16(dy) && aligned16(w) && aligned16(dx) && aligned16(ws);
    if (path != 1 && al && k3_thin_supported(Cout, Cin))
        return SIV_PICK("thin"), k3_thin_fwd_launch(dy, w, bias, dx, N, D, H, W, Cout, Cin, 1, s, nullptr);
    if (use_b3(path, al, 1, Cout, Cin))
        return SIV_PICK("wino_b3"), k3_fwd_wino_b3_launch(dy, w, bias, dx, N, D, H, W, Cout, Cin, 1, ws, ws_bytes, s, nullptr);
    if (use_w2d(path, al, 1, N, D, H, W, Cout, Cin))
        return SIV_PICK("wino2d"), k3_fwd_wino2d_launch(dy, w, bias, dx, N, D, H, W, Cout, Cin, 1, ws, ws_bytes, s, nullptr,
                                                        ws_tail(ws, ws_bytes, k3_fwd_wino2d_ws_bytes(Cout, Cin)),
                                                        ws_tail_bytes(ws_bytes, k3_fwd_wino2d_ws_bytes(Cout, Cin)));
    if (use_wino(path, al, N, D, H, W, Cout, Cin))
        return SIV_PICK("wino"), k3_fwd_wino_launch(dy, w, bias, dx, N, D, H, W, Cout, Cin, 1, 1, ws, ws_bytes, s, nullptr);
    if (path != 1 && al && k3_fwd_mfma_supported(Cout, Cin))
        return SIV_PICK("mfma"), k3_fwd_mfma_launch(dy, w, bias, dx, N, D, H, W, Cout, Cin, 1, 1, ws, ws_bytes, s);
    SIV_CHECK_ARG(path != 2, "%s: no specialised kernel for Cin=%d Cout=%d", what, Cin, Cout);
    return SIV_PICK("generic"), k3_dgrad_generic_launch(dy, w, bias, dx, N, D, H, W, Cin, Cout, s);
}

extern "C" int siv_conv3d_k3_dgrad(const float* dy, const float* w, float* dx, int N, int D, int H,
                                   int W, int Cin, int Cout, void* ws, size_t ws_bytes,
                                   siv_stream_t stream) {
    SIV_CHECK_ARG(dy && w && dx, "conv3d_k3_dgrad: null pointer");
    int rc = check_conv_shape("conv3d_k3_dgrad", N, D, H, W, Cin, Cout, 1);
    if (rc) return rc;
    return dgrad_dispatch("conv3d_k3_dgrad", dy, w, nullptr, dx, N, D, H, W, Cin, Cout, ws, ws_bytes,
                          (hipStream_t)stream);
}

// ---- nn.ConvTranspose3d(k=3, s=1, p=1): the adjoint of the convolution above ----------------------
// A transposed convolution Cin -> Cout with weight [Cin][kd][kh][kw][Cout] is the input gradient of the
// convolution Cout -> Cin that has the same weight array; its own input gradient is that convolution's
// forward and its weight gradient that convolution's weight gradient with the roles of x and dy swapped.
extern "C" int siv_conv_transpose3d_k3_fwd(const float* x, const float* w, const float* bias, float* y,
                                           int N, int D, int H, int W, int Cin, int Cout, void* ws,
                                           size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(x && w && y, "conv_transpose3d_k3_fwd: null pointer");
    int rc = check_conv_shape("conv_transpose3d_k3_fwd", N, D, H, W, Cin, Cout, 1);
    if (rc) return rc;
    return dgrad_dispatch("conv_transpose3d_k3_fwd", x, w, bias, y, N, D, H, W, Cout, Cin, ws, ws_bytes,
                          (hipStream_t)stream);
}

extern "C" int siv_conv_transpose3d_k3_dgrad(const float* dy, const float* w, float* dx, int N, int D,
                                             int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes,
                                             siv_stream_t stream) {
    SIV_CHECK_ARG(dy && w && dx, "conv_transpose3d_k3_dgrad: null pointer");
    return siv_conv3d_k3_fwd(dy, w, nullptr, dx, N, D, H, W, Cout, Cin, 1, ws, ws_bytes, stream);
}

extern "C" int siv_conv_transpose3d_k3_wgrad(const float* x, const float* dy, float* dw, float* dbias,
                                             int N, int D, int H, int W, int Cin, int Cout,
                                             int accumulate, void* ws, size_t ws_bytes,
                                             siv_stream_t stream) {
    SIV_CHECK_ARG(x && dy && dw, "conv_transpose3d_k3_wgrad: null pointer");
    int rc = siv_conv3d_k3_wgrad(dy, x, dw, nullptr, N, D, H, W, Cout, Cin, 1, accumulate, ws, ws_bytes, stream);
    if (rc) return rc;
    if (dbias)
        return colsum_launch(dy, dbias, (long)N * D * H * W, Cout, accumulate, ws, ws_bytes, (hipStream_t)stream);
    return SIV_OK;
}

extern "C" int siv_conv3d_k3_dgrad_up_supported(int Cin, int Cout) {
    const int path = conv_path();
    return (path == 0 || path == 2 || path >= 5) && k2_up_dgrad_supported(Cin, Cout) ? 1 : 0;
}

extern "C" int siv_conv3d_k3_dgrad_up(const float* dy, const float* w, float* dx, int N, int D, int H,
                                      int W, int Cin, int Cout, void* ws, size_t ws_bytes,
                                      siv_stream_t stream) {
    SIV_CHECK_ARG(dy && w && dx, "conv3d_k3_dgrad_up: null pointer");
    int rc = check_conv_shape("conv3d_k3_dgrad_up", N, D, H, W, Cin, Cout, 2);
    if (rc) return rc;
    SIV_CHECK_ARG(aligned16(dy) && aligned16(w) && aligned16(dx) && aligned16(ws),
                  "conv3d_k3_dgrad_up: pointers must be 16-byte aligned");
    SIV_CHECK_ARG(k2_up_dgrad_supported(Cin, Cout),
                  "conv3d_k3_dgrad_up: needs Cin %% 16 == 0 and Cout %% 8 == 0 (got Cin=%d Cout=%d); use "
                  "siv_conv3d_k3_dgrad followed by siv_upsample2_bwd", Cin, Cout);
    SIV_PICK("up");
    return k2_up_dgrad_launch(dy, w, dx, N, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream, conv_path() == 5,
                              conv_path() == 6);
}

extern "C" int siv_conv3d_k3_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N,
                                   int D, int H, int W, int Cin, int Cout, int ups, int accumulate,
                                   void* ws, size_t ws_bytes, siv_stream_t stream) {
    SIV_CHECK_ARG(x && dy && dw, "conv3d_k3_wgrad: null pointer");
    int rc = check_conv_shape("conv3d_k3_wgrad", N, D, H, W, Cin, Cout, ups);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int path = conv_path();
    const bool al = aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(ws);
    if (path != 1 && al && ups == 1 && k3_thin_supported(Cin, Cout))
        return SIV_PICK("thin"), k3_thin_wgrad_launch(x, dy, dw, dbias, N, D, H, W, Cin, Cout, accumulate, ws, ws_bytes, s, nullptr);
    if (use_gw(path, al, ups, N, D, H, W, Cin, Cout)) {
        SIV_PICK(ups == 2 ? "up_wino" : "wino2d");
        rc = ups == 2 ? k2_up_wgrad_wino_launch(x, dy, dw, N, D, H, W, Cin, Cout, accumulate, ws, ws_bytes, s)
                      : k3_wgrad_wino2d_launch(x, dy, dw, N, D, H, W, Cin, Cout, accumulate, ws, ws_bytes, s);
        if (rc) return rc;
        if (dbias) return colsum_launch(dy, dbias, (long)N * D * H * W, Cout, accumulate, ws, ws_bytes, s);
        return SIV_OK;
    }
    if (path != 1 && al && k3_wgrad_mfma_supported(Cin, Cout)) {
        SIV_PICK("mfma");
        if (ups == 2 && (path == 0 || path == 2 || path >= 5) && k2_up_wgrad_supported(Cin, Cout))
            SIV_PICK("up"), rc = k2_up_wgrad_launch(x, dy, dw, N, D, H, W, Cin, Cout, accumulate, ws, ws_bytes, s);
        else
            rc = k3_wgrad_mfma_launch(x, dy, dw, N, D, H, W, Cin, Cout, ups, accumulate, ws, ws_bytes, s);
        if (rc) return rc;
        if (dbias) return colsum_launch(dy, dbias, (long)N * D * H * W, Cout, accumulate, ws, ws_bytes, s);
        return SIV_OK;
    }
    SIV_CHECK_ARG(path != 2, "conv3d_k3_wgrad: no specialised kernel for Cin=%d Cout=%d ups=%d", Cin,
                  Cout, ups);
    return SIV_PICK("generic"), k3_wgrad_generic_launch(x, dy, dw, dbias, N, D, H, W, Cin, Cout, ups, accumulate, ws, ws_bytes, s);
}
