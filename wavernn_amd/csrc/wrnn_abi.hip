// wrnn_abi.hip -- host side of the C ABI declared in include/wavernn_amd.h.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>

#include "../../include/wavernn_amd.h"
#include "wrnn_device.h"

namespace wrnn {
hipError_t launch_cond(const CondArgs &a, int n_cus, hipStream_t stream);
hipError_t launch_noise_mol(const float *in, float *out, long n, int B, int n_cus, hipStream_t stream);
hipError_t launch_stream(const LoopArgs &args, int mode, hipStream_t stream);
hipError_t launch_persist(const LoopArgs &args, int U, int mode, hipStream_t stream);
hipError_t launch_cluster(const LoopArgs &args, int U, int ncl, int mode, int nl, hipStream_t stream);
int cluster_count(int U, int n_cus);
hipError_t launch_pipe(const LoopArgs &args, int G, int ncl, int nl, hipStream_t stream);
int pipe_rows(int G);
int pipe_clusters(int n_cus);
hipError_t launch_sparse(const LoopArgs &args, int G, int ncl, int nbp, hipStream_t stream);
int sparse_clusters(int n_cus);
size_t persist_lds_bytes();
int selftest_mfma(char *msg, size_t n);
int selftest_allgather(int n_cus, char *msg, size_t n, float *us_per_round);
}  // namespace wrnn

using namespace wrnn;

static thread_local char g_err[512] = "";
static void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return WRNN_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

struct wrnn_pack {
    int device, n_cus, C, mode;
    size_t weight_bytes;
    char *dev;          // one allocation
    size_t dev_bytes;
    // device pointers into `dev`
    const float *I_w0, *I_b, *I_cT;
    const float *w_ih1, *w_hh1, *b_ih1, *b_hh1, *w_ih2, *w_hh2, *b_ih2, *b_hh2;
    const float *fc1_w, *fc1_b, *fc2_w, *fc2_b, *fc3_w, *fc3_b;
    const float *w_ih1T, *w_hh1T, *w_ih2T, *w_hh2T, *fc1T, *fc2T, *fc3T, *c2_wT, *c3_wT, *c4_wT;
    hipEvent_t ev0, ev1;
    bool timed;
    const char *last_kernel;
    int last_U, last_ncl, last_G;
    int sp_nbp;                // 0 = the GRU matrices are not block-sparse enough for wrnn_sparse_kernel; else 48 / 64
    int sp_max_blocks;
    const float *sp_vals;
    const int *sp_cols;
};

extern "C" const char *wrnn_last_error(void) { return g_err; }
extern "C" int wrnn_abi_version(void) { return WRNN_ABI_VERSION; }

extern "C" int wrnn_device_cus(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device >= n) {
        set_err("no HIP device %d (count %d)", device, n);
        return WRNN_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return WRNN_ERR_NO_DEVICE;
    return prop.multiProcessorCount;
}

namespace {
struct Builder {
    std::vector<float> host;
    size_t add(const float *src, size_t n)
    {
        size_t off = (host.size() + 63) / 64 * 64;   // 256-byte alignment
        host.resize(off + n);
        if (src) memcpy(host.data() + off, src, n * sizeof(float));
        return off;
    }
    // dst[k][r] = src[r][col0 + k], src is [rows][ld]
    size_t add_T(const float *src, int rows, int ld, int col0, int ncols)
    {
        size_t off = add(nullptr, (size_t)ncols * rows);
        float *d = host.data() + off;
        for (int r = 0; r < rows; ++r)
            for (int k = 0; k < ncols; ++k) d[(size_t)k * rows + r] = src[(size_t)r * ld + col0 + k];
        return off;
    }
};
}  // namespace

extern "C" int wrnn_pack_create(const wrnn_weights *w, int device, wrnn_pack **out)
{
    if (!w || !out) { set_err("NULL argument"); return WRNN_ERR_ARG; }
    if (w->rnn_dims != H || w->fc_dims != H || w->feat_dims != MEL || w->aux_dims != AUX) {
        set_err("this build supports rnn_dims=fc_dims=512, feat_dims=80, aux_dims=32 (got %d,%d,%d,%d)",
                w->rnn_dims, w->fc_dims, w->feat_dims, w->aux_dims);
        return WRNN_ERR_ARG;
    }
    const int C = w->n_classes;
    if (w->mode == WRNN_MODE_MOL) {
        if (C != 30) { set_err("MOL needs n_classes == 30"); return WRNN_ERR_ARG; }
    } else if (w->mode == WRNN_MODE_RAW) {
        if (C < 2 || C > H) { set_err("RAW needs 2 <= n_classes <= 512"); return WRNN_ERR_ARG; }
    } else { set_err("unknown mode %d", w->mode); return WRNN_ERR_ARG; }
    const float *const *ptrs = &w->I_w;
    for (int i = 0; i < 16; ++i)
        if (!ptrs[i]) { set_err("NULL weight pointer #%d", i); return WRNN_ERR_ARG; }
    const int cus = wrnn_device_cus(device);
    if (cus < 0) return cus;
    HIPCHK(hipSetDevice(device));

    const int KI = 1 + MEL + AUX, K2 = H + AUX;
    Builder b;
    std::vector<float> col0(H);
    for (int r = 0; r < H; ++r) col0[r] = w->I_w[(size_t)r * KI];
    const size_t o_I_w0 = b.add(col0.data(), H), o_I_b = b.add(w->I_b, H);
    const size_t o_I_cT = b.add_T(w->I_w, H, KI, 1, KCOND);
    const size_t o_w_ih1 = b.add(w->w_ih1, (size_t)3 * H * H), o_w_hh1 = b.add(w->w_hh1, (size_t)3 * H * H);
    const size_t o_b_ih1 = b.add(w->b_ih1, 3 * H), o_b_hh1 = b.add(w->b_hh1, 3 * H);
    const size_t o_w_ih2 = b.add(w->w_ih2, (size_t)3 * H * K2), o_w_hh2 = b.add(w->w_hh2, (size_t)3 * H * H);
    const size_t o_b_ih2 = b.add(w->b_ih2, 3 * H), o_b_hh2 = b.add(w->b_hh2, 3 * H);
    const size_t o_fc1_w = b.add(w->fc1_w, (size_t)H * K2), o_fc1_b = b.add(w->fc1_b, H);
    const size_t o_fc2_w = b.add(w->fc2_w, (size_t)H * K2), o_fc2_b = b.add(w->fc2_b, H);
    const size_t o_fc3_w = b.add(w->fc3_w, (size_t)C * H), o_fc3_b = b.add(w->fc3_b, C);
    const size_t o_w_ih1T = b.add_T(w->w_ih1, 3 * H, H, 0, H), o_w_hh1T = b.add_T(w->w_hh1, 3 * H, H, 0, H);
    const size_t o_w_ih2T = b.add_T(w->w_ih2, 3 * H, K2, 0, H), o_w_hh2T = b.add_T(w->w_hh2, 3 * H, H, 0, H);
    const size_t o_fc1T = b.add_T(w->fc1_w, H, K2, 0, H), o_fc2T = b.add_T(w->fc2_w, H, K2, 0, H);
    const size_t o_fc3T = b.add_T(w->fc3_w, C, H, 0, H);
    const size_t o_c2_wT = b.add_T(w->w_ih2, 3 * H, K2, H, AUX);
    const size_t o_c3_wT = b.add_T(w->fc1_w, H, K2, H, AUX), o_c4_wT = b.add_T(w->fc2_w, H, K2, H, AUX);
    // ---- block-sparse view of the GRU matrices (16x1 blocks: 16 consecutive rows of one gate x 1 column) -----------
    // usable by wrnn_sparse_kernel when every block row keeps <= 64 columns (~5 % density keeps ~26 +- 5)
    int sp_nbp = 0, sp_max = 0;
    size_t o_spv = 0, o_spc = 0;
    {
        const float *mats[4] = {w->w_ih1, w->w_hh1, w->w_ih2, w->w_hh2};
        const int lds_[4] = {H, H, K2, H};
        std::vector<std::vector<int>> cols((size_t)4 * 32 * 3);
        for (int m = 0; m < 4; ++m)
            for (int wg = 0; wg < 32; ++wg)
                for (int g = 0; g < 3; ++g) {
                    std::vector<int> &cv = cols[((size_t)m * 32 + wg) * 3 + g];
                    const float *base = mats[m] + ((size_t)g * H + 16 * wg) * lds_[m];
                    for (int c = 0; c < H; ++c) {
                        bool nz = false;
                        for (int r = 0; r < 16 && !nz; ++r) nz = base[(size_t)r * lds_[m] + c] != 0.0f;
                        if (nz) cv.push_back(c);
                    }
                    if ((int)cv.size() > sp_max) sp_max = (int)cv.size();
                }
        if (sp_max <= 64 && w->mode == WRNN_MODE_MOL) {
            sp_nbp = sp_max <= 48 ? 48 : 64;
            o_spv = b.add(nullptr, (size_t)4 * 32 * 3 * sp_nbp * 16);
            o_spc = b.add(nullptr, (size_t)4 * 32 * 3 * sp_nbp);        // ints stored in the float builder (same width)
            for (size_t q = 0; q < (size_t)4 * 32 * 3 * sp_nbp * 16; ++q) b.host[o_spv + q] = 0.f;
            int *ci = reinterpret_cast<int *>(b.host.data() + o_spc);
            for (size_t q = 0; q < (size_t)4 * 32 * 3 * sp_nbp; ++q) ci[q] = 0;
            for (int m = 0; m < 4; ++m)
                for (int wg = 0; wg < 32; ++wg)
                    for (int g = 0; g < 3; ++g) {
                        const size_t br = ((size_t)m * 32 + wg) * 3 + g;
                        const std::vector<int> &cv = cols[br];
                        const float *base = mats[m] + ((size_t)g * H + 16 * wg) * lds_[m];
                        for (size_t k = 0; k < cv.size(); ++k) {
                            ci[br * sp_nbp + k] = cv[k];
                            for (int r = 0; r < 16; ++r) b.host[o_spv + (br * sp_nbp + k) * 16 + r] = base[(size_t)r * lds_[m] + cv[k]];
                        }
                    }
        }
    }
    b.add(nullptr, 64);   // tail padding so vector loads past the last array stay inside the allocation

    wrnn_pack *p = new wrnn_pack();
    p->device = device; p->n_cus = cus; p->C = C; p->mode = w->mode;
    p->dev_bytes = b.host.size() * sizeof(float);
    // W of SURVEY.md section 8(d): every loop parameter the reference touches per step
    p->weight_bytes = sizeof(float) * ((size_t)H * KI + H + 2 * ((size_t)3 * H * H) + (size_t)3 * H * K2 + (size_t)3 * H * H +
                                       4 * 3 * H + 2 * ((size_t)H * K2 + H) + (size_t)C * H + C);
    hipError_t e = hipMalloc((void **)&p->dev, p->dev_bytes);
    if (e != hipSuccess) { set_err("hipMalloc(%zu) failed: %s", p->dev_bytes, hipGetErrorString(e)); delete p; return WRNN_ERR_HIP; }
    e = hipMemcpy(p->dev, b.host.data(), p->dev_bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_err("hipMemcpy failed: %s", hipGetErrorString(e)); (void)hipFree(p->dev); delete p; return WRNN_ERR_HIP; }
    const float *base = (const float *)p->dev;
    p->I_w0 = base + o_I_w0; p->I_b = base + o_I_b; p->I_cT = base + o_I_cT;
    p->w_ih1 = base + o_w_ih1; p->w_hh1 = base + o_w_hh1; p->b_ih1 = base + o_b_ih1; p->b_hh1 = base + o_b_hh1;
    p->w_ih2 = base + o_w_ih2; p->w_hh2 = base + o_w_hh2; p->b_ih2 = base + o_b_ih2; p->b_hh2 = base + o_b_hh2;
    p->fc1_w = base + o_fc1_w; p->fc1_b = base + o_fc1_b; p->fc2_w = base + o_fc2_w; p->fc2_b = base + o_fc2_b;
    p->fc3_w = base + o_fc3_w; p->fc3_b = base + o_fc3_b;
    p->w_ih1T = base + o_w_ih1T; p->w_hh1T = base + o_w_hh1T; p->w_ih2T = base + o_w_ih2T; p->w_hh2T = base + o_w_hh2T;
    p->fc1T = base + o_fc1T; p->fc2T = base + o_fc2T; p->fc3T = base + o_fc3T;
    p->c2_wT = base + o_c2_wT; p->c3_wT = base + o_c3_wT; p->c4_wT = base + o_c4_wT;
    p->sp_nbp = sp_nbp; p->sp_max_blocks = sp_max;
    p->sp_vals = sp_nbp ? base + o_spv : nullptr;
    p->sp_cols = sp_nbp ? reinterpret_cast<const int *>(base + o_spc) : nullptr;
    HIPCHK(hipEventCreate(&p->ev0));
    HIPCHK(hipEventCreate(&p->ev1));
    p->timed = false;
    p->last_kernel = "";
    p->last_U = 0; p->last_ncl = 0; p->last_G = 0;
    *out = p;
    return WRNN_OK;
}

extern "C" void wrnn_pack_destroy(wrnn_pack *p)
{
    if (!p) return;
    (void)hipEventDestroy(p->ev0);
    (void)hipEventDestroy(p->ev1);
    (void)hipFree(p->dev);
    delete p;
}

extern "C" size_t wrnn_pack_weight_bytes(const wrnn_pack *p) { return p ? p->weight_bytes : 0; }

extern "C" int wrnn_pack_sparse_blocks(const wrnn_pack *p) { return p ? (p->sp_nbp ? p->sp_max_blocks : -p->sp_max_blocks) : 0; }

namespace {
struct WsLayout {
    size_t status, prof, gran, segs, c2f, c3f, c4f, cI, npre, total;
};
constexpr size_t GRAN_BYTES = (size_t)GRAN_WORDS * sizeof(u64);
size_t al(size_t x) { return (x + 255) / 256 * 256; }
WsLayout ws_layout(int B, int T, int n_frames, bool mol = true)
{
    WsLayout l;
    size_t o = 0;
    l.status = o; o = al(o + STATUS_WORDS * sizeof(unsigned));
    l.prof = o;   o = al(o + (size_t)MAXWG * NPROF * sizeof(u64));     // fixed offset: wrnn_profile_read finds it
    l.gran = o;   o = al(o + GRAN_BYTES);
    l.segs = o;   o = al(o + (size_t)2 * B * sizeof(int));
    l.c2f = o;    o = al(o + (size_t)(n_frames + 1) * 3 * H * sizeof(float));
    l.c3f = o;    o = al(o + (size_t)(n_frames + 1) * H * sizeof(float));
    l.c4f = o;    o = al(o + (size_t)(n_frames + 1) * H * sizeof(float));
    l.cI = o;     o = al(o + (size_t)T * B * H * sizeof(float));
    l.npre = o;   if (mol) o = al(o + (size_t)T * 11 * B * sizeof(float));   // derived MOL noise (pipelined kernel)
    l.total = o;
    return l;
}
int check_geometry(const wrnn_geometry *g)
{
    if (!g) { set_err("NULL geometry"); return WRNN_ERR_ARG; }
    if (g->B < 1 || g->T < 1 || g->L < 1 || g->hop < 1 || g->n_frames < 1 || g->stride < 0) {
        set_err("bad geometry B=%d T=%d stride=%d L=%d hop=%d n_frames=%d", g->B, g->T, g->stride, g->L, g->hop, g->n_frames);
        return WRNN_ERR_ARG;
    }
    if ((long)g->n_frames * g->hop < g->L) { set_err("n_frames*hop < L"); return WRNN_ERR_ARG; }
    if ((double)g->B * g->stride + g->T > 2.0e9) { set_err("geometry overflows int32"); return WRNN_ERR_ARG; }
    return WRNN_OK;
}
int check_segments(int B, int T, const int32_t *seg_pos, const int32_t *seg_lim, int L, int hop, int n_frames)
{
    if (B < 1 || T < 1 || L < 1 || hop < 1 || n_frames < 1 || !seg_pos || !seg_lim) {
        set_err("bad segment table B=%d T=%d L=%d hop=%d n_frames=%d", B, T, L, hop, n_frames);
        return WRNN_ERR_ARG;
    }
    if ((long)n_frames * hop < L) { set_err("n_frames*hop < L"); return WRNN_ERR_ARG; }
    for (int b = 0; b < B; ++b) {
        if (seg_pos[b] < 0 || seg_lim[b] < 0 || seg_lim[b] > L || (double)seg_pos[b] + T > 2.0e9) {
            set_err("segment %d: pos=%d lim=%d outside [0,%d]", b, seg_pos[b], seg_lim[b], L);
            return WRNN_ERR_ARG;
        }
    }
    return WRNN_OK;
}
}  // namespace

extern "C" size_t wrnn_workspace_bytes(const wrnn_pack *p, const wrnn_geometry *g)
{
    if (!p || check_geometry(g) != WRNN_OK) return 0;
    return ws_layout(g->B, g->T, g->n_frames).total;
}

extern "C" size_t wrnn_workspace_bytes_segments(const wrnn_pack *p, int32_t n_segments, int32_t T, int32_t n_frames)
{
    if (!p || n_segments < 1 || T < 1 || n_frames < 1) return 0;
    return ws_layout(n_segments, T, n_frames).total;
}

extern "C" int wrnn_generate_segments(const wrnn_pack *pc, int32_t B, int32_t T, const int32_t *seg_pos,
                                      const int32_t *seg_lim, int32_t L, int32_t hop, int32_t n_frames,
                                      const float *mels_up, const float *aux, const float *noise, float *out,
                                      void *workspace, size_t workspace_bytes, int algo, const wrnn_debug *dbg,
                                      void *stream_)
{
    wrnn_pack *p = const_cast<wrnn_pack *>(pc);
    if (!p || !mels_up || !aux || !noise || !out || !workspace) { set_err("NULL argument"); return WRNN_ERR_ARG; }
    int rc = check_segments(B, T, seg_pos, seg_lim, L, hop, n_frames);
    if (rc != WRNN_OK) return rc;
    const WsLayout l = ws_layout(B, T, n_frames);
    if (workspace_bytes < l.total) { set_err("workspace %zu < required %zu", workspace_bytes, l.total); return WRNN_ERR_WORKSPACE; }
    if (((uintptr_t)workspace & 255) != 0) { set_err("workspace must be 256-byte aligned"); return WRNN_ERR_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipSetDevice(p->device));
    char *ws = (char *)workspace;

    HIPCHK(hipMemsetAsync(ws + l.status, 0, STATUS_WORDS * sizeof(unsigned), stream));
    // segment table -> device (pageable source: the runtime stages it before returning)
    HIPCHK(hipMemcpyAsync(ws + l.segs, seg_pos, (size_t)B * sizeof(int), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(ws + l.segs + (size_t)B * sizeof(int), seg_lim, (size_t)B * sizeof(int), hipMemcpyHostToDevice, stream));
    const int *d_pos = (const int *)(ws + l.segs), *d_lim = d_pos + B;

    CondArgs c;
    c.mels_up = mels_up; c.aux = aux; c.I_cT = p->I_cT; c.I_b = p->I_b; c.c2_wT = p->c2_wT; c.b_ih2 = p->b_ih2;
    c.c3_wT = p->c3_wT; c.fc1_b = p->fc1_b; c.c4_wT = p->c4_wT; c.fc2_b = p->fc2_b;
    c.cI = (float *)(ws + l.cI); c.c2f = (float *)(ws + l.c2f); c.c3f = (float *)(ws + l.c3f); c.c4f = (float *)(ws + l.c4f);
    c.seg_pos = d_pos; c.seg_lim = d_lim;
    c.B = B; c.T = T; c.hop = hop; c.NF = n_frames;
    HIPCHK(launch_cond(c, p->n_cus, stream));

    LoopArgs a;
    memset(&a, 0, sizeof a);
    a.I_w0 = p->I_w0; a.w_ih1 = p->w_ih1; a.w_hh1 = p->w_hh1; a.b_ih1 = p->b_ih1; a.b_hh1 = p->b_hh1;
    a.w_ih2 = p->w_ih2; a.w_hh2 = p->w_hh2; a.b_hh2 = p->b_hh2; a.fc1_w = p->fc1_w; a.fc2_w = p->fc2_w;
    a.fc3_w = p->fc3_w; a.fc3_b = p->fc3_b;
    a.w_ih1T = p->w_ih1T; a.w_hh1T = p->w_hh1T; a.w_ih2T = p->w_ih2T; a.w_hh2T = p->w_hh2T;
    a.fc1T = p->fc1T; a.fc2T = p->fc2T; a.fc3T = p->fc3T;
    a.cI = c.cI; a.c2f = c.c2f; a.c3f = c.c3f; a.c4f = c.c4f;
    a.noise = noise; a.force_x = dbg ? dbg->force_x : nullptr; a.out = out; a.dbg_logits = dbg ? dbg->logits : nullptr;
    a.gran = (u64 *)(ws + l.gran); a.status = (unsigned *)(ws + l.status);
    a.prof = nullptr;
    if (getenv("WRNN_PROF") && atoi(getenv("WRNN_PROF")) != 0) {
        a.prof = (u64 *)(ws + l.prof);
        HIPCHK(hipMemsetAsync(ws + l.prof, 0, (size_t)MAXWG * NPROF * sizeof(u64), stream));
    }
    a.seg_pos = d_pos; a.seg_lim = d_lim;
    a.Btot = B; a.T = T; a.hop = hop; a.NF = n_frames; a.C = p->C;
    a.NG = (B + SEG - 1) / SEG;

    // ---- kernel choice --------------------------------------------------------------------------------
    // sparse: block-sparse GRU pack, 8 XCD clusters x <= 2 groups in flight; pipe: MOL, 4 clusters x 2-3 groups in flight;
    // cluster: 1/2/4 clusters, one group in flight each (all groups in one launch); persist: the chip-wide kernel, one
    // launch per 16-segment group; stream: one workgroup per segment.
    // auto = sparse if the pack qualifies, else pipe when a cluster has more than one group to run (MOL), else cluster,
    // else (device too small / odd class count) stream.
    const bool shape_ok = (p->mode == WRNN_MODE_MOL) || (p->C == H);
    enum { K_STREAM, K_PERSIST, K_CLUSTER, K_PIPE, K_SPARSE } kind = K_STREAM;
    int U = 0, ncl = 0, G = 0;
    // auto: a pack whose GRU matrices are block-sparse runs on the block-sparse kernel (measured 1.7-2.3x the dense pipelined
    // kernel on the same weights, profiles/r01r_probe_block_sparse.json); dense packs never qualify (512 blocks per row)
    if (algo == WRNN_ALGO_SPARSE || (algo == WRNN_ALGO_AUTO && p->sp_nbp && sparse_clusters(p->n_cus) >= 1)) {
        const int scl = sparse_clusters(p->n_cus);
        if (!p->sp_nbp || scl < 1) {
            set_err("block-sparse kernel needs MOL, >= 32 CUs and GRU matrices with <= 64 surviving 16x1 blocks per block row "
                    "(this pack: up to %d)", p->sp_max_blocks);
            return WRNN_ERR_ARG;
        }
        const char *envg = getenv("WRNN_SPARSE_G");
        const int groups = (B + SEG - 1) / SEG;
        int g = envg ? atoi(envg) : 0;
        if (g < 1 || g > SPG) g = groups > scl ? 2 : 1;
        const int rounds = (groups + scl * g - 1) / (scl * g);
        const long ng = (long)rounds * scl * g;
        if ((double)rounds * T >= 4.0e9) { set_err("too many steps"); return WRNN_ERR_ARG; }
        kind = K_SPARSE; G = g; ncl = scl; U = 16;
        a.NG = ng < B ? (int)ng : B;
        a.sp_vals = p->sp_vals; a.sp_cols = p->sp_cols;
    }
    if (kind == K_SPARSE) {
        // chosen above
    } else if ((algo == WRNN_ALGO_AUTO || algo == WRNN_ALGO_PIPE) && p->mode == WRNN_MODE_MOL) {
        // Pipelined kernel: G groups in flight per cluster.  Measured step times on MI355X (profiles/r01g_*): single-depth
        // cluster kernel 25 us per round-step, G = 2 33.7 us, G = 3 (15-row groups) 49.2 us.  One launch costs
        // rounds x step: pick the cheapest depth; auto falls back to the cluster kernel when depth 1 wins.
        const int pcl = pipe_clusters(p->n_cus);
        const char *envg = getenv("WRNN_PIPE_G");
        int g = envg ? atoi(envg) : 0;
        if (pcl >= 1) {
            static const double step_us[MAXG + 1] = {0.0, 25.0, 33.7, 49.2};
            if (g < 1 || g > MAXG) {
                double best = 1e30;
                for (int c = 1; c <= MAXG; ++c) {
                    const int rows_c = pipe_rows(c);
                    const int rounds_c = ((B + rows_c - 1) / rows_c + pcl * c - 1) / (pcl * c);
                    const double cost = rounds_c * step_us[c];
                    if (cost < best - 1e-9) { best = cost; g = c; }
                }
                if (g == 1 && algo == WRNN_ALGO_AUTO) g = 0;             // the cluster kernel is the better depth-1 kernel
            }
            if (g >= 1) {
                const int rows = pipe_rows(g);
                const int groups = (B + rows - 1) / rows;
                const int rounds = (groups + pcl * g - 1) / (pcl * g);
                const long ng = (long)rounds * pcl * g;
                if ((double)rounds * T < 4.0e9) {
                    kind = K_PIPE; G = g; ncl = pcl; U = 8;
                    a.NG = ng < B ? (int)ng : B;
                }
            }
        }
        if (kind != K_PIPE && algo == WRNN_ALGO_PIPE) {
            set_err("pipelined kernel needs >= 64 CUs and MOL mode; device has %d CUs", p->n_cus);
            return WRNN_ERR_RESIDENCY;
        }
    } else if (algo == WRNN_ALGO_PIPE) {
        set_err("the pipelined kernel exists for MOL only");
        return WRNN_ERR_ARG;
    }
    if (kind == K_PIPE || kind == K_SPARSE) {
        // chosen above
    } else if (algo == WRNN_ALGO_AUTO || algo == WRNN_ALGO_CLUSTER) {
        if (shape_ok && (double)a.NG * T < 4.0e9) {
            const char *envu = getenv("WRNN_CLUSTER_U");
            const int want = envu ? atoi(envu) : 0;
            // default split: as many clusters as there are groups to keep busy (fewer, larger clusters have the
            // shorter per-step MFMA chain; more, smaller clusters run more groups at once)
            const int umax = (p->mode == WRNN_MODE_MOL) ? 8 : 4;          // the U = 8 split exists for MOL only
            // measured (profiles/r01d_probe_pipe_depths.json, B = 12): two clusters x 6 segments 18.5 us per step against 22.6 us
            // for the chip-wide split with all 12 in one group -- fewer granule rows per sweep win, so never start at U = 2
            const int first = a.NG >= 3 ? umax : 4;
            const int pref[3] = {first, 4, umax};
            if ((want == 2 || want == 4 || want == 8) && want <= umax) { U = want; ncl = cluster_count(U, p->n_cus); }
            for (int i = 0; i < 3 && ncl < 1; ++i) { U = pref[i]; ncl = cluster_count(U, p->n_cus); }
            if (ncl >= 1) {
                kind = K_CLUSTER;
                // balance: the launch takes `rounds` group-runs per cluster however the segments are cut, so cut
                // them into rounds*ncl groups (smaller groups = fewer granule rows per sweep, every cluster busy)
                const int rounds = (a.NG + ncl - 1) / ncl;
                a.NG = rounds * ncl < B ? rounds * ncl : B;
            }
        }
        if (kind != K_CLUSTER && algo == WRNN_ALGO_CLUSTER) {
            set_err("cluster kernel needs >= 64 CUs and (MOL or RAW with 512 classes); device has %d CUs, C=%d", p->n_cus, p->C);
            return WRNN_ERR_RESIDENCY;
        }
    } else if (algo == WRNN_ALGO_PERSIST) {
        const char *envu = getenv("WRNN_PERSIST_U");
        if (shape_ok) {
            if (envu && (atoi(envu) == 2 || atoi(envu) == 4)) U = atoi(envu);
            else if (p->n_cus >= H / 2) U = 2;
            else if (p->n_cus >= H / 4) U = 4;
        }
        if (U != 0 && p->n_cus < H / U) U = 0;
        if (U == 0) {
            set_err("persistent kernel needs >= 128 CUs and (MOL or RAW with 512 classes); device has %d CUs, C=%d", p->n_cus, p->C);
            return WRNN_ERR_RESIDENCY;
        }
        kind = K_PERSIST;
    } else if (algo != WRNN_ALGO_STREAM && algo != WRNN_ALGO_PIPE && algo != WRNN_ALGO_SPARSE) { set_err("unknown algo %d", algo); return WRNN_ERR_ARG; }

    if (kind == K_PIPE || kind == K_SPARSE) {
        HIPCHK(launch_noise_mol(noise, (float *)(ws + l.npre), (long)T * 11 * B, B, p->n_cus, stream));
        a.noise_pre = (const float *)(ws + l.npre);
    }
    HIPCHK(hipEventRecord(p->ev0, stream));
    if (kind == K_SPARSE) {
        p->last_kernel = "wrnn_sparse_kernel";
        p->last_U = U; p->last_ncl = ncl; p->last_G = G;
        HIPCHK(hipMemsetAsync(ws + l.gran, 0, GRAN_BYTES, stream));
        hipError_t e = launch_sparse(a, G, ncl, p->sp_nbp, stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_err("block-sparse cooperative launch failed: %s", hipGetErrorString(e));
            return e == hipErrorCooperativeLaunchTooLarge ? WRNN_ERR_RESIDENCY : WRNN_ERR_HIP;
        }
    } else if (kind == K_PIPE) {
        p->last_kernel = "wrnn_pipe_kernel";
        p->last_U = U; p->last_ncl = ncl; p->last_G = G;
        HIPCHK(hipMemsetAsync(ws + l.gran, 0, GRAN_BYTES, stream));
        const char *envn = getenv("WRNN_PIPE_NL");
        hipError_t e = launch_pipe(a, G, ncl, envn ? atoi(envn) : 16, stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_err("pipelined cooperative launch failed: %s", hipGetErrorString(e));
            return e == hipErrorCooperativeLaunchTooLarge ? WRNN_ERR_RESIDENCY : WRNN_ERR_HIP;
        }
    } else if (kind == K_CLUSTER) {
        p->last_kernel = "wrnn_cluster_kernel";
        p->last_U = U; p->last_ncl = ncl; p->last_G = 1;
        HIPCHK(hipMemsetAsync(ws + l.gran, 0, GRAN_BYTES, stream));
        const char *envn = getenv("WRNN_CLUSTER_NL");
        hipError_t e = launch_cluster(a, U, ncl, p->mode, envn ? atoi(envn) : 0, stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (algo == WRNN_ALGO_AUTO) {
                fprintf(stderr, "[wavernn_amd] cluster launch refused (%s); using the stream kernel\n", hipGetErrorString(e));
                kind = K_STREAM;
            } else {
                set_err("cluster cooperative launch failed: %s", hipGetErrorString(e));
                return e == hipErrorCooperativeLaunchTooLarge ? WRNN_ERR_RESIDENCY : WRNN_ERR_HIP;
            }
        }
    } else if (kind == K_PERSIST) {
        p->last_kernel = "wrnn_persist_kernel";
        p->last_U = U; p->last_ncl = 1; p->last_G = 1;
        for (int b0 = 0; b0 < B; b0 += SEG) {
            a.b0 = b0;
            a.nb = (B - b0 < SEG) ? (B - b0) : SEG;
            HIPCHK(hipMemsetAsync(ws + l.gran, 0, (size_t)NGRAN * SEG * H * sizeof(u64), stream));
            hipError_t e = launch_persist(a, U, p->mode, stream);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                set_err("persistent cooperative launch failed: %s", hipGetErrorString(e));
                return e == hipErrorCooperativeLaunchTooLarge ? WRNN_ERR_RESIDENCY : WRNN_ERR_HIP;
            }
        }
    }
    if (kind == K_STREAM) {
        p->last_kernel = "wrnn_stream_kernel";
        p->last_U = 0; p->last_ncl = 0; p->last_G = 0;
        a.b0 = 0;
        a.nb = B;
        HIPCHK(launch_stream(a, p->mode, stream));
    }
    HIPCHK(hipEventRecord(p->ev1, stream));
    p->timed = true;
    return WRNN_OK;
}

extern "C" int wrnn_generate(const wrnn_pack *pc, const wrnn_geometry *g, const float *mels_up, const float *aux,
                             const float *noise, float *out, void *workspace, size_t workspace_bytes, int algo,
                             const wrnn_debug *dbg, void *stream_)
{
    int rc = check_geometry(g);
    if (rc != WRNN_OK) return rc;
    std::vector<int32_t> pos(g->B), lim(g->B, g->L);
    for (int b = 0; b < g->B; ++b) pos[b] = b * g->stride;
    return wrnn_generate_segments(pc, g->B, g->T, pos.data(), lim.data(), g->L, g->hop, g->n_frames, mels_up, aux,
                                  noise, out, workspace, workspace_bytes, algo, dbg, stream_);
}

extern "C" int wrnn_status(void *workspace, void *stream)
{
    if (!workspace) { set_err("NULL workspace"); return WRNN_ERR_ARG; }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    unsigned st[STATUS_WORDS];
    HIPCHK(hipMemcpy(st, workspace, sizeof st, hipMemcpyDeviceToHost));
    if (st[0] != 0 || st[1] != 0) {
        set_err("loop kernel gave up: code 0x%x (layer %u) workgroup %u step %u thread %u", st[1], st[1] & 0xff, st[2], st[3], st[4]);
        return WRNN_ERR_KERNEL;
    }
    return WRNN_OK;
}

extern "C" int wrnn_profile_read(void *workspace, unsigned long long *out, int max_words, void *stream)
{
    if (!workspace || !out || max_words < 1) { set_err("bad argument"); return WRNN_ERR_ARG; }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    size_t n = (size_t)MAXWG * NPROF;
    if ((size_t)max_words < n) n = (size_t)max_words;
    const size_t off = (STATUS_WORDS * sizeof(unsigned) + 255) / 256 * 256;
    HIPCHK(hipMemcpy(out, (char *)workspace + off, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return (int)n;
}

extern "C" float wrnn_last_loop_ms(const wrnn_pack *p)
{
    if (!p || !p->timed) return -1.f;
    if (hipEventSynchronize(p->ev1) != hipSuccess) return -1.f;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, p->ev0, p->ev1) != hipSuccess) return -1.f;
    return ms;
}

extern "C" const char *wrnn_last_loop_kernel(const wrnn_pack *p) { return p ? p->last_kernel : ""; }

extern "C" int wrnn_last_loop_split(const wrnn_pack *p, int *units_per_wg, int *clusters, int *groups_in_flight)
{
    if (!p) { set_err("NULL pack"); return WRNN_ERR_ARG; }
    if (units_per_wg) *units_per_wg = p->last_U;
    if (clusters) *clusters = p->last_ncl;
    if (groups_in_flight) *groups_in_flight = p->last_G;
    return WRNN_OK;
}

static float g_selftest_metric = -1.f;
extern "C" float wrnn_selftest_metric(void) { return g_selftest_metric; }

extern "C" int wrnn_selftest(int device, int which)
{
    const int cus = wrnn_device_cus(device);
    if (cus < 0) return cus;
    HIPCHK(hipSetDevice(device));
    char msg[400] = "";
    int rc;
    if (which == 1) rc = selftest_mfma(msg, sizeof msg);
    else if (which == 2) rc = selftest_allgather(cus, msg, sizeof msg, &g_selftest_metric);
    else { set_err("unknown selftest %d", which); return WRNN_ERR_ARG; }
    if (rc != 0) { set_err("selftest %d failed: %s", which, msg); return WRNN_ERR_KERNEL; }
    set_err("selftest %d ok: %s", which, msg);
    return WRNN_OK;
}
