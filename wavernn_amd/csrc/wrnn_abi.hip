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
hipError_t launch_cond(const CondArgs &a, int n_cus, bool valu, hipStream_t stream);
hipError_t launch_cond_frames(const CondArgs &a, hipStream_t stream);
hipError_t launch_cond_frames_slab(const CondArgs &a, hipStream_t stream);
hipError_t launch_cond_frag(const CondArgs &a, int n_cus, hipStream_t stream);
hipError_t launch_noise_mol(const float *in, float *out, long n, int B, int n_cus, hipStream_t stream);
hipError_t launch_stream(const LoopArgs &args, int mode, hipStream_t stream);
hipError_t launch_loop(const LoopArgs &args, int ncl, int mode, hipStream_t stream);
int loop_max_depth(int mode);
int loop_clusters(int n_cus);
size_t loop_state_floats(int G);
hipError_t launch_generic(const GenArgs &args, int mode, hipStream_t stream);
bool generic_dims_ok(int H, int F, int M, int A, int C, int mode);
hipError_t launch_duo(const LoopArgs &args, int ncl, int mode, hipStream_t stream);
int duo_clusters(int n_cus);
int duo_max_depth();
size_t duo_xbuf_bytes(int G);
size_t duo_xbuf_bytes_max();
hipError_t launch_sparse(const LoopArgs &args, int nbp, hipStream_t stream);
int sparse_clusters(int n_cus);
size_t sparse_state_floats();
size_t sparse_xbuf_bytes();
hipError_t launch_put_floats(float *dst, const float *src, int n, hipStream_t stream);
hipError_t launch_chain(const LoopArgs &args, int mode, hipStream_t stream);
hipError_t launch_octo(const LoopArgs &args, int ncl, int mode, hipStream_t stream);
int octo_clusters(int n_cus);
int octo_max_depth();
int chain_clusters(int n_cus);
int chain_max_depth();
size_t chain_state_floats(int G);
size_t chain_xbuf_bytes(int G);
int selftest_mfma(char *msg, size_t n);
int selftest_allgather(int n_cus, char *msg, size_t n, float *us_per_round);
int selftest_tanh(char *msg, size_t n);
int selftest_xor(char *msg, size_t n);
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
    const float *fc3f;         // MOL: fc3.weight in A-fragment order (wrnn_duo.hip)
    const float *u1;           // MOL: rnn1.weight_ih . I.weight[:,0] [3H] -- the x_{t-1} term of rnn1's gi (wrnn_duo.hip)
    // dimension-generic pack (any hparams but the shipped ones): only the k-major copies + biases, run by wrnn_generic_kernel
    bool generic;
    int gH, gF, gM, gA;
    const float *g_I_T, *g_fc1T, *g_fc2T, *g_w_ih2T;      // (the other k-major copies are the fields above)
    int sp_nbp;                // 0 = the GRU matrices are not block-sparse enough for wrnn_sparse_kernel; else 48 / 64
    int sp_max_blocks;
    const float *sp_vals;
    const int *sp_cols;
    int sp_fc_max_blocks;      // fc1 / fc2 (columns [0, H): the aux columns live in the per-frame tables): largest block row
    const float *sp_fc_vals;   // non-null: the Linear layers are block-sparse too (the notebook prunes them as well) -> wrnn_sparse_kernel's gathered fc stages
    const int *sp_fc_cols;
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
    const bool shipped = w->rnn_dims == H && w->fc_dims == H && w->feat_dims == MEL && w->aux_dims == AUX;
    const int C = w->n_classes;
    if (w->mode == WRNN_MODE_MOL) {
        if (C != 30) { set_err("MOL needs n_classes == 30"); return WRNN_ERR_ARG; }
    } else if (w->mode == WRNN_MODE_RAW) {
        if (shipped && (C < 2 || C > H)) { set_err("RAW needs 2 <= n_classes <= 512 with the shipped dims"); return WRNN_ERR_ARG; }
    } else { set_err("unknown mode %d", w->mode); return WRNN_ERR_ARG; }
    if (!shipped && !generic_dims_ok(w->rnn_dims, w->fc_dims, w->feat_dims, w->aux_dims, C, w->mode)) {
        set_err("unsupported dims rnn=%d fc=%d feat=%d aux=%d classes=%d: the generic kernel takes rnn, fc, classes <= 2048, feat + aux <= 1024 "
                "(MOL: 30 classes)", w->rnn_dims, w->fc_dims, w->feat_dims, w->aux_dims, C);
        return WRNN_ERR_ARG;
    }
    const float *const *ptrs = &w->I_w;
    for (int i = 0; i < 16; ++i)
        if (!ptrs[i]) { set_err("NULL weight pointer #%d", i); return WRNN_ERR_ARG; }
    const int cus = wrnn_device_cus(device);
    if (cus < 0) return cus;
    DeviceGuard dg(device);
    HIPCHK(dg.err);

    if (!shipped) {
        // ---- generic pack: k-major copies of the eight matrices + the biases (wrnn_generic.hip reads nothing else)
        const int gH = w->rnn_dims, gF = w->fc_dims, gM = w->feat_dims, gA = w->aux_dims;
        Builder b;
        const size_t oI = b.add_T(w->I_w, gH, 1 + gM + gA, 0, 1 + gM + gA), oIb = b.add(w->I_b, gH);
        const size_t o1i = b.add_T(w->w_ih1, 3 * gH, gH, 0, gH), o1h = b.add_T(w->w_hh1, 3 * gH, gH, 0, gH);
        const size_t o1bi = b.add(w->b_ih1, 3 * gH), o1bh = b.add(w->b_hh1, 3 * gH);
        const size_t o2i = b.add_T(w->w_ih2, 3 * gH, gH + gA, 0, gH + gA), o2h = b.add_T(w->w_hh2, 3 * gH, gH, 0, gH);
        const size_t o2bi = b.add(w->b_ih2, 3 * gH), o2bh = b.add(w->b_hh2, 3 * gH);
        const size_t of1 = b.add_T(w->fc1_w, gF, gH + gA, 0, gH + gA), of1b = b.add(w->fc1_b, gF);
        const size_t of2 = b.add_T(w->fc2_w, gF, gF + gA, 0, gF + gA), of2b = b.add(w->fc2_b, gF);
        const size_t of3 = b.add_T(w->fc3_w, C, gF, 0, gF), of3b = b.add(w->fc3_b, C);
        b.add(nullptr, 64);
        wrnn_pack *p = new wrnn_pack();
        memset(p, 0, sizeof *p);
        p->device = device; p->n_cus = cus; p->C = C; p->mode = w->mode; p->generic = true;
        p->gH = gH; p->gF = gF; p->gM = gM; p->gA = gA;
        p->dev_bytes = b.host.size() * sizeof(float);
        p->weight_bytes = sizeof(float) * ((size_t)gH * (1 + gM + gA) + gH + 2 * ((size_t)3 * gH * gH) + (size_t)3 * gH * (gH + gA) + (size_t)3 * gH * gH +
                                           4 * 3 * gH + (size_t)gF * (gH + gA) + (size_t)gF * (gF + gA) + 2 * gF + (size_t)C * gF + C);
        hipError_t e = hipMalloc((void **)&p->dev, p->dev_bytes);
        if (e != hipSuccess) { set_err("hipMalloc(%zu) failed: %s", p->dev_bytes, hipGetErrorString(e)); delete p; return WRNN_ERR_HIP; }
        e = hipMemcpy(p->dev, b.host.data(), p->dev_bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { set_err("hipMemcpy failed: %s", hipGetErrorString(e)); (void)hipFree(p->dev); delete p; return WRNN_ERR_HIP; }
        const float *base = (const float *)p->dev;
        p->g_I_T = base + oI; p->I_b = base + oIb;
        p->w_ih1T = base + o1i; p->w_hh1T = base + o1h; p->b_ih1 = base + o1bi; p->b_hh1 = base + o1bh;
        p->g_w_ih2T = base + o2i; p->w_hh2T = base + o2h; p->b_ih2 = base + o2bi; p->b_hh2 = base + o2bh;
        p->g_fc1T = base + of1; p->fc1_b = base + of1b; p->g_fc2T = base + of2; p->fc2_b = base + of2b;
        p->fc3T = base + of3; p->fc3_b = base + of3b;
        p->sp_max_blocks = gH;
        *out = p;
        return WRNN_OK;
    }

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
    // MOL: fc3 (30 x 512) as two 16-row MFMA A tiles in fragment order: [tile][wave][k-block r][lane (row fi, k-quad kq)][4]
    // = fc3_w[16 tile + fi][128 wave + 16 r + 4 kq ..], rows >= 30 zero
    size_t o_fc3f = 0;
    if (w->mode == WRNN_MODE_MOL) {
        o_fc3f = b.add(nullptr, (size_t)2 * SEG * H);
        for (int q = 0; q < 2 * SEG * H / 4; ++q) {
            const int l6 = q & 63, r = (q >> 6) & 7, wv = (q >> 9) & 3, tile = q >> 11;
            const int row = 16 * tile + (l6 & 15), kq = l6 >> 4;
            for (int e = 0; e < 4; ++e)
                b.host[o_fc3f + (size_t)q * 4 + e] = row < C ? w->fc3_w[(size_t)row * H + 128 * wv + 16 * r + 4 * kq + e] : 0.f;
        }
    }
    // u1 = rnn1.weight_ih . I.weight[:,0] (double accumulation, rounded once): gi = W_ih . (cI + w0 x) + b = W_ih . cI + x u1 + b
    size_t o_u1 = 0;
    {
        o_u1 = b.add(nullptr, (size_t)3 * H);
        for (int r = 0; r < 3 * H; ++r) {
            double acc = 0.0;
            for (int k = 0; k < H; ++k) acc += (double)w->w_ih1[(size_t)r * H + k] * (double)col0[k];
            b.host[o_u1 + r] = (float)acc;
        }
    }
    // ---- block-sparse view of the GRU matrices (16x1 blocks: 16 consecutive rows of one gate x 1 column) -----------
    // usable by wrnn_sparse_kernel when every block row keeps <= 64 columns (~5 % density keeps ~26 +- 5)
    int sp_nbp = 0, sp_max = 0, sp_fc_max = 0;
    bool sp_fc_ok = false;
    size_t o_spv = 0, o_spc = 0, o_sfv = 0, o_sfc = 0;
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
        // fc1 / fc2 (round 6; "Pruning - Scratchpad.ipynb" :199-204 prunes the Linear layers too): their first H columns, 32 block rows each
        const float *fmats[2] = {w->fc1_w, w->fc2_w};
        std::vector<std::vector<int>> fcols((size_t)2 * 32);
        for (int m = 0; m < 2; ++m)
            for (int wg = 0; wg < 32; ++wg) {
                std::vector<int> &cv = fcols[(size_t)m * 32 + wg];
                const float *base = fmats[m] + (size_t)(16 * wg) * K2;
                for (int c = 0; c < H; ++c) {
                    bool nz = false;
                    for (int r = 0; r < 16 && !nz; ++r) nz = base[(size_t)r * K2 + c] != 0.0f;
                    if (nz) cv.push_back(c);
                }
                if ((int)cv.size() > sp_fc_max) sp_fc_max = (int)cv.size();
            }
        if (sp_max <= 64 && w->mode == WRNN_MODE_MOL) {
            sp_nbp = sp_max <= 48 ? 48 : 64;
            if (sp_fc_max <= 64) {                        // one NBP for the gate and the fc tiles of a launch
                if (sp_fc_max > 48) sp_nbp = 64;
                o_sfv = b.add(nullptr, (size_t)2 * 32 * sp_nbp * 16);
                o_sfc = b.add(nullptr, (size_t)2 * 32 * sp_nbp);
                for (size_t q = 0; q < (size_t)2 * 32 * sp_nbp * 16; ++q) b.host[o_sfv + q] = 0.f;
                int *fi_ = reinterpret_cast<int *>(b.host.data() + o_sfc);
                for (size_t q = 0; q < (size_t)2 * 32 * sp_nbp; ++q) fi_[q] = 0;
                for (int m = 0; m < 2; ++m)
                    for (int wg = 0; wg < 32; ++wg) {
                        const size_t br = (size_t)m * 32 + wg;
                        const std::vector<int> &cv = fcols[br];
                        const float *base = fmats[m] + (size_t)(16 * wg) * K2;
                        for (size_t k = 0; k < cv.size(); ++k) {
                            reinterpret_cast<int *>(b.host.data() + o_sfc)[br * sp_nbp + k] = cv[k];
                            for (int r = 0; r < 16; ++r) b.host[o_sfv + (br * sp_nbp + k) * 16 + r] = base[(size_t)r * K2 + cv[k]];
                        }
                    }
                sp_fc_ok = true;
            }
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
    p->fc3f = w->mode == WRNN_MODE_MOL ? base + o_fc3f : nullptr;
    p->u1 = base + o_u1;
    p->sp_nbp = sp_nbp; p->sp_max_blocks = sp_max;
    p->sp_vals = sp_nbp ? base + o_spv : nullptr;
    p->sp_cols = sp_nbp ? reinterpret_cast<const int *>(base + o_spc) : nullptr;
    p->sp_fc_max_blocks = sp_fc_max;
    p->sp_fc_vals = sp_fc_ok ? base + o_sfv : nullptr;
    p->sp_fc_cols = sp_fc_ok ? reinterpret_cast<const int *>(base + o_sfc) : nullptr;
    *out = p;
    return WRNN_OK;
}

extern "C" void wrnn_pack_destroy(wrnn_pack *p)
{
    if (!p) return;
    (void)hipFree(p->dev);
    delete p;
}

extern "C" size_t wrnn_pack_weight_bytes(const wrnn_pack *p) { return p ? p->weight_bytes : 0; }

extern "C" int wrnn_pack_sparse_blocks(const wrnn_pack *p) { return p ? (p->sp_nbp ? p->sp_max_blocks : -p->sp_max_blocks) : 0; }
extern "C" int wrnn_pack_sparse_fc_blocks(const wrnn_pack *p) { return p ? (p->sp_fc_vals ? p->sp_fc_max_blocks : -p->sp_fc_max_blocks) : 0; }

struct wrnn_timer {
    int device;
    std::vector<hipEvent_t> ev;     // pairs (start, stop), grown on demand and reused
    int used;                       // events recorded by the last call
};

extern "C" int wrnn_timer_create(int device, wrnn_timer **out)
{
    if (!out) { set_err("NULL argument"); return WRNN_ERR_ARG; }
    const int cus = wrnn_device_cus(device);
    if (cus < 0) return cus;
    wrnn_timer *t = new wrnn_timer();
    t->device = device;
    t->used = 0;
    *out = t;
    return WRNN_OK;
}

extern "C" void wrnn_timer_destroy(wrnn_timer *t)
{
    if (!t) return;
    for (hipEvent_t e : t->ev) (void)hipEventDestroy(e);
    delete t;
}

extern "C" int wrnn_timer_launches(const wrnn_timer *t) { return t ? t->used / 2 : 0; }

extern "C" float wrnn_timer_ms(wrnn_timer *t)
{
    if (!t || t->used < 2) return -1.f;
    float total = 0.f;
    for (int i = 0; i + 1 < t->used; i += 2) {
        if (hipEventSynchronize(t->ev[i + 1]) != hipSuccess) return -1.f;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t->ev[i], t->ev[i + 1]) != hipSuccess) return -1.f;
        total += ms;
    }
    return total;
}

namespace {
int timer_mark(wrnn_timer *t, hipStream_t stream)
{
    if (!t) return WRNN_OK;
    if (t->used == (int)t->ev.size()) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        t->ev.push_back(e);
    }
    HIPCHK(hipEventRecord(t->ev[t->used], stream));
    t->used += 1;
    return WRNN_OK;
}

// progress read-out (wrnn_options.progress): a host function enqueued behind the launches of a slab
struct ProgressCtx {
    void (*fn)(int32_t, int32_t, int32_t, void *);
    void *user;
    int32_t done, T, n;
};
void progress_trampoline(void *p)
{
    ProgressCtx *c = static_cast<ProgressCtx *>(p);
    c->fn(c->done, c->T, c->n, c->user);
    delete c;
}
int enqueue_progress(const wrnn_options *o, int done, int T, int n, hipStream_t stream)
{
    if (!o->progress) return WRNN_OK;
    ProgressCtx *c = new ProgressCtx{o->progress, o->progress_user, done, T, n};
    hipError_t e = hipLaunchHostFunc(stream, progress_trampoline, c);
    if (e != hipSuccess) {
        delete c;
        set_err("hipLaunchHostFunc failed: %s", hipGetErrorString(e));
        return WRNN_ERR_HIP;
    }
    return WRNN_OK;
}

enum Kind { K_STREAM, K_LOOP, K_SPARSE, K_DUO, K_GENERIC, K_CHAIN };
constexpr int DUO_TAB_FPS = 8;       // wrnn_duo_kernel: rows per segment of the per-slab aux tables: a slab covers at most (DUO_TAB_FPS - 2) hops + 1 steps
constexpr bool DUO_AUTO = true;      // `auto` runs MoL on wrnn_duo_kernel at every depth (round 4, profiles/r04a_probe_new.json: 13.5 vs 16.6 us per step
constexpr int CHAIN_AUTO_GROUPS = 8;   // `auto` runs calls of up to this many groups (128 segments) on wrnn_chain_kernel: one / two groups per cluster, 10.4 / 13.8 us (RAW: 12.8 / 16.3)
                                       // per step against wrnn_duo_kernel's 12.2 / 16.9; from three groups on the duo kernel wins (profiles/r05i_probe_chain_depths.json)
constexpr bool OCTO_AUTO = false;    // `auto` runs dense MoL batches beyond wrnn_chain_kernel's range on wrnn_octo_kernel (round 6)
constexpr int DUO_MIN_DEPTH = 1;     // with one group in flight, 17.9 vs 21.7 with two, 26.2 vs 36 with four; round 3's kernel paid off from depth 3 on)

// what a call will run: kernel, split, rounds, slab length
struct Plan {
    Kind kind;
    int ncl, G, rounds, per_round, slab, ngr_max;
    int tab_fps;                        // K_DUO: rows per segment of the per-slab aux tables
    int t0, t1;
    bool octo;                          // K_DUO planned onto wrnn_octo_kernel (one 512-thread workgroup per CU: matrix + service waves; MOL): the same split,
                                        // workspace, exchange buffer and state layout
};

struct WsLayout {
    size_t status, xcc, segs, melc, c2f, c3f, c4f;
    size_t cI, npre;                    // stream kernel: whole-T conditioning; persistent kernels: one slab of derived MOL noise
    size_t xbuf, state, cIf;            // loop kernel: exchange buffer, per-round state, conditioning slab
    size_t total;
};
size_t al(size_t x) { return (x + 255) / 256 * 256; }

const wrnn_options *norm_options(const wrnn_options *opt, wrnn_options *tmp)
{
    memset(tmp, 0, sizeof *tmp);
    if (opt) {
        size_t n = opt->struct_bytes > 0 ? (size_t)opt->struct_bytes : sizeof *tmp;
        if (n > sizeof *tmp) n = sizeof *tmp;
        memcpy(tmp, opt, n);
    }
    tmp->struct_bytes = (int32_t)sizeof *tmp;
    return tmp;
}

// kernel choice: the block-sparse kernel if the pack qualifies (MOL, every 16-row block row of the GRU matrices keeps <= 64 columns) on a
// 256-CU device, else the two-workgroups-per-CU loop kernel (RAW with 512 classes or MOL, >= 128 CUs), else the one-per-CU loop kernel,
// else stream
int make_plan(const wrnn_pack *p, int B, int T, const wrnn_options *o, Plan *pl)
{
    const bool shape_ok = (p->mode == WRNN_MODE_MOL) || (p->C == H);
    pl->t0 = o->t_begin; pl->t1 = o->t_end;
    if (pl->t0 == 0 && pl->t1 == 0) pl->t1 = T;
    if (pl->t0 < 0 || pl->t1 > T || pl->t0 >= pl->t1) { set_err("bad step range [%d, %d) of T=%d", pl->t0, pl->t1, T); return WRNN_ERR_ARG; }
    const int algo = o->algo;
    pl->octo = false;
    if (algo != WRNN_ALGO_AUTO && algo != WRNN_ALGO_STREAM && algo != WRNN_ALGO_LOOP && algo != WRNN_ALGO_SPARSE && algo != WRNN_ALGO_DUO && algo != WRNN_ALGO_CHAIN &&
        algo != WRNN_ALGO_OCTO) {
        set_err("unknown algo %d", algo);
        return WRNN_ERR_ARG;
    }
    const int groups = (B + SEG - 1) / SEG;
    pl->kind = K_STREAM; pl->ncl = 0; pl->G = 0; pl->rounds = 1; pl->per_round = B; pl->slab = T; pl->ngr_max = groups;
    if (p->generic) {           // non-shipped hparams: the dimension-generic kernel is the only one that takes them
        if (algo != WRNN_ALGO_AUTO && algo != WRNN_ALGO_STREAM) {
            set_err("this pack has non-shipped dims (rnn %d, fc %d, feat %d, aux %d): only the generic kernel (algo auto / stream) runs it", p->gH, p->gF, p->gM, p->gA);
            return WRNN_ERR_ARG;
        }
        if (pl->t0 != 0 || pl->t1 != T) { set_err("a partial step range needs a persistent loop kernel"); return WRNN_ERR_ARG; }
        pl->kind = K_GENERIC;
        return WRNN_OK;
    }
    // wrnn_chain_kernel (MOL and 9-bit RAW, 256 CUs): one workgroup per CU, one instruction stream per wave.  `auto`: <= 128 segments -- <= 64 (one utterance
    // of BASELINE config 2 / 3): one group per 64-CU cluster, a step is the latency of one chain; <= 128: two groups per cluster --; on
    // request (algo = chain) also with up to 4 groups in flight per cluster (wrnn_options.depth) and rounds beyond that
    const int ccl = chain_clusters(p->n_cus);
    const bool chain_hw = shape_ok && ccl >= 1;
    if (algo == WRNN_ALGO_CHAIN && !chain_hw) {
        set_err("wrnn_chain_kernel needs MOL or RAW with 512 classes, and >= 256 CUs (C = %d, device: %d CUs)", p->C, p->n_cus);
        return !shape_ok ? WRNN_ERR_ARG : WRNN_ERR_RESIDENCY;
    }
    if (algo == WRNN_ALGO_CHAIN || (algo == WRNN_ALGO_AUTO && chain_hw && groups <= CHAIN_AUTO_GROUPS && !p->sp_nbp)) {
        const int gmax = chain_max_depth();
        int g = o->depth;
        if (g < 1 || g > gmax) {
            const int rounds = (groups + ccl * gmax - 1) / (ccl * gmax);
            g = (groups + ccl * rounds - 1) / (ccl * rounds);
            if (g < 1) g = 1;
            if (g > gmax) g = gmax;
        }
        pl->kind = K_CHAIN; pl->ncl = ccl; pl->G = g;
        pl->rounds = (groups + ccl * g - 1) / (ccl * g);
        pl->per_round = (B + pl->rounds - 1) / pl->rounds;
        pl->ngr_max = (pl->per_round + SEG - 1) / SEG;
        if (pl->ngr_max > ccl * g) { pl->rounds += 1; pl->per_round = (B + pl->rounds - 1) / pl->rounds; pl->ngr_max = (pl->per_round + SEG - 1) / SEG; }
        int slab = o->slab_steps;
        if (slab < 1) {                      // (an explicit slab length is taken as given -- short slabs included: tests -- as the duo branch does)
            slab = p->mode == WRNN_MODE_MOL ? (int)((32u << 20) / ((size_t)pl->per_round * 11 * sizeof(float) * pl->rounds)) : 4096;
            if (slab < 16) slab = 16;
            if (slab > 4096) slab = 4096;
        }
        if (slab > T) slab = T;
        pl->slab = slab;
        pl->tab_fps = DUO_TAB_FPS;
    }
    // a block-sparse pack runs on wrnn_sparse_kernel (round 5: 16 clusters of 16 CUs, one group of 16 segments each -- the step is the
    // latency of one chain, and sixteen chains run side by side): `auto` picks it whenever the pack and the device qualify
    const int scl = sparse_clusters(p->n_cus);
    if (algo == WRNN_ALGO_SPARSE && (!p->sp_nbp || scl < 1)) {
        set_err("block-sparse kernel needs MOL, >= 256 CUs and GRU matrices with <= 64 surviving 16x1 blocks per block row "
                "(this pack: up to %d; device: %d CUs)", p->sp_max_blocks, p->n_cus);
        return (!p->sp_nbp) ? WRNN_ERR_ARG : WRNN_ERR_RESIDENCY;
    }
    if (pl->kind == K_CHAIN) {
        // (planned above)
    } else if (algo == WRNN_ALGO_SPARSE || (algo == WRNN_ALGO_AUTO && p->sp_nbp && scl >= 1)) {
        pl->kind = K_SPARSE; pl->ncl = scl; pl->G = 1;
        pl->rounds = (groups + scl - 1) / scl;
        pl->per_round = (B + pl->rounds - 1) / pl->rounds;
        pl->ngr_max = (pl->per_round + SEG - 1) / SEG;
        if (pl->ngr_max > scl) { pl->rounds += 1; pl->per_round = (B + pl->rounds - 1) / pl->rounds; pl->ngr_max = (pl->per_round + SEG - 1) / SEG; }
        int slab = o->slab_steps;
        if (slab < 1) {
            slab = (int)((32u << 20) / ((size_t)pl->per_round * 11 * sizeof(float) * pl->rounds));      // one slab of derived noise
            if (slab < 16) slab = 16;
            if (slab > 4096) slab = 4096;
        }
        if (slab > T) slab = T;
        pl->slab = slab;
        pl->tab_fps = DUO_TAB_FPS;
    } else if (algo == WRNN_ALGO_AUTO || algo == WRNN_ALGO_LOOP || algo == WRNN_ALGO_DUO || algo == WRNN_ALGO_OCTO) {
        int ncl = loop_clusters(p->n_cus);
        if (algo == WRNN_ALGO_OCTO && (p->mode != WRNN_MODE_MOL || octo_clusters(p->n_cus) != MAXCL)) {
            set_err("wrnn_octo_kernel needs MOL and >= 256 CUs (mode %d, device has %d CUs)", p->mode, p->n_cus);
            return p->mode != WRNN_MODE_MOL ? WRNN_ERR_ARG : WRNN_ERR_RESIDENCY;
        }
        if (algo == WRNN_ALGO_DUO && (!shape_ok || duo_clusters(p->n_cus) < 1)) {
            set_err("the two-workgroups-per-CU loop kernel needs MOL or RAW with 512 classes, and >= 64 CUs (C = %d, device has %d CUs)", p->C, p->n_cus);
            return !shape_ok ? WRNN_ERR_ARG : WRNN_ERR_RESIDENCY;
        }
        if (shape_ok && ncl >= 1) {
            if (o->clusters == 1 || o->clusters == 2 || o->clusters == 4) ncl = o->clusters < ncl ? o->clusters : ncl;
            else if (groups < ncl && !(DUO_AUTO && algo != WRNN_ALGO_LOOP && ncl == MAXCL)) {
                // no more clusters than groups (rounded up to 1, 2, 4).  Not for the duo kernel: its grid is always 4 clusters (a cluster
                // without a group leaves at once), so that a small batch sits on whole XCDs exactly as a large one does
                int c2 = 1; while (c2 < groups) c2 *= 2; if (c2 < ncl) ncl = c2;
            }
            const int gmax = algo == WRNN_ALGO_OCTO ? octo_max_depth() : loop_max_depth(p->mode);      // (wrnn_octo_kernel: what its LDS carve holds)
            int g = o->depth;
            if (g < 1 || g > gmax) {
                // as deep as the segments fill evenly: rounds = ceil(groups / (ncl * gmax)), then the shallowest depth that
                // still needs only that many rounds (deeper pipelines hide the exchange latency; empty slots cost nothing)
                const int rounds = (groups + ncl * gmax - 1) / (ncl * gmax);
                g = (groups + ncl * rounds - 1) / (ncl * rounds);
                if (g < 1) g = 1;
                if (g > gmax) g = gmax;
            }
            pl->kind = K_LOOP; pl->ncl = ncl; pl->G = g;
            // the two-workgroups-per-CU form (MOL): on request, or when `auto` has >= DUO_MIN_DEPTH groups in flight per cluster
            // (busy time bounds a step there; with fewer the latency of a slot's chain does, and the duo kernel's chain is one hop longer)
            if (algo == WRNN_ALGO_DUO || algo == WRNN_ALGO_OCTO || (algo == WRNN_ALGO_AUTO && DUO_AUTO && g >= DUO_MIN_DEPTH && ncl == MAXCL))
                pl->kind = K_DUO;
            pl->octo = pl->kind == K_DUO && (algo == WRNN_ALGO_OCTO || (algo == WRNN_ALGO_AUTO && OCTO_AUTO && p->mode == WRNN_MODE_MOL && ncl == MAXCL && !o->clusters));
            pl->rounds = (groups + ncl * g - 1) / (ncl * g);
            // balanced rounds of whole segments; every round is cut into <= ncl * g groups of <= 16
            pl->per_round = (B + pl->rounds - 1) / pl->rounds;
            pl->ngr_max = (pl->per_round + SEG - 1) / SEG;
            if (pl->ngr_max > ncl * g) { pl->rounds += 1; pl->per_round = (B + pl->rounds - 1) / pl->rounds; pl->ngr_max = (pl->per_round + SEG - 1) / SEG; }
            int slab = o->slab_steps;
            if (slab < 1) {
                // what a slab holds: wrnn_loop_kernel -- the hoisted conditioning cI (2 KB per segment-step) + the derived MoL noise;
                // wrnn_duo_kernel forms cI in the loop: only the derived noise (44 B per segment-step)
                if (pl->kind == K_DUO) slab = p->mode == WRNN_MODE_MOL ? (int)((32u << 20) / ((size_t)pl->per_round * 11 * sizeof(float) * pl->rounds)) : 4096;
                else slab = (int)((96u << 20) / ((size_t)pl->ngr_max * SEG * H * sizeof(float)));
                if (slab < 16) slab = 16;
                if (slab > (pl->kind == K_DUO ? 4096 : 1024)) slab = pl->kind == K_DUO ? 4096 : 1024;
            }
            if (slab > T) slab = T;
            pl->slab = slab;
            pl->tab_fps = DUO_TAB_FPS;
        } else if (algo == WRNN_ALGO_LOOP) {
            set_err("the loop kernel needs >= 64 CUs and (MOL or RAW with 512 classes); device has %d CUs, C=%d", p->n_cus, p->C);
            return WRNN_ERR_RESIDENCY;
        }
    }
    if (pl->kind != K_LOOP && pl->kind != K_DUO && pl->kind != K_SPARSE && pl->kind != K_CHAIN && (pl->t0 != 0 || pl->t1 != T)) {
        set_err("a partial step range [%d, %d) needs a persistent loop kernel", pl->t0, pl->t1);
        return WRNN_ERR_ARG;
    }
    // the kernels that form their conditioning in the loop index the per-segment, per-slab aux tables with 32-bit byte offsets inside one buffer
    // resource: refused HERE, so that wrnn_plan_segments / wrnn_workspace_bytes_segments report it and nothing has been queued when a call fails
    if ((pl->kind == K_DUO || pl->kind == K_SPARSE || pl->kind == K_CHAIN) && ((size_t)B * pl->tab_fps + 1) * 3 * H * sizeof(float) >= 0x7FFFF000ull) {
        set_err("%d segments in one call: the per-slab aux tables exceed a 2 GB buffer resource; split the call (generate_corpus caps a launch at 4096 segments)", B);
        return WRNN_ERR_ARG;
    }
    return WRNN_OK;
}

WsLayout ws_layout(const wrnn_pack *p, const Plan &pl, int B, int T, int n_frames)
{
    WsLayout l;
    memset(&l, 0, sizeof l);
    size_t o = 0;
    l.status = o; o = al(o + STATUS_WORDS * sizeof(unsigned));
    l.xcc = o;    o = al(o + XCC_WORDS * sizeof(unsigned));
    l.segs = o;   o = al(o + (size_t)3 * B * sizeof(int));       // positions | limits | mel offsets (wrnn_options.mel_stage)
    l.melc = o;   o = al(o + (size_t)3 * LAST_SCALE * sizeof(float));
    if (pl.kind == K_GENERIC) { l.total = o; return l; }
    // per-frame aux tables: one row per frame of the call's conditioning (+ the zero row) -- or, for wrnn_duo_kernel, per SEGMENT and
    // slab: (slab - 1) / hop + 2 rows per segment (+ the zero row), refilled for every slab: independent of the corpus' length
    const bool slabbed = pl.kind == K_DUO || pl.kind == K_SPARSE || pl.kind == K_CHAIN;       // conditioning formed in the loop, per-segment aux tables per slab
    const size_t tab_rows = slabbed ? (size_t)B * pl.tab_fps + 1 : (size_t)n_frames + 1;
    l.c2f = o;    o = al(o + tab_rows * 3 * H * sizeof(float));
    l.c3f = o;    o = al(o + tab_rows * H * sizeof(float));
    l.c4f = o;    o = al(o + tab_rows * H * sizeof(float));
    const bool mol = p->mode == WRNN_MODE_MOL;
    if (pl.kind == K_LOOP || slabbed) {
        // (the exchange regions a kernel touches: wrnn_chain_kernel G x 4, wrnn_sparse_kernel 16, wrnn_duo_kernel up to 8 x 4 -- its launches may differ in depth)
        l.xbuf = o;  o = al(o + (pl.kind == K_CHAIN ? chain_xbuf_bytes(pl.G) : pl.kind == K_SPARSE ? sparse_xbuf_bytes() : slabbed ? duo_xbuf_bytes_max() : XBUF_FLOATS * sizeof(float)));
        l.state = o; o = al(o + (size_t)pl.rounds * (pl.kind == K_SPARSE ? sparse_state_floats() : pl.kind == K_CHAIN ? chain_state_floats(pl.G) : loop_state_floats(pl.G)) * sizeof(float));
        l.cIf = o;   if (pl.kind == K_LOOP) o = al(o + (size_t)pl.slab * pl.ngr_max * SEG * H * sizeof(float));      // (the duo kernel forms cI in the loop)
        l.npre = o;  if (mol) o = al(o + (size_t)pl.slab * 11 * B * sizeof(float));      // derived MOL noise of one slab
    } else {
        l.cI = o;    o = al(o + (size_t)T * B * H * sizeof(float));
        l.npre = o;
    }
    l.total = o;
    return l;
}
// p / hop == __umulhi(p, magic) >> shift for every 0 <= p < 2^31 (Granlund-Montgomery, N = 31: magic = ceil(2^(31 + s) / hop) with
// s = ceil(log2 hop) fits 32 bits and its error term is <= 2^s); checked below on the multiples of hop and their neighbours.  magic = 0:
// the kernel divides.
void hop_magic(int hop, unsigned *magic, int *shift)
{
    *magic = 0u; *shift = 0;
    if (hop < 2) return;
    int s = 0;
    while ((1ll << s) < hop) ++s;
    const unsigned long long m = ((1ull << (31 + s)) + (unsigned long long)hop - 1ull) / (unsigned long long)hop;
    if (m >> 32) return;
    const int sh = s - 1;
    for (long long q = 0; q * hop < (1ll << 31); q += 997) {            // spot check (the bound is a theorem; this guards the arithmetic)
        for (long long p = q * hop - 1; p <= q * hop + 1; ++p) {
            if (p < 0 || p >= (1ll << 31)) continue;
            if ((long long)(((unsigned long long)p * m) >> (32 + sh)) != p / hop) return;
        }
    }
    *magic = (unsigned)m; *shift = sh;
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

extern "C" size_t wrnn_workspace_bytes_segments(const wrnn_pack *p, int32_t n_segments, int32_t T, int32_t n_frames,
                                                const wrnn_options *opt)
{
    if (!p || n_segments < 1 || T < 1 || n_frames < 1) return 0;
    wrnn_options tmp;
    const wrnn_options *o = norm_options(opt, &tmp);
    Plan pl;
    wrnn_options whole = *o;                 // the workspace of a partial-range call is the whole call's
    whole.t_begin = 0; whole.t_end = 0;
    if (make_plan(p, n_segments, T, &whole, &pl) != WRNN_OK) return 0;
    return ws_layout(p, pl, n_segments, T, n_frames).total;
}

extern "C" int wrnn_plan_segments(const wrnn_pack *p, int32_t n_segments, int32_t T, const wrnn_options *opt, wrnn_run_info *out)
{
    if (!p || !out || n_segments < 1 || T < 1) { set_err("bad argument"); return WRNN_ERR_ARG; }
    wrnn_options tmp;
    wrnn_options whole = *norm_options(opt, &tmp);
    whole.t_begin = 0; whole.t_end = 0;
    Plan pl;
    int rc = make_plan(p, n_segments, T, &whole, &pl);
    if (rc != WRNN_OK) return rc;
    memset(out, 0, sizeof *out);
    out->kernel = pl.kind == K_GENERIC ? "wrnn_generic_kernel" : pl.kind == K_CHAIN ? "wrnn_chain_kernel" : pl.kind == K_DUO ? (pl.octo ? "wrnn_octo_kernel" : "wrnn_duo_kernel") : pl.kind == K_LOOP ? "wrnn_loop_kernel" : (pl.kind == K_SPARSE ? "wrnn_sparse_kernel" : "wrnn_stream_kernel");
    out->units_per_wg = pl.kind == K_STREAM ? 0 : (pl.kind == K_SPARSE ? 64 : 16);
    out->clusters = pl.ncl; out->depth = pl.G; out->rounds = pl.rounds; out->slab_steps = pl.slab;
    return WRNN_OK;
}

extern "C" size_t wrnn_workspace_bytes(const wrnn_pack *p, const wrnn_geometry *g, const wrnn_options *opt)
{
    if (!p || check_geometry(g) != WRNN_OK) return 0;
    return wrnn_workspace_bytes_segments(p, g->B, g->T, g->n_frames, opt);
}

extern "C" int wrnn_generate_segments(const wrnn_pack *p, int32_t B, int32_t T, const int32_t *seg_pos,
                                      const int32_t *seg_lim, int32_t L, int32_t hop, int32_t n_frames,
                                      const float *mels_up, const float *aux, const float *noise, float *out,
                                      void *workspace, size_t workspace_bytes, const wrnn_options *opt, void *stream_)
{
    if (!p || !mels_up || !aux || !noise || !out || !workspace) { set_err("NULL argument"); return WRNN_ERR_ARG; }
    int rc = check_segments(B, T, seg_pos, seg_lim, L, hop, n_frames);
    if (rc != WRNN_OK) return rc;
    wrnn_options tmp;
    const wrnn_options *o = norm_options(opt, &tmp);
    Plan pl;
    {   // plan for the WHOLE call (a continuation must land on the same split and workspace layout)
        wrnn_options whole = *o;
        whole.t_begin = 0; whole.t_end = 0;
        if ((rc = make_plan(p, B, T, &whole, &pl)) != WRNN_OK) return rc;
        pl.t0 = o->t_begin; pl.t1 = o->t_end;
        if (pl.t0 == 0 && pl.t1 == 0) pl.t1 = T;
        if (pl.t0 < 0 || pl.t1 > T || pl.t0 >= pl.t1) { set_err("bad step range [%d, %d) of T=%d", pl.t0, pl.t1, T); return WRNN_ERR_ARG; }
        if (pl.kind != K_LOOP && pl.kind != K_DUO && pl.kind != K_SPARSE && pl.kind != K_CHAIN && (pl.t0 != 0 || pl.t1 != T)) { set_err("a partial step range needs a persistent loop kernel"); return WRNN_ERR_ARG; }
    }
    const WsLayout l = ws_layout(p, pl, B, T, n_frames);
    if (workspace_bytes < l.total) { set_err("workspace %zu < required %zu", workspace_bytes, l.total); return WRNN_ERR_WORKSPACE; }
    if (((uintptr_t)workspace & 255) != 0) { set_err("workspace must be 256-byte aligned"); return WRNN_ERR_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard dg(p->device);
    HIPCHK(dg.err);
    char *ws = (char *)workspace;
    wrnn_timer *timer = o->timer;
    if (timer && pl.t0 == 0) timer->used = 0;            // a continuing call (t_begin > 0) adds its launches to the same total

    // (a continuation keeps the status words: a give-up in an earlier slice must stay visible to wrnn_status(), and the
    // kernels leave at once when the abort flag is already up)
    if (pl.t0 == 0) HIPCHK(hipMemsetAsync(ws + l.status, 0, STATUS_WORDS * sizeof(unsigned), stream));
    // segment table -> device (pageable source: the runtime stages it before returning)
    HIPCHK(hipMemcpyAsync(ws + l.segs, seg_pos, (size_t)B * sizeof(int), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(ws + l.segs + (size_t)B * sizeof(int), seg_lim, (size_t)B * sizeof(int), hipMemcpyHostToDevice, stream));
    const int *d_pos = (const int *)(ws + l.segs), *d_lim = d_pos + B;
    if (o->mel_stage) {
        // the last up-sampling stage inside the loop: `mels_up` is that stage's input.  Everything is checked here, on the host: the
        // kernel reads rows j / s - 1 .. j / s + 1 without a bound.
        if (pl.kind != K_DUO && pl.kind != K_SPARSE && pl.kind != K_CHAIN) { set_err("wrnn_options.mel_stage: only wrnn_duo_kernel / wrnn_sparse_kernel / wrnn_chain_kernel form the last up-sampling stage (this call runs on another kernel)"); return WRNN_ERR_ARG; }
        if (o->mel_stage != 1 || o->mel_scale != LAST_SCALE || !o->mel_taps || !o->seg_moff || o->mel_rows < 3) {
            set_err("wrnn_options.mel_stage=%d: needs mel_scale == %d (got %d), mel_taps, seg_moff, mel_rows >= 3", o->mel_stage, LAST_SCALE, o->mel_scale);
            return WRNN_ERR_ARG;
        }
        for (int b = 0; b < B; ++b) {
            const long last = (seg_lim[b] < (long)seg_pos[b] + T ? seg_lim[b] : (long)seg_pos[b] + T) - 1;      // last position that is not zero padding
            if (last < seg_pos[b]) continue;
            const long j0 = (long)seg_pos[b] + o->seg_moff[b], j1 = last + o->seg_moff[b];
            if (o->seg_moff[b] < 0 || j0 / LAST_SCALE < 1 || j1 / LAST_SCALE + 1 >= o->mel_rows) {
                set_err("segment %d: positions %ld..%ld of the last up-sampling stage reach outside its %d input rows", b, j0, j1, o->mel_rows);
                return WRNN_ERR_ARG;
            }
        }
        // out(q) = sum_j w[j] rep(q + j - s), rep(u) = in[u / s]: with q = s a + ph the taps j < s - ph fall on row a - 1, the next s on
        // row a, the last ph + 1 on row a + 1
        float coef[3 * LAST_SCALE];
        for (int ph = 0; ph < LAST_SCALE; ++ph) {
            double c0 = 0, c1 = 0, c2 = 0;
            for (int j = 0; j <= 2 * LAST_SCALE; ++j) {
                const double w = o->mel_taps[j];
                if (j < LAST_SCALE - ph) c0 += w; else if (j < 2 * LAST_SCALE - ph) c1 += w; else c2 += w;
            }
            coef[3 * ph] = (float)c0; coef[3 * ph + 1] = (float)c1; coef[3 * ph + 2] = (float)c2;
        }
        HIPCHK(hipMemcpyAsync(ws + l.segs + (size_t)2 * B * sizeof(int), o->seg_moff, (size_t)B * sizeof(int), hipMemcpyHostToDevice, stream));
        HIPCHK(launch_put_floats((float *)(ws + l.melc), coef, 3 * LAST_SCALE, stream));       // (by value, as kernel arguments: `coef` is a stack array)
    }
    if (pl.kind == K_GENERIC) {
        GenArgs g;
        memset(&g, 0, sizeof g);
        g.I_T = p->g_I_T; g.I_b = p->I_b; g.w_ih1T = p->w_ih1T; g.w_hh1T = p->w_hh1T; g.b_ih1 = p->b_ih1; g.b_hh1 = p->b_hh1;
        g.w_ih2T = p->g_w_ih2T; g.w_hh2T = p->w_hh2T; g.b_ih2 = p->b_ih2; g.b_hh2 = p->b_hh2;
        g.fc1T = p->g_fc1T; g.fc1_b = p->fc1_b; g.fc2T = p->g_fc2T; g.fc2_b = p->fc2_b; g.fc3T = p->fc3T; g.fc3_b = p->fc3_b;
        g.mels_up = mels_up; g.aux = aux; g.noise = noise; g.force_x = o->force_x; g.out = out; g.dbg_logits = o->logits;
        g.seg_pos = d_pos; g.seg_lim = d_lim;
        g.H = p->gH; g.F = p->gF; g.M = p->gM; g.A = p->gA; g.C = p->C; g.B = B; g.T = T; g.hop = hop;
        if ((rc = timer_mark(timer, stream)) != WRNN_OK) return rc;
        HIPCHK(launch_generic(g, p->mode, stream));
        if ((rc = timer_mark(timer, stream)) != WRNN_OK) return rc;
        wrnn_run_info gi;
        memset(&gi, 0, sizeof gi);
        gi.kernel = "wrnn_generic_kernel"; gi.rounds = 1; gi.slab_steps = T; gi.launches = 1;
        if (o->info) *o->info = gi;
        return enqueue_progress(o, T, T, B, stream);
    }

    CondArgs c;
    memset(&c, 0, sizeof c);
    c.mels_up = mels_up; c.aux = aux; c.I_cT = p->I_cT; c.I_b = p->I_b; c.c2_wT = p->c2_wT; c.b_ih2 = p->b_ih2;
    c.c3_wT = p->c3_wT; c.fc1_b = p->fc1_b; c.c4_wT = p->c4_wT; c.fc2_b = p->fc2_b;
    c.c2f = (float *)(ws + l.c2f); c.c3f = (float *)(ws + l.c3f); c.c4f = (float *)(ws + l.c4f);
    c.seg_pos = d_pos; c.seg_lim = d_lim;
    c.B = B; c.T = T; c.hop = hop; c.NF = n_frames;
    const bool slabbed = pl.kind == K_DUO || pl.kind == K_SPARSE || pl.kind == K_CHAIN;
    if (slabbed) {
        // the aux tables are per segment and slab (filled in the slab loop): a slab may not span more than tab_fps - 2 whole hops
        const long eff = (long)(pl.tab_fps - 2) * hop + 1;
        if (pl.slab > eff) pl.slab = (int)eff;
    } else HIPCHK(launch_cond_frames(c, stream));

    LoopArgs a;
    memset(&a, 0, sizeof a);
    a.I_w0 = p->I_w0; a.w_ih1 = p->w_ih1; a.w_hh1 = p->w_hh1; a.b_ih1 = p->b_ih1; a.b_hh1 = p->b_hh1;
    a.w_ih2 = p->w_ih2; a.w_hh2 = p->w_hh2; a.b_hh2 = p->b_hh2; a.fc1_w = p->fc1_w; a.fc2_w = p->fc2_w;
    a.fc3_w = p->fc3_w; a.fc3_b = p->fc3_b; a.fc3f = p->fc3f; a.u1 = p->u1;
    a.w_ih1T = p->w_ih1T; a.w_hh1T = p->w_hh1T; a.w_ih2T = p->w_ih2T; a.w_hh2T = p->w_hh2T;
    a.fc1T = p->fc1T; a.fc2T = p->fc2T; a.fc3T = p->fc3T;
    a.c2f = c.c2f; a.c3f = c.c3f; a.c4f = c.c4f;
    a.noise = noise; a.force_x = o->force_x; a.out = out; a.dbg_logits = o->logits;
    a.status = (unsigned *)(ws + l.status);
    a.prof = (u64 *)o->phase_clocks;
    a.tuning = o->tuning;
    a.seg_pos = d_pos; a.seg_lim = d_lim;
    a.Btot = B; a.T = T; a.hop = hop; a.NF = n_frames; a.C = p->C;
    a.NG = (B + SEG - 1) / SEG;
    a.Nall = B;
    a.xcc_tab = (unsigned *)(ws + l.xcc);
    a.mels_up = mels_up; a.aux_fr = aux; a.I_cT = p->I_cT; a.I_b = p->I_b;
    a.mel_stage = o->mel_stage; a.seg_moff = d_pos + 2 * B; a.mel_coef = (const float *)(ws + l.melc);
    hop_magic(hop, &a.hop_magic, &a.hop_shift);

    wrnn_run_info info;
    memset(&info, 0, sizeof info);
    info.clusters = pl.ncl; info.depth = pl.G; info.rounds = pl.rounds; info.slab_steps = pl.slab;

    if (pl.kind == K_LOOP || slabbed) {
        // ---- persistent loop kernels: for every slab of steps { derived noise; for every round { conditioning slab; loop } } ----
        const bool duo = slabbed, sparse = pl.kind == K_SPARSE, chain = pl.kind == K_CHAIN;      // (duo: the conditioning is formed in the loop)
        const bool octo = pl.kind == K_DUO && pl.octo;
        info.kernel = chain ? "wrnn_chain_kernel" : sparse ? "wrnn_sparse_kernel" : (duo ? (octo ? "wrnn_octo_kernel" : "wrnn_duo_kernel") : "wrnn_loop_kernel"); info.units_per_wg = sparse ? 64 : 16;
        a.sp_vals = p->sp_vals; a.sp_cols = p->sp_cols;
        a.sp_fc_vals = (o->tuning & 2048) ? nullptr : p->sp_fc_vals; a.sp_fc_cols = p->sp_fc_cols;      // (tuning bit 11: dense fc stages on a pack whose Linear layers are sparse too -- A/B)
        const bool mol = p->mode == WRNN_MODE_MOL;
        a.xbuf = (float *)(ws + l.xbuf);
        a.cIf = (const float *)(ws + l.cIf);
        a.G = pl.G;
        c.cI = (float *)(ws + l.cIf);
        for (int s0 = pl.t0; s0 < pl.t1; s0 += pl.slab) {
            const int s1 = s0 + pl.slab < pl.t1 ? s0 + pl.slab : pl.t1;
            if (mol) {   // noise rows are relative to t_begin (wrnn_options.t_begin)
                HIPCHK(launch_noise_mol(noise + (size_t)(s0 - pl.t0) * 11 * B, (float *)(ws + l.npre), (long)(s1 - s0) * 11 * B, B, p->n_cus, stream));
                a.noise_pre = (const float *)(ws + l.npre);
                a.noise_t0 = s0;
            } else {
                a.noise_t0 = pl.t0;
            }
            if (duo) {   // this slab's aux tables of every segment of the call
                c.t0 = s0; c.B = B; c.FPS = pl.tab_fps;
                HIPCHK(launch_cond_frames_slab(c, stream));
                a.tab_fps = pl.tab_fps; a.tab_t0 = s0;
            }
            for (int r = 0; r < pl.rounds; ++r) {
                const int rb0 = (int)(((long)r * B) / pl.rounds), rb1 = (int)(((long)(r + 1) * B) / pl.rounds);
                const int nr = rb1 - rb0;
                if (nr < 1) continue;
                const int ngr = (nr + SEG - 1) / SEG;
                c.t0 = s0; c.t1 = s1; c.rb0 = rb0; c.B = nr; c.NG = ngr;
                if (!duo) HIPCHK(launch_cond_frag(c, p->n_cus, stream));
                // every word of the exchange ring = the sentinel.  The duo kernel leaves its ring consistent at the end of a launch
                // (every step re-arms the entries it will write two or three steps later), so it needs the fill only where a round starts: the first
                // launch of a call that starts at step 0, or any launch when several rounds share the buffer
                if (!duo) HIPCHK(hipMemsetAsync(ws + l.xbuf, 0xFF, XBUF_FLOATS * sizeof(float), stream));
                else if (s0 == 0 || pl.rounds > 1 || (o->tuning & 4)) HIPCHK(hipMemsetAsync(ws + l.xbuf, 0xFF, chain ? chain_xbuf_bytes(pl.G) : sparse ? sparse_xbuf_bytes() : duo_xbuf_bytes(pl.G), stream));
                if (duo) HIPCHK(hipMemsetAsync(ws + l.xcc, 0, XCC_WORDS * sizeof(unsigned), stream));      // placement handshake of this launch
                a.state = (float *)(ws + l.state) + (size_t)r * (sparse ? sparse_state_floats() : chain ? chain_state_floats(pl.G) : loop_state_floats(pl.G));
                a.t0 = s0; a.t1 = s1; a.cI_t0 = s0; a.rb0 = rb0; a.Btot = nr; a.NG = ngr; a.resume = s0 > 0 ? 1 : 0;
                a.kind_tag = chain ? 4 : sparse ? 3 : (duo ? (octo ? 5 : 2) : 1);
                if ((rc = timer_mark(timer, stream)) != WRNN_OK) return rc;
                hipError_t e = chain ? launch_chain(a, p->mode, stream) : sparse ? launch_sparse(a, p->sp_nbp, stream) : (duo ? (octo ? launch_octo(a, pl.ncl, p->mode, stream) : launch_duo(a, pl.ncl, p->mode, stream)) : launch_loop(a, pl.ncl, p->mode, stream));
                // (two workgroups per CU not co-resident right now: WRNN_ERR_RESIDENCY -- the caller re-plans with WRNN_ALGO_LOOP, whose
                // workspace layout is another one: wavernn_amd/engine.py does)
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    set_err("%s cooperative launch failed: %s", info.kernel, hipGetErrorString(e));
                    return e == hipErrorCooperativeLaunchTooLarge ? WRNN_ERR_RESIDENCY : WRNN_ERR_HIP;
                }
                if ((rc = timer_mark(timer, stream)) != WRNN_OK) return rc;
                info.launches += 1;
            }
            if ((rc = enqueue_progress(o, s1, T, B, stream)) != WRNN_OK) return rc;
        }
    } else {
        // ---- stream kernel: whole-T conditioning in [t][segment][H] order, one launch --------------------
        c.cI = (float *)(ws + l.cI);
        a.cI = c.cI;
        HIPCHK(launch_cond(c, p->n_cus, o->cond_valu != 0, stream));
        info.kernel = "wrnn_stream_kernel";
        a.b0 = 0;
        a.nb = B;
        if ((rc = timer_mark(timer, stream)) != WRNN_OK) return rc;
        HIPCHK(launch_stream(a, p->mode, stream));
        if ((rc = timer_mark(timer, stream)) != WRNN_OK) return rc;
        info.launches = 1;
        if ((rc = enqueue_progress(o, T, T, B, stream)) != WRNN_OK) return rc;
    }
    if (o->info) *o->info = info;
    return WRNN_OK;
}

extern "C" int wrnn_generate(const wrnn_pack *p, const wrnn_geometry *g, const float *mels_up, const float *aux,
                             const float *noise, float *out, void *workspace, size_t workspace_bytes, const wrnn_options *opt,
                             void *stream_)
{
    int rc = check_geometry(g);
    if (rc != WRNN_OK) return rc;
    std::vector<int32_t> pos(g->B), lim(g->B, g->L);
    for (int b = 0; b < g->B; ++b) pos[b] = b * g->stride;
    return wrnn_generate_segments(p, g->B, g->T, pos.data(), lim.data(), g->L, g->hop, g->n_frames, mels_up, aux,
                                  noise, out, workspace, workspace_bytes, opt, stream_);
}

extern "C" int wrnn_status(void *workspace, void *stream)
{
    if (!workspace) { set_err("NULL workspace"); return WRNN_ERR_ARG; }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    unsigned st[STATUS_WORDS];
    HIPCHK(hipMemcpy(st, workspace, sizeof st, hipMemcpyDeviceToHost));
    if (st[0] != 0 || st[1] != 0) {
        set_err("loop kernel gave up: code 0x%x (phase %u) workgroup %u step %u thread %u", st[1], st[1] & 0xff, st[2], st[3], st[4]);
        return WRNN_ERR_KERNEL;
    }
    return WRNN_OK;
}

// Test hook: one exchanged layer (0 h1, 1 h2, 2 y1, 3 y2, 4 RAW logits) of (cluster, slot) at ring position `ring` of the loop
// kernel's exchange buffer, un-permuted from fragment order to host [16 segments][512].  Synchronises.
extern "C" int wrnn_debug_read_exchange(const wrnn_pack *p, void *workspace, int32_t n_segments, int32_t T, int32_t n_frames,
                                        const wrnn_options *opt, int cluster, int slot, int layer, int ring, float *host_out)
{
    if (!p || !workspace || !host_out) { set_err("NULL argument"); return WRNN_ERR_ARG; }
    wrnn_options tmp;
    wrnn_options whole = *norm_options(opt, &tmp);
    whole.t_begin = 0; whole.t_end = 0;
    Plan pl;
    int rc = make_plan(p, n_segments, T, &whole, &pl);
    if (rc != WRNN_OK) return rc;
    if (pl.kind != K_LOOP || cluster < 0 || cluster >= MAXCL || slot < 0 || slot >= LMAXG || layer < 0 || layer >= NXLAYER || ring < 0 || ring >= XRING) {
        set_err("no such exchange layer");
        return WRNN_ERR_ARG;
    }
    const WsLayout l = ws_layout(p, pl, n_segments, T, n_frames);
    HIPCHK(hipDeviceSynchronize());
    std::vector<float> frag((size_t)SEG * H);
    const size_t off = ((((size_t)cluster * LMAXG + slot) * NXLAYER + layer) * XRING + ring) * SEG * H;
    HIPCHK(hipMemcpy(frag.data(), (char *)workspace + l.xbuf + off * sizeof(float), frag.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int j = 0; j < SEG; ++j)
        for (int k = 0; k < H; ++k) {
            const int w = k >> 7, r = (k >> 4) & 7, kq = (k >> 2) & 3, e = k & 3;
            host_out[(size_t)j * H + k] = frag[(size_t)(((w * 8 + r) * 64 + kq * 16 + j) * 4 + e)];
        }
    return WRNN_OK;
}

static float g_selftest_metric = -1.f;
extern "C" float wrnn_selftest_metric(void) { return g_selftest_metric; }

extern "C" int wrnn_selftest(int device, int which)
{
    const int cus = wrnn_device_cus(device);
    if (cus < 0) return cus;
    DeviceGuard dg(device);
    HIPCHK(dg.err);
    char msg[400] = "";
    int rc;
    if (which == 1) rc = selftest_mfma(msg, sizeof msg);
    else if (which == 2) rc = selftest_allgather(cus, msg, sizeof msg, &g_selftest_metric);
    else if (which == 3) rc = selftest_tanh(msg, sizeof msg);
    else if (which == 4) rc = selftest_xor(msg, sizeof msg);
    else { set_err("unknown selftest %d", which); return WRNN_ERR_ARG; }
    if (rc != 0) { set_err("selftest %d failed: %s", which, msg); return WRNN_ERR_KERNEL; }
    set_err("selftest %d ok: %s", which, msg);
    return WRNN_OK;
}
