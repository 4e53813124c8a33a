// gnnb_runtime.hip -- host side of libgnnb_hip.so: model / workspace handles, the batched
// forward that sequences the kernels of k_*.hip on one HIP stream, and the C ABI
// declared in include/gnnb_hip.h.
//
// Sequencing follows the reference's generated top (gnnbuilder/templates/model.cpp.jinja):
//   load_parameters once (:724-730)            -> gnnb_model_create (device-resident weights)
//   compute_degree/neighbor_tables (:737-758)  -> k_graph_prep, once per batch, shared by all layers
//   compute_gnn_head (:151-359)                -> per layer: k_aggregate + k_linear (+skip +act fused)
//   compute_global_graph_pooling (:413-449)    -> k_global_pool
//   compute_mlp_head (:454-530)                -> k_linear chain
// There is no CPU fallback: every entry point fails with GNNB_ERR_NO_DEVICE / GNNB_ERR_HIP when
// the GPU path cannot run.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gnnb_internal.h"

namespace gnnb {

static thread_local std::string g_last_error;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define GNNB_HIP_TRY(expr)                                                                        \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail(GNNB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                        __FILE__, __LINE__);                                                      \
    } while (0)

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

Options &options()
{
    static Options o = {env_int("GNNB_TILE_ROWS", 8), env_int("GNNB_AGG_LDS_KB", 0),
                                                env_int("GNNB_AGG_RING_WAVES", 0),    env_int("GNNB_AGG_RING_SLOTS", 2),
                        env_int("GNNB_AGG_RING_WG_PER_CU", 1), env_int("GNNB_AGG_NT_STORE", 1), env_int("GNNB_AGG_BALANCE", 0),
                        env_int("GNNB_GEMM_VARIANT", 0),
                        env_int("GNNB_GEMM_MAX_WG_PER_CU", 2), env_int("GNNB_GEMM_DMA", 1),
                        env_int("GNNB_GEMM_WLDS", 1),             env_int("GNNB_GEMM_WLDS_SLOTS", 2),
                        env_int("GNNB_FUSE_NARROW", 1), env_int("GNNB_FIRST_RING", 1),    env_int("GNNB_FUSE_ZF", 1),   env_int("GNNB_LARGE_FORK", 2), env_int("GNNB_ZF_SHAPE", 2),
                        env_int("GNNB_FUSE_GCN2", 1),         env_int("GNNB_FUSE_HEAD", 1), env_int("GNNB_FUSE_POOL", 1),
                        env_int("GNNB_HEAD_SMALL", 1),        env_int("GNNB_HEAD_SPLIT", 0),
                        env_int("GNNB_MATH", 0),              env_int("GNNB_GEMM_TAIL_SPLIT", 2), env_int("GNNB_PNA_FOLD_LIN", 1), env_int("GNNB_PNA_CLASSES", 1), env_int("GNNB_FOLD_SKIP", 1), env_int("GNNB_SAGE_FIRST_MEAN", 1), env_int("GNNB_PNA_FIRST", 1), env_int("GNNB_PNA_PAGG", 1), env_int("GNNB_STAGE_CUT", 0), env_int("GNNB_ZF_HEAD", 0),
                        env_int("GNNB_AGG_FORM", 0), env_int("GNNB_AGG_RG_R", 0), env_int("GNNB_AGG_RG_WGS", 0), env_int("GNNB_AGG_RG_FLAGS", 1), env_int("GNNB_PREP_GROUP", 4), env_int("GNNB_HEAD_PAIRS", 1), env_int("GNNB_GUEST_PREP", 1)};
    return o;
}

static thread_local int tl_math = -1; // >= 0: the calling thread is inside an entry point of a model with its own math mode
static thread_local FlagWord tl_flag = {nullptr, nullptr};
int launch_math() { return tl_math >= 0 ? tl_math : (int)options().math; }
static thread_local GuestPrep *tl_guest = nullptr;
GuestPrep *&guest_prep_slot() { return tl_guest; }
FlagWord launch_flag_word() { return tl_flag; }
MathScope::MathScope(int model_math, int32_t *err, int32_t *err_host) : prev(tl_math), prev_flag(tl_flag)
{
    if (model_math >= 0)
        tl_math = model_math;
    if (err)
        tl_flag = FlagWord{err, err_host};
}
MathScope::~MathScope()
{
    tl_math = prev;
    tl_flag = prev_flag;
}

// ---------------------------------------------------------------------------------------
struct LayerDims {
    int fin, fout;
};

static int conv_slots(int conv)
{
    switch (conv) {
    case GNNB_CONV_GCN: return 2;
    case GNNB_CONV_GIN: return 4;
    case GNNB_CONV_SAGE: return 3;
    case GNNB_CONV_PNA: return 6;
    default: return -1;
    }
}

// gnnbuilder/models.py:519-549
static LayerDims layer_dims(const gnnb_model_desc &d, int l)
{
    if (d.num_layers == 1)
        return {d.in_dim, d.out_dim};
    if (l == 0)
        return {d.in_dim, d.hidden_dim};
    if (l == d.num_layers - 1)
        return {d.hidden_dim, d.out_dim};
    return {d.hidden_dim, d.hidden_dim};
}

static int gnn_out_width(const gnnb_model_desc &d) { return d.num_layers == 0 ? d.in_dim : d.out_dim; }

// gnnbuilder/models.py:398-415
static void mlp_dims(const gnnb_model_desc &d, int i, int *din, int *dout)
{
    const int pooled = d.num_pools * gnn_out_width(d);
    *din = (i == 0) ? pooled : d.mlp_hidden;
    *dout = (i == d.mlp_num_linear - 1) ? d.mlp_out : d.mlp_hidden;
}

static int validate_desc(const gnnb_model_desc *d)
{
    if (!d)
        return fail(GNNB_ERR_INVALID, "null model description");
    if (conv_slots(d->conv_type) < 0)
        return fail(GNNB_ERR_INVALID, "unsupported conv_type %d", d->conv_type);
    if (d->num_layers < 0 || d->num_layers > GNNB_MAX_LAYERS)
        return fail(GNNB_ERR_INVALID, "num_layers %d out of range", d->num_layers);
    if (d->in_dim < 1 || d->out_dim < 1 || (d->num_layers > 1 && d->hidden_dim < 1))
        return fail(GNNB_ERR_INVALID, "feature dims must be positive");
    if (d->num_layers == 0 && d->in_dim != d->out_dim) // models.py:512-518
        return fail(GNNB_ERR_INVALID, "gnn_num_layers=0 needs gnn_output_dim == graph_input_feature_dim");
    if (d->activation < 0 || d->activation > GNNB_ACT_TANH || d->mlp_activation < 0 ||
        d->mlp_activation > GNNB_ACT_TANH)
        return fail(GNNB_ERR_INVALID, "unsupported activation"); // models.py:362
    if (d->num_pools < 1 || d->num_pools > 3)
        return fail(GNNB_ERR_INVALID, "num_pools must be 1..3"); // models.py:332-333
    for (int i = 0; i < d->num_pools; i++)
        if (d->pools[i] < 0 || d->pools[i] > GNNB_POOL_MAX)
            return fail(GNNB_ERR_INVALID, "unsupported pooling %d", d->pools[i]);
    if (d->mlp_num_linear < 1 || d->mlp_num_linear > GNNB_MAX_LAYERS || d->mlp_out < 1 ||
        (d->mlp_num_linear > 1 && d->mlp_hidden < 1))
        return fail(GNNB_ERR_INVALID, "bad MLP head shape");
    if (d->conv_type == GNNB_CONV_PNA && !(d->pna_delta > 0.0f))
        return fail(GNNB_ERR_INVALID, "pna_delta must be > 0");
    if (d->output_activation < GNNB_OUT_NONE || d->output_activation > GNNB_OUT_LOG_SOFTMAX)
        return fail(GNNB_ERR_INVALID, "unsupported output_activation %d", d->output_activation);
    if (d->fpx_w != 0 && (d->fpx_w < 2 || d->fpx_w > 32 || d->fpx_i < 1 || d->fpx_i > 33 || d->fpx_i > d->fpx_w ||
                          d->fpx_w - d->fpx_i > 24))
        return fail(GNNB_ERR_INVALID, "fixed-point emulation takes 2 <= W <= 32, 1 <= I <= W, W - I <= 24 (fp32 carries the "
                                      "grid values exactly only up to 24 fractional bits)");
    if (d->math < -1 || d->math > 3)
        return fail(GNNB_ERR_INVALID, "math must be -1 (follow the process-wide option) or 0 .. 3 (fp32, bf16x6, bf16x3, f16x3)");
    return GNNB_OK;
}

} // namespace gnnb

using namespace gnnb;

struct gnnb_model {
    gnnb_model_desc desc;
    float *blob = nullptr; // all weights, device
    size_t blob_floats = 0;
    // per conv layer device pointers (canonical slots; SAGE slot 0 is the fused [Wl|Wr])
    std::vector<std::vector<const float *>> conv;
    std::vector<const float *> head_w, head_b;
    const float *zf_w1f = nullptr; // 2-layer GCN: layer 1's weight once more, in MFMA-fragment order (see k_gcn2_zf)
    // GIN stacks (k_gcn2_fused<GIN>): every wide matrix once more in EXECUTION order, each hidden x hidden at one stride --
    // Wb0 | Wa1 Wb1 | ... | Wa(L-1) Wb(L-1) -- and the biases likewise.  A last layer narrower than hidden (the reference's
    // benchmark model: 128 -> 64, models.py:530-545) is zero-padded to hidden x hidden: its extra output columns are
    // act(0 + 0) and never leave the kernel.  nullptr when the model is no GIN stack the kernel takes.
    const float *gin_w = nullptr, *gin_b = nullptr;
    HeadArgs *head_dev = nullptr; // the MLP head's {weights, biases, widths} once more in device memory: k_gcn2_zf reads it at the
                                  // end of a workgroup's life (by value the 42 dwords stayed in scalar registers through its stage loop)
    int device = 0;
};

struct gnnb_workspace {
    gnnb_model_desc desc;
    int max_graphs = 0, max_nodes = 0, max_edges = 0;
    char *blob = nullptr;
    size_t bytes = 0;
    BatchTables t{};
    float *act[2] = {nullptr, nullptr}; // ping-pong node embeddings [max_nodes, maxw]
    float *agg = nullptr;               // aggregate output [max_nodes, aggw]
    float *tmp0 = nullptr, *tmp1 = nullptr; // GIN hidden / PNA p,q
    float *pooled = nullptr;            // [max_graphs, np*d]
    float *mlp[2] = {nullptr, nullptr}; // [max_graphs, max(mlp_hidden, mlp_out)]
    bool prepared = false;
    float2 *pool_part = nullptr; // pieces of graphs that cross the 32-row blocks of the pooling GEMM epilogue (PoolEpilogue::part)
    bool gcoef_ready = false; // t.gcoef holds the prepared batch's GCN coefficients (ensure_gcoef)
    int max_graph_nodes = 0; // caller's promise (0 = none)
    int max_degree = 0;      // caller's promise on the in-degree (0 = none): gnnb_workspace_set_max_degree
    // PNA degree classes of the prepared batch (launch_degree_classes): valid when deg_ready; deg_delta = the delta it was prepared with
    int32_t *deg_work = nullptr, *deg_perm = nullptr, *deg_tile_cls = nullptr;
    int deg_max_tiles = 0;
    bool deg_ready = false;
    float deg_delta = 0.0f;
    int32_t *plan_scratch = nullptr; // k_stage_cut's binary-lifting tables (GCN / GIN workspaces: stage_cut_levels x (max tiles + 1) ints)
    float prep_delta = 0.0f; // the delta the prepared batch's amp / att tables were computed with (PNA workspaces; 0: none)
    int last_path = GNNB_PATH_NONE; // which kernels the last forward on this workspace ran (gnnb_workspace_last_path)
    // "large segment" of the NEXT batches (gnnb_workspace_set_large_segment): graphs [large_g, B) -- nodes from large_n,
    // edges from large_e -- are exempt from the max_graph_nodes promise and run layer by layer; -1 = no such segment
    int large_g = -1, large_n = -1, large_e = -1;
    // fork / join for the large segment: its small kernels run on `side` beside the stack kernel on the caller's stream
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int device = 0;
    int32_t *err_host = nullptr; // host-mapped word the prep kernel drops "flagged" into (lazy detection, see gnnb_graph_prep)
    StreamK sk{};            // this workspace's own stream-K scratch (k_linear_dma's large-K tail; part == nullptr: the model has no such GEMM)
    char *stage = nullptr;   // device staging of the host-buffer entry (x | coo | node_ptr | edge_ptr | out), sized for
    size_t stage_bytes = 0;  // the workspace's capacities; allocated by the first gnnb_forward_batched_host call
};

static HeadArgs model_head_args(const gnnb_model *model)
{
    const gnnb_model_desc &d = model->desc;
    HeadArgs head;
    memset(&head, 0, sizeof(head));
    head.nlin = d.mlp_num_linear;
    for (int i = 0; i < head.nlin && i < 8; i++) {
        int din, dout;
        mlp_dims(d, i, &din, &dout);
        head.w[i] = model->head_w[i];
        head.b[i] = model->head_b[i];
        head.dims[i] = din;
        head.dims[i + 1] = dout;
    }
    return head;
}

// ---------------------------------------------------------------------------------------
extern "C" {

int gnnb_version(void) { return GNNB_VERSION; }

const char *gnnb_last_error(void) { return g_last_error.c_str(); }

int gnnb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int gnnb_stream_sync(void *stream)
{
    GNNB_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return GNNB_OK;
}

int gnnb_set_option(const char *name, int value)
{
    Options &o = options();
    if (!name)
        return fail(GNNB_ERR_INVALID, "null option name");
    if (!strcmp(name, "tile_rows") && value >= 4)
        o.tile_rows = value;
    else if (!strcmp(name, "agg_lds_kb") && value >= 0 && value <= 160)
        o.agg_lds_kb = value; // 0 = default of the selected form
    else if (!strcmp(name, "agg_ring_waves") && (value == 0 || value == 1 || value == 2 || value == 4 || value == 8 || value == 16))
        o.agg_ring_waves = value;
    else if (!strcmp(name, "agg_ring_slots") && value >= 1 && value <= 4)
        o.agg_ring_slots = value;
    else if (!strcmp(name, "agg_ring_wg_per_cu") && value >= 1 && value <= 4)
        o.agg_ring_wg_per_cu = value;
    else if (!strcmp(name, "agg_nt_store") && value >= 0 && value <= 1)
        o.agg_nt_store = value;
    else if (!strcmp(name, "agg_balance") && value >= 0 && value <= 1)
        o.agg_balance = value;
    else if (!strcmp(name, "sage_first_mean") && value >= 0 && value <= 1)
        o.sage_first_mean = value;
    else if (!strcmp(name, "pna_first") && value >= 0 && value <= 1)
        o.pna_first = value;
    else if (!strcmp(name, "pna_pagg") && value >= 0 && value <= 1)
        o.pna_pagg = value;
    else if (!strcmp(name, "stage_cut") && value >= 0 && value <= 1)
        o.stage_cut = value;
    else if (!strcmp(name, "zf_head") && value >= 0 && value <= 1)
        o.zf_head = value;
    else if (!strcmp(name, "prep_group") && (value == 1 || value == 4))
        o.prep_group = value;
    else if (!strcmp(name, "head_pairs") && value >= 0 && value <= 1)
        o.head_pairs = value;
    else if (!strcmp(name, "guest_prep") && value >= 0 && value <= 1)
        o.guest_prep = value;
    else if (!strcmp(name, "agg_form") && value >= 0 && value <= 2)
        o.agg_form = value;
    else if (!strcmp(name, "agg_rg_r") && value >= 0 && value <= 4)
        o.agg_rg_r = value;
    else if (!strcmp(name, "agg_rg_wgs") && value >= 0 && value <= 64)
        o.agg_rg_wgs = value;
    else if (!strcmp(name, "agg_rg_flags") && value >= 0 && value <= 15)
        o.agg_rg_flags = value;
    else if (!strcmp(name, "fuse_narrow") && value >= 0 && value <= 1)
        o.fuse_narrow = value;
    else if (!strcmp(name, "first_ring") && value >= 0 && value <= 1)
        o.first_ring = value;
    else if (!strcmp(name, "fuse_gcn2") && value >= 0 && value <= 1)
        o.fuse_gcn2 = value;
    else if (!strcmp(name, "fuse_zf") && value >= 0 && value <= 1)
        o.fuse_zf = value;
    else if (!strcmp(name, "zf_shape") && value >= 0 && value <= 2)
        o.zf_shape = value;
    else if (!strcmp(name, "large_fork") && value >= 0 && value <= 2)
        o.large_fork = value;
    else if (!strcmp(name, "fuse_head") && value >= 0 && value <= 1)
        o.fuse_head = value;
    else if (!strcmp(name, "fuse_pool") && value >= 0 && value <= 1)
        o.fuse_pool = value;
    else if (!strcmp(name, "head_split") && value >= 0 && value <= 1)
        o.head_split = value;
    else if (!strcmp(name, "head_small") && value >= 0 && value <= 1)
        o.head_small = value;
    else if (!strcmp(name, "math") && value >= 0 && value <= 3)
        o.math = value;
    else if (!strcmp(name, "gemm_variant") && value >= 0 && value <= 1)
        o.gemm_variant = value;
    else if (!strcmp(name, "gemm_dma") && value >= 0 && value <= 1)
        o.gemm_dma = value;
    else if (!strcmp(name, "gemm_tail_split") && value >= 0 && value <= 2)
        o.gemm_tail_split = value;
    else if (!strcmp(name, "pna_fold_lin") && value >= 0 && value <= 1)
        o.pna_fold_lin = value;
    else if (!strcmp(name, "pna_classes") && value >= 0 && value <= 1)
        o.pna_classes = value;
    else if (!strcmp(name, "fold_skip") && value >= 0 && value <= 1)
        o.fold_skip = value;
    else if (!strcmp(name, "gemm_wlds") && value >= 0 && value <= 1)
        o.gemm_wlds = value;
    else if (!strcmp(name, "gemm_wlds_slots") && value >= 1 && value <= 4)
        o.gemm_wlds_slots = value;
    else if (!strcmp(name, "gemm_max_wg_per_cu") && value >= 1 && value <= 8)
        o.gemm_max_wg_per_cu = value;
    else
        return fail(GNNB_ERR_INVALID, "unknown option or bad value: %s=%d", name, value);
    return GNNB_OK;
}

int gnnb_model_num_params(const gnnb_model_desc *desc)
{
    int rc = validate_desc(desc);
    if (rc != GNNB_OK)
        return rc;
    return conv_slots(desc->conv_type) * desc->num_layers + 2 * desc->mlp_num_linear;
}

int gnnb_model_create(const gnnb_model_desc *desc, const float *const *host_params, int num_params,
                      gnnb_model **out_model)
{
    if (!out_model)
        return fail(GNNB_ERR_INVALID, "null out_model");
    *out_model = nullptr;
    int expect = gnnb_model_num_params(desc);
    if (expect < 0)
        return expect;
    if (num_params != expect || !host_params)
        return fail(GNNB_ERR_INVALID, "expected %d parameter tensors, got %d", expect, num_params);
    for (int i = 0; i < num_params; i++)
        if (!host_params[i])
            return fail(GNNB_ERR_INVALID, "parameter %d is NULL", i);
    if (gnnb_device_count() <= 0)
        return fail(GNNB_ERR_NO_DEVICE, "no HIP device visible: the MI355X path cannot run");

    const gnnb_model_desc &d = *desc;
    // host staging image: every tensor padded to a 16-byte boundary
    std::vector<float> img;
    const bool fpx = d.fpx_w > 0;
    const float q_inv = fpx ? ldexpf(1.0f, d.fpx_w - d.fpx_i) : 1.0f, q_step = fpx ? ldexpf(1.0f, -(d.fpx_w - d.fpx_i)) : 1.0f;
    const float q_span = fpx ? ldexpf(1.0f, d.fpx_i) : 1.0f, q_half = fpx ? ldexpf(1.0f, d.fpx_i - 1) : 1.0f;
    auto push = [&](const float *src, size_t n) -> size_t {
        size_t off = img.size();
        img.insert(img.end(), src, src + n);
        if (fpx) // W_TYPE = ap_fixed<W, I>: the weights live on the grid (model.h.jinja:41-45)
            for (size_t i = off; i < off + n; i++) {
                float v = floorf(img[i] * q_inv) * q_step;
                img[i] = v - q_span * floorf((v + q_half) / q_span);
            }
        while (img.size() % 4)
            img.push_back(0.0f);
        return off;
    };
    std::vector<std::vector<size_t>> conv_off(d.num_layers);
    std::vector<size_t> hw, hb;
    const int slots = conv_slots(d.conv_type);
    int pi = 0;
    for (int l = 0; l < d.num_layers; l++) {
        const LayerDims ld = layer_dims(d, l);
        const size_t fi = ld.fin, fo = ld.fout;
        const float *const *p = host_params + pi;
        // the layer's skip connection (middle layers: y = conv(x) + x, models.py:562-564) where x itself is an operand of
        // the layer's GEMM (GraphSAGE's root term, PNA's x segment): folded into that operand's weights as + I, so that the
        // [N, out] skip operand is not read again in the epilogue (45 of 293 us of a 128-wide PNA layer's GEMM went there:
        // 32-byte pieces of 128-byte lines)
        const bool skip_fold = d.skip && l != 0 && l != d.num_layers - 1 && fi == fo && !fpx;
        switch (d.conv_type) {
        case GNNB_CONV_GCN:
            conv_off[l] = {push(p[0], fo * fi), push(p[1], fo)};
            break;
        case GNNB_CONV_GIN: // hidden = out_channels (models.py:90)
            conv_off[l] = {push(p[0], fo * fi), push(p[1], fo), push(p[2], fo * fo), push(p[3], fo)};
            break;
        case GNNB_CONV_SAGE: {
            // fuse lin_l and lin_r into one [out, 2*in] matrix: [Wl | Wr]
            std::vector<float> cat(fo * 2 * fi);
            for (size_t o = 0; o < fo; o++) {
                memcpy(&cat[o * 2 * fi], p[0] + o * fi, fi * sizeof(float));
                memcpy(&cat[o * 2 * fi + fi], p[2] + o * fi, fi * sizeof(float));
            }
            conv_off[l] = {push(cat.data(), cat.size()), push(p[1], fo)};
            if (skip_fold) { // slot 2: [Wl | Wr + I]
                for (size_t o = 0; o < fo; o++)
                    cat[o * 2 * fi + fi + o] += 1.0f;
                conv_off[l].push_back(push(cat.data(), cat.size()));
            }
            break;
        }
        case GNNB_CONV_PNA:
            conv_off[l] = {push(p[0], fi * 2 * fi), push(p[1], fi), push(p[2], fo * 13 * fi),
                           push(p[3], fo),          push(p[4], fo * fo), push(p[5], fo)};
            // slots 6, 7: `lin` folded into the post-NN.  PNAConv applies them back to back with nothing in between
            // (out = W_lin (W_post [x | S] + b_post) + b_lin, gnn_builder_lib.h:2081-2157), so
            // W' = W_lin W_post [out, 13 F] and b' = W_lin b_post + b_lin (formed in double, rounded once) give the layer
            // in ONE 13F-wide GEMM whose epilogue carries the skip operand and the activation: the out x out GEMM and the
            // [N, out] hand-over between the two are gone (3 x ~55 us of a BASELINE config 4 step).  Not under the
            // fixed-point emulation: the folded matrix is not on the weight grid.
            if (!fpx) {
                const size_t K13 = 13 * fi;
                std::vector<double> acc(K13);
                std::vector<float> wm(fo * K13), bm(fo);
                for (size_t o = 0; o < fo; o++) {
                    std::fill(acc.begin(), acc.end(), 0.0);
                    double ab = (double)p[5][o];
                    for (size_t h = 0; h < fo; h++) {
                        const double wl = (double)p[4][o * fo + h];
                        const float *wp = p[2] + h * K13;
                        for (size_t k = 0; k < K13; k++)
                            acc[k] += wl * (double)wp[k];
                        ab += wl * (double)p[3][h];
                    }
                    if (skip_fold)
                        acc[o] += 1.0; // (+ I on the x segment: the skip connection)
                    for (size_t k = 0; k < K13; k++)
                        wm[o * K13 + k] = (float)acc[k];
                    bm[o] = (float)ab;
                }
                conv_off[l].push_back(push(wm.data(), wm.size()));
                conv_off[l].push_back(push(bm.data(), bm.size()));
                // slots 8, 9: the degree-class form (gnnb_workspace_set_max_degree): for every in-degree c = 0 .. 15 the matrix
                //   ( W'_x + S_c Wq | W'_1 + amp W'_2 + att W'_3 ),  [out, 5 F],   and the bias  b' + S_c bq,
                // of the folded W' above; amp / att as graph prep computes them (k_prep.hip: logf(d + 1) / delta and its
                // reciprocal, d = max(c, 1)).  S_c = the max + min + mean column blocks of the class's A matrix: the
                // destination's own pre-NN term q_i = Wq x_i + bq shifts max, min and mean of its messages by q_i and
                // leaves std alone (gnn_builder_lib.h:1801-1850), so it is folded into x's weights too and the q GEMM is
                // not run at all.  Class 0 (no messages: the four aggregates are 0, not q) carries no S term.
                if (fo > 32) { // (any input width: whole 32-wide chunks take k_linear_dma's row-class mode, others the generic kernel's)
                    const size_t K5 = 5 * fi;
                    std::vector<float> wc((size_t)GNNB_DEG_CLASSES * fo * K5), bc((size_t)GNNB_DEG_CLASSES * fo);
                    std::vector<double> wa(4 * fi), sq(fi);
                    const float *wq = p[0]; // W_pre [F, 2F]: columns [0, F) act on the destination x_i (lib:1801-1802), bias p[1]
                    for (int c = 0; c < GNNB_DEG_CLASSES; c++) {
                        const float lg = logf((float)std::max(c, 1) + 1.0f);
                        const double amp = (double)(lg / d.pna_delta), att = (double)(d.pna_delta / lg);
                        float *dst = &wc[(size_t)c * fo * K5];
                        for (size_t o = 0; o < fo; o++) {
                            const float *src = &wm[o * K13];
                            for (size_t k = 0; k < 4 * fi; k++)
                                wa[k] = (double)src[fi + k] + amp * (double)src[5 * fi + k] + att * (double)src[9 * fi + k];
                            for (size_t k = 0; k < 4 * fi; k++)
                                dst[o * K5 + fi + k] = (float)wa[k];
                            double bsum = (double)bm[o];
                            if (c > 0) {
                                for (size_t k = 0; k < fi; k++)
                                    sq[k] = wa[k] + wa[fi + k] + wa[2 * fi + k]; // S_c[o][k]: max + min + mean
                                for (size_t j = 0; j < fi; j++) {               // (S_c Wq)[o][j] = sum_k S_c[o][k] Wq[k][j]
                                    double a = (double)src[j];
                                    for (size_t k = 0; k < fi; k++)
                                        a += sq[k] * (double)wq[k * 2 * fi + j];
                                    dst[o * K5 + j] = (float)a;
                                }
                                for (size_t k = 0; k < fi; k++)
                                    bsum += sq[k] * (double)p[1][k];
                            } else {
                                for (size_t j = 0; j < fi; j++)
                                    dst[o * K5 + j] = src[j];
                            }
                            bc[(size_t)c * fo + o] = (float)bsum;
                        }
                    }
                    conv_off[l].push_back(push(wc.data(), wc.size()));
                    conv_off[l].push_back(push(bc.data(), bc.size()));
                }
            }
            break;
        }
        pi += slots;
    }
    // k_gcn2_zf reads its 16-column slice of the last GCN layer's weight as MFMA B fragments: lane (li, lg) of the wave that
    // owns slice s takes W[16 s + li][16 q + 4 lg .. + 3] for q = 0 .. K/16 - 1.  Straight from the [out][in] matrix that is
    // 16 rows x 64 B per load instruction (half of every 128-B line unused, 32 MB of L2 traffic per launch over the chip);
    // a second copy in fragment order -- float4 index ((s K/16 + q) 4 + lg) 16 + li -- makes every load instruction one
    // contiguous KiB.  Rows past `out` are zero.
    size_t w1f_off = 0;
    bool have_w1f = false;
    if (d.conv_type == GNNB_CONV_GCN && d.num_layers == 2 && d.hidden_dim % 16 == 0 && d.hidden_dim <= 128 && d.out_dim <= 128) {
        const int K = d.hidden_dim, KQ = K / 16, NS = (d.out_dim + 15) / 16;
        std::vector<float> frag((size_t)NS * 16 * K, 0.0f);
        const float *w1 = &img[conv_off[1][0]]; // (the image copy: already on the fixed-point grid when fpx is set)
        for (int s = 0; s < NS; s++)
            for (int q = 0; q < KQ; q++)
                for (int lg = 0; lg < 4; lg++)
                    for (int li = 0; li < 16; li++) {
                        const int n = 16 * s + li;
                        if (n >= d.out_dim)
                            continue;
                        for (int e = 0; e < 4; e++)
                            frag[((((size_t)s * KQ + q) * 4 + lg) * 16 + li) * 4 + e] = w1[(size_t)n * K + 16 * q + 4 * lg + e];
                    }
        const bool fpx_save = fpx;
        (void)fpx_save;
        // (push() would quantise again: harmless -- the grid is idempotent)
        w1f_off = push(frag.data(), frag.size());
        have_w1f = true;
    }
    size_t gin_w_off = 0, gin_b_off = 0;
    bool have_gin = false;
    if (d.conv_type == GNNB_CONV_GIN && d.num_layers >= 2 && (d.hidden_dim == 32 || d.hidden_dim == 64 || d.hidden_dim == 128) &&
        d.out_dim <= d.hidden_dim && d.out_dim % 4 == 0) {
        const size_t h = d.hidden_dim, ho = d.out_dim;
        const int L = d.num_layers, nm = 2 * L - 1;
        std::vector<float> gw((size_t)nm * h * h, 0.0f), gb((size_t)nm * h, 0.0f);
        auto put = [&](int idx, size_t off_w, size_t off_b, size_t rows, size_t cols) { // [rows, cols] -> top-left of slot idx
            for (size_t r = 0; r < rows; r++)
                memcpy(&gw[(size_t)idx * h * h + r * h], &img[off_w + r * cols], cols * sizeof(float));
            memcpy(&gb[(size_t)idx * h], &img[off_b], rows * sizeof(float));
        };
        put(0, conv_off[0][2], conv_off[0][3], L == 1 ? ho : h, L == 1 ? ho : h); // (layer 0's second linear)
        for (int l = 1; l < L; l++) {
            const size_t fo = l == L - 1 ? ho : h;
            put(2 * l - 1, conv_off[l][0], conv_off[l][1], fo, h); // Wa [fo, h]
            put(2 * l, conv_off[l][2], conv_off[l][3], fo, fo);     // Wb [fo, fo]
        }
        gin_w_off = push(gw.data(), gw.size());
        gin_b_off = push(gb.data(), gb.size());
        have_gin = true;
    }
    for (int i = 0; i < d.mlp_num_linear; i++) {
        int din, dout;
        mlp_dims(d, i, &din, &dout);
        hw.push_back(push(host_params[pi], (size_t)din * dout));
        hb.push_back(push(host_params[pi + 1], (size_t)dout));
        pi += 2;
    }

    gnnb_model *m = new gnnb_model();
    m->desc = d;
    (void)hipGetDevice(&m->device);
    m->blob_floats = img.size();
    hipError_t e = hipMalloc((void **)&m->blob, std::max<size_t>(img.size(), 4) * sizeof(float));
    if (e == hipSuccess && !img.empty())
        e = hipMemcpy(m->blob, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (m->blob)
            (void)hipFree(m->blob);
        delete m;
        return fail(GNNB_ERR_HIP, "weight upload failed: %s", hipGetErrorString(e));
    }
    m->conv.resize(d.num_layers);
    for (int l = 0; l < d.num_layers; l++)
        for (size_t off : conv_off[l])
            m->conv[l].push_back(m->blob + off);
    if (have_w1f)
        m->zf_w1f = m->blob + w1f_off;
    if (have_gin) {
        m->gin_w = m->blob + gin_w_off;
        m->gin_b = m->blob + gin_b_off;
    }
    for (int i = 0; i < d.mlp_num_linear; i++) {
        m->head_w.push_back(m->blob + hw[i]);
        m->head_b.push_back(m->blob + hb[i]);
    }
    if (d.mlp_num_linear <= 8) { // (best effort: without the device copy the stack kernels leave the head to its own launch)
        const HeadArgs h = model_head_args(m);
        if (hipMalloc((void **)&m->head_dev, sizeof(HeadArgs)) != hipSuccess ||
            hipMemcpy(m->head_dev, &h, sizeof(HeadArgs), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            if (m->head_dev)
                (void)hipFree(m->head_dev);
            m->head_dev = nullptr;
        }
    }
    *out_model = m;
    return GNNB_OK;
}

void gnnb_model_destroy(gnnb_model *model)
{
    if (!model)
        return;
    if (model->blob)
        (void)hipFree(model->blob);
    if (model->head_dev)
        (void)hipFree(model->head_dev);
    delete model;
}

int gnnb_model_get_desc(const gnnb_model *model, gnnb_model_desc *out_desc)
{
    if (!model || !out_desc)
        return fail(GNNB_ERR_INVALID, "null argument");
    *out_desc = model->desc;
    return GNNB_OK;
}

// ---------------------------------------------------------------------------------------
int gnnb_workspace_create(const gnnb_model *model, int max_graphs, int max_nodes, int max_edges,
                          gnnb_workspace **out_ws)
{
    if (!out_ws)
        return fail(GNNB_ERR_INVALID, "null out_ws");
    *out_ws = nullptr;
    if (!model)
        return fail(GNNB_ERR_INVALID, "null model");
    if (max_graphs < 1 || max_nodes < 1 || max_edges < 0)
        return fail(GNNB_ERR_INVALID, "workspace capacities must be positive");
    const gnnb_model_desc &d = model->desc;

    int maxw = d.in_dim, aggw = 4, tmpw = 4;
    for (int l = 0; l < d.num_layers; l++) {
        const LayerDims ld = layer_dims(d, l);
        maxw = std::max(maxw, std::max(ld.fin, ld.fout));
        aggw = std::max(aggw, d.conv_type == GNNB_CONV_PNA ? 4 * ld.fin : ld.fin);
    }
    tmpw = std::max(tmpw, maxw);
    // the widest GEMM K of the conv layers: from 1024 on k_linear_dma may cut its tiles along K (stream-K) and needs a scratch,
    // which belongs to the workspace -- forwards of different workspaces never share one, whatever streams or graphs run them
    int max_k = 0;
    for (int l = 0; l < d.num_layers; l++) {
        const LayerDims ld = layer_dims(d, l);
        max_k = std::max(max_k, d.conv_type == GNNB_CONV_PNA ? 13 * ld.fin : d.conv_type == GNNB_CONV_SAGE ? 2 * ld.fin : std::max(ld.fin, ld.fout));
    }
    const bool want_sk = max_k >= 1024;
    // (the stage-cut planner's tables: opt-in -- carved only for workspaces created while the option is on; a workspace created
    // without them keeps equal tile counts whatever the option says later: round-5 advisor finding)
    const bool want_plan = options().stage_cut && (d.conv_type == GNNB_CONV_GCN || d.conv_type == GNNB_CONV_GIN) && d.num_layers >= 2;
    const int pooledw = d.num_pools * gnn_out_width(d);
    const int mlpw = std::max(d.mlp_hidden, d.mlp_out);
    const int min_tile_rows = 4;
    const size_t max_tiles = (size_t)max_nodes / min_tile_rows + 2;

    gnnb_workspace *ws = new gnnb_workspace();
    ws->desc = d;
    ws->max_graphs = max_graphs;
    ws->max_nodes = max_nodes;
    ws->max_edges = max_edges;
    (void)hipGetDevice(&ws->device);

    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t N = max_nodes, E = std::max(max_edges, 1), B = max_graphs;
    // (models whose last conv layer ends in the large-K segmented GEMM -- GraphSAGE -- pool in that GEMM's epilogue: a
    // node -> graph table and the buffer for the pieces of graphs that cross 32-row blocks)
    const bool pool_epi = (d.conv_type == GNNB_CONV_SAGE || d.conv_type == GNNB_CONV_PNA) && d.num_layers >= 1 && d.fpx_w <= 0;
    const size_t o_rp = carve((N + 1) * 4), o_col = carve(E * 4), o_eid = carve(E * 4), o_rec = carve(N * 32), o_dinv = carve(N * 4), o_amp = carve(N * 4),
                 o_att = carve(N * 4), o_gcoef = carve(N * 16), o_tile = carve((max_tiles + 1) * 4), o_tedge = carve((max_tiles + 1) * 4), o_gptr = carve((B + 1) * 4),
                 o_tgraph = carve((max_tiles + 1) * 4), o_err = carve(4), o_cut = carve((4096 + 1) * 16), o_scut = carve((1024 + 2) * 4),
                 o_plan = carve(want_plan ? (size_t)stage_cut_levels((int)max_tiles) * (max_tiles + 1) * 4 : 0),
                 o_dwork = carve(d.conv_type == GNNB_CONV_PNA ? 1024 * 16 * 4 : 0),
                 o_dperm = carve(d.conv_type == GNNB_CONV_PNA ? ((N + 127) / 128 + GNNB_DEG_CLASSES + 1) * 128 * 4 : 0),
                 o_dcls = carve(d.conv_type == GNNB_CONV_PNA ? ((N + 127) / 128 + GNNB_DEG_CLASSES + 1) * 4 : 0),
                 o_ngraph = carve(pool_epi ? N * 4 : 0), o_part = carve(pool_epi ? ((N + 31) / 32) * 2 * (size_t)gnn_out_width(d) * 8 : 0),
                 o_a0 = carve(N * maxw * 4), o_a1 = carve(N * maxw * 4), o_agg = carve(N * aggw * 4),
                 o_t0 = carve(N * tmpw * 4), o_t1 = carve(N * tmpw * 4),
                 o_pool = carve(B * pooledw * 4), o_m0 = carve(B * mlpw * 4),
                 o_m1 = carve(B * mlpw * 4), o_sk = carve(want_sk ? stream_k_scratch_bytes() : 0);
    ws->bytes = off;
    hipError_t e = hipMalloc((void **)&ws->blob, ws->bytes);
    if (e != hipSuccess) {
        delete ws;
        return fail(GNNB_ERR_HIP, "workspace allocation of %zu bytes failed: %s", off,
                    hipGetErrorString(e));
    }
    char *b = ws->blob;
    ws->t.row_ptr = (int32_t *)(b + o_rp);
    ws->t.col = (int32_t *)(b + o_col);
    ws->t.eid = (int32_t *)(b + o_eid);
    ws->t.node_rec = (int4 *)(b + o_rec);
    ws->t.dinv = (float *)(b + o_dinv);
    ws->t.amp = (float *)(b + o_amp);
    ws->t.att = (float *)(b + o_att);
    ws->t.gcoef = (float4 *)(b + o_gcoef);
    ws->t.tile_first = (int32_t *)(b + o_tile);
    ws->t.graph_ptr = (int32_t *)(b + o_gptr);
    ws->t.tile_edge = (int32_t *)(b + o_tedge);
    ws->t.tile_graph = (int32_t *)(b + o_tgraph);
    ws->t.err = (int32_t *)(b + o_err);
    ws->t.agg_cut = (int4 *)(b + o_cut);
    ws->t.agg_cut_n = 0;
    ws->t.stage_cut = (int32_t *)(b + o_scut);
    ws->t.stage_cut_n = 0;
    ws->t.stage_cut_cap = 1024;
    ws->plan_scratch = want_plan ? (int32_t *)(b + o_plan) : nullptr;
    ws->t.node_graph = pool_epi ? (int32_t *)(b + o_ngraph) : nullptr;
    if (d.conv_type == GNNB_CONV_PNA) {
        ws->deg_work = (int32_t *)(b + o_dwork);
        ws->deg_perm = (int32_t *)(b + o_dperm);
        ws->deg_tile_cls = (int32_t *)(b + o_dcls);
    }
    ws->pool_part = pool_epi ? (float2 *)(b + o_part) : nullptr;
    ws->act[0] = (float *)(b + o_a0);
    ws->act[1] = (float *)(b + o_a1);
    ws->agg = (float *)(b + o_agg);
    ws->tmp0 = (float *)(b + o_t0);
    ws->tmp1 = (float *)(b + o_t1);
    ws->pooled = (float *)(b + o_pool);
    ws->mlp[0] = (float *)(b + o_m0);
    ws->mlp[1] = (float *)(b + o_m1);
    (void)hipMemset(ws->t.err, 0, sizeof(int32_t));
    if (want_sk) {
        ws->sk = stream_k_scratch_at(b + o_sk);
        // arrival counters zero (as every launch leaves them), guard pattern behind them.  Synchronous memsets, not the null
        // stream + a synchronise: that would join -- and could disturb a capture in progress on -- every blocking stream of the
        // process (round-5 advisor finding)
        (void)stream_k_scratch_init_sync(b + o_sk);
    }
    // best effort: without the mapped word only gnnb_workspace_check reports a malformed batch
    ws->t.err_host_dev = nullptr;
    if (hipHostMalloc((void **)&ws->err_host, 64, hipHostMallocMapped) == hipSuccess && ws->err_host) {
        *ws->err_host = 0;
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, ws->err_host, 0) == hipSuccess)
            ws->t.err_host_dev = (int32_t *)dp;
    } else {
        ws->err_host = nullptr;
        (void)hipGetLastError();
    }
    *out_ws = ws;
    return GNNB_OK;
}

void gnnb_workspace_destroy(gnnb_workspace *ws)
{
    if (!ws)
        return;
    if (ws->blob)
        (void)hipFree(ws->blob);
    if (ws->stage)
        (void)hipFree(ws->stage);
    if (ws->err_host)
        (void)hipHostFree(ws->err_host);
    if (ws->ev_fork)
        (void)hipEventDestroy(ws->ev_fork);
    if (ws->ev_join)
        (void)hipEventDestroy(ws->ev_join);
    if (ws->side)
        (void)hipStreamDestroy(ws->side);
    delete ws;
}

size_t gnnb_workspace_bytes(const gnnb_workspace *ws) { return ws ? ws->bytes : 0; }

int gnnb_workspace_last_path(const gnnb_workspace *ws) { return ws ? ws->last_path : GNNB_PATH_NONE; }

int gnnb_workspace_set_large_segment(gnnb_workspace *ws, int first_graph, int first_node, int first_edge)
{
    if (!ws)
        return fail(GNNB_ERR_INVALID, "null workspace");
    if (first_graph < 0) { // no large segment
        ws->large_g = ws->large_n = ws->large_e = -1;
        return GNNB_OK;
    }
    if (first_node < 0 || first_edge < 0)
        return fail(GNNB_ERR_INVALID, "the large segment needs the node and edge offsets of its first graph");
    ws->large_g = first_graph;
    ws->large_n = first_node;
    ws->large_e = first_edge;
    return GNNB_OK;
}

int gnnb_workspace_set_max_graph_nodes(gnnb_workspace *ws, int n)
{
    if (!ws || n < 0)
        return fail(GNNB_ERR_INVALID, "bad argument to gnnb_workspace_set_max_graph_nodes");
    ws->max_graph_nodes = n;
    return GNNB_OK;
}

int gnnb_workspace_set_max_degree(gnnb_workspace *ws, int d)
{
    if (!ws || d < 0)
        return fail(GNNB_ERR_INVALID, "bad argument to gnnb_workspace_set_max_degree");
    ws->max_degree = d;
    return GNNB_OK;
}

static BatchTables small_segment(const gnnb_workspace *ws);

// ---------------------------------------------------------------------------------------
// Can this workspace's graph prep run as a guest of the forward's readout kernel (k_head_small's extra workgroups)?  The molecule
// path (promise <= 64 nodes) of a batch that needs NOTHING launched behind its tables: no stage cuts, no degree classes, no
// coefficient table -- the conditions below are the ones graph_prep_impl launches those under.
static bool guest_prep_eligible(const gnnb_workspace *ws, int num_nodes)
{
    if (!options().guest_prep || ws->max_graph_nodes <= 0 || ws->max_graph_nodes > 64 || ws->large_g >= 0 || num_nodes <= 0)
        return false;
    if (options().stage_cut && ws->plan_scratch)
        return false;
    if (ws->desc.conv_type == GNNB_CONV_PNA && ws->max_degree > 0 && ws->max_degree <= GNNB_DEG_MAX && options().pna_classes && ws->deg_perm &&
        ws->desc.fpx_w <= 0)
        return false;
    if (ws->desc.conv_type == GNNB_CONV_GCN) {
        const bool stack_expected = options().fuse_gcn2 && ws->desc.num_layers >= 2 && ws->max_graph_nodes > 0 && ws->large_g < 0 && ws->desc.fpx_w <= 0;
        if (!stack_expected)
            return false;
    }
    // (the row-balanced aggregate ranges are part of the prep kernel itself: nothing behind it)
    return true;
}

// defer != nullptr (and guest_prep_eligible): everything gnnb_graph_prep does EXCEPT the launch -- *defer receives the kernel's arguments
static int graph_prep_impl(gnnb_workspace *ws, const int32_t *coo_dev, const int32_t *node_ptr_dev, const int32_t *edge_ptr_dev,
                           int num_graphs, int num_nodes, int num_edges, float pna_delta, void *stream, PrepParams *defer);

int gnnb_graph_prep(gnnb_workspace *ws, const int32_t *coo_dev, const int32_t *node_ptr_dev,
                    const int32_t *edge_ptr_dev, int num_graphs, int num_nodes, int num_edges,
                    float pna_delta, void *stream)
{
    return graph_prep_impl(ws, coo_dev, node_ptr_dev, edge_ptr_dev, num_graphs, num_nodes, num_edges, pna_delta, stream, nullptr);
}

static int graph_prep_impl(gnnb_workspace *ws, const int32_t *coo_dev, const int32_t *node_ptr_dev, const int32_t *edge_ptr_dev,
                           int num_graphs, int num_nodes, int num_edges, float pna_delta, void *stream, PrepParams *defer)
{
    if (!ws || !node_ptr_dev || !edge_ptr_dev || (num_edges > 0 && !coo_dev))
        return fail(GNNB_ERR_INVALID, "null argument to gnnb_graph_prep");
    MathScope math_scope(ws->desc.math); // (the tile granularity depends on which stack kernel the mode selects)
    if (num_graphs < 0 || num_nodes < 0 || num_edges < 0)
        return fail(GNNB_ERR_INVALID, "negative batch size");
    if (num_graphs > ws->max_graphs || num_nodes > ws->max_nodes || num_edges > ws->max_edges)
        return fail(GNNB_ERR_CAPACITY,
                    "batch (%d graphs, %d nodes, %d edges) exceeds workspace (%d, %d, %d)",
                    num_graphs, num_nodes, num_edges, ws->max_graphs, ws->max_nodes, ws->max_edges);
    // lazy detection: a batch prepared EARLIER on this workspace was flagged on the device (edge leaving its graph,
    // broken ptr arrays, broken max_graph_nodes promise) and nobody called gnnb_workspace_check since.  Read from a
    // host-mapped word without synchronising: it reports what has already run, never the batch being enqueued now.
    if (ws->err_host && *(volatile int32_t *)ws->err_host != 0) {
        const int32_t seen = *(volatile int32_t *)ws->err_host;
        *(volatile int32_t *)ws->err_host = 0;
        // reported now: the device word is cleared as well (in stream order), or a later gnnb_workspace_check would blame
        // a good batch for it
        (void)hipMemsetAsync(ws->t.err, 0, sizeof(int32_t), (hipStream_t)stream);
        if (seen == GNNB_FLAG_RANGE)
            return fail(GNNB_ERR_RANGE, "an earlier forward on this workspace met a non-finite value in a reduced-precision math mode "
                                        "(fp16's range exceeded: its results were unspecified); run the model with math = 0");
        return fail(GNNB_ERR_GRAPH, "an earlier batch on this workspace was flagged as malformed (its results were "
                                    "unspecified); gnnb_workspace_check reports and clears the flags");
    }
    if (ws->large_g >= 0 && (ws->large_g > num_graphs || ws->large_n > num_nodes || ws->large_e > num_edges))
        return fail(GNNB_ERR_INVALID, "large segment (graph %d, node %d, edge %d) lies outside the batch (%d, %d, %d)",
                    ws->large_g, ws->large_n, ws->large_e, num_graphs, num_nodes, num_edges);
    BatchTables &t = ws->t;
    t.promise_graphs = ws->large_g >= 0 ? ws->large_g : num_graphs; // the promise covers graphs [0, promise_graphs)
    t.large_n = ws->large_g >= 0 ? ws->large_n : -1; // (checked against node_ptr / edge_ptr on the device: flag 16)
    t.large_e = ws->large_g >= 0 ? ws->large_e : -1;
    t.tile_lo = 0;
    t.node_ptr = node_ptr_dev;
    t.num_graphs = num_graphs;
    t.num_nodes = num_nodes;
    t.num_edges = num_edges;
    t.max_graph_nodes_hint = ws->max_graph_nodes;
    t.tile_rows = std::max((int)options().tile_rows, 4);
    // A 2-layer GCN with a promise takes the fused stack only if a whole tile (tile_rows - 1 + largest graph) fits
    // one 64-row stage (48 in the bf16x6 mode): for graphs of 50..61 nodes finer tiles (8, 4) keep that path open
    if (options().fuse_gcn2 && (ws->desc.conv_type == GNNB_CONV_GCN || ws->desc.conv_type == GNNB_CONV_GIN) && ws->desc.num_layers >= 2 &&
        ws->max_graph_nodes > 0) {
        // (a 2-layer fp32 GCN stack runs k_gcn2_zf with its 96-row stages; everything else k_gcn2_fused)
        const bool zf = ws->desc.conv_type == GNNB_CONV_GCN && ws->desc.num_layers == 2 && options().fuse_zf;
        const bool bf6 = !zf && launch_math() && ws->desc.conv_type == GNNB_CONV_GCN && ws->desc.num_layers == 2; // (the only bf16x6 stack form)
        const int stage_rows = zf ? zf_stage_rows(ws->desc.in_dim, ws->max_graph_nodes) : bf6 ? GNNB_G2_STAGE_ROWS_BF6 : GNNB_G2_STAGE_ROWS;
        while (t.tile_rows > 4 && ws->max_graph_nodes + t.tile_rows - 1 > stage_rows)
            t.tile_rows >>= 1;
        // very large batches: coarser tiles (while a tile still fits a stage) keep the per-workgroup tile table in LDS
        const long tile_cap = zf ? gcn2_zf_tile_capacity(ws->desc.in_dim, ws->max_graph_nodes) : gcn2_fused_tile_capacity();
        while ((num_nodes + t.tile_rows - 1) / t.tile_rows > tile_cap && ws->max_graph_nodes + 2 * t.tile_rows - 1 <= stage_rows)
            t.tile_rows <<= 1;
    }
    t.num_tiles = (num_nodes + t.tile_rows - 1) / t.tile_rows;
    // row-balanced ranges for the gather-aggregate kernels (one per workgroup of the ring kernel's grid on this device):
    // not for a batch that is expected on the stack kernels entirely (their graph prep is on the pipeline's critical path
    // and pays for every instruction), not with a large segment (its aggregates walk a tile sub-range)
    {
        const bool stack_expected = options().fuse_gcn2 && (ws->desc.conv_type == GNNB_CONV_GCN || ws->desc.conv_type == GNNB_CONV_GIN) &&
                                    ws->desc.num_layers >= 2 && ws->max_graph_nodes > 0 && ws->desc.fpx_w <= 0;
        const int rings = aggregate_ring_grid();
        t.agg_cut_n = (options().agg_balance && !stack_expected && ws->large_g < 0 && rings <= 4096 && (rings & (rings - 1)) == 0 &&
                       t.num_tiles >= rings) ? rings : 0;
    }
    if (!(pna_delta > 0.0f))
        pna_delta = 1.0f;
    // the degree scalers (amp / att) are only read by PNA layers: a model-bound workspace of another conv type
    // skips their computation and their 8 B/node of writes (delta <= 0 tells the kernel)
    const float prep_delta = ws->desc.conv_type == GNNB_CONV_PNA ? pna_delta : -1.0f;
    // GCN: an explicit self-loop edge is not entered into the tables (PyG's gcn_norm replaces the self loops of the
    // input by exactly one per node; the reference C++ would count it on top of its own self term, see gnnb_hip.h)
    const int drop_self = ws->desc.conv_type == GNNB_CONV_GCN ? 1 : 0;
    const bool deferred = defer && guest_prep_eligible(ws, num_nodes);
    if (deferred)
        *defer = make_prep_params(coo_dev, node_ptr_dev, edge_ptr_dev, t, prep_delta, drop_self);
    else {
        if (defer)
            return fail(GNNB_ERR_INVALID, "graph prep deferred for a workspace that is not eligible");
        GNNB_HIP_TRY(launch_graph_prep(coo_dev, node_ptr_dev, edge_ptr_dev, t, prep_delta, drop_self,
                                       (hipStream_t)stream));
    }
    ws->prepared = true;
    ws->prep_delta = prep_delta > 0.0f ? prep_delta : 0.0f;
    if (deferred) { // (eligible = nothing below would be launched: the tables do not exist yet)
        t.stage_cut_n = 0;
        ws->gcoef_ready = false;
        ws->deg_ready = false;
        return GNNB_OK;
    }
    // the conv-stack kernel's workgroup runs as whole stages of the global greedy stage list (k_plan.hip), right behind the tables
    // on the prep stream: for the batches that k_gcn2_fused takes (GIN stacks, GCN stacks deeper than two layers, the bf16x6 mode)
    t.stage_cut_n = 0;
    if (options().stage_cut && options().fuse_gcn2 && ws->plan_scratch && ws->max_graph_nodes > 0 && ws->desc.fpx_w <= 0 && num_nodes > 0) {
        const bool zf_route = ws->desc.conv_type == GNNB_CONV_GCN && ws->desc.num_layers == 2 && options().fuse_zf;
        const bool bf6 = !zf_route && launch_math() && ws->desc.conv_type == GNNB_CONV_GCN && ws->desc.num_layers == 2;
        const int cap = bf6 ? GNNB_G2_STAGE_ROWS_BF6 : GNNB_G2_STAGE_ROWS;
        const BatchTables ts = small_segment(ws);
        const int grid = gcn2_fused_grid(ts.num_tiles);
        if (!zf_route && ws->max_graph_nodes + t.tile_rows - 1 <= cap && grid <= t.stage_cut_cap && ts.num_tiles > 0) {
            GNNB_HIP_TRY(launch_stage_cut(t.tile_first, ts.num_tiles, ts.num_nodes, cap, grid, gcn2_fused_tile_window(), ws->plan_scratch,
                                          t.stage_cut, (hipStream_t)stream));
            t.stage_cut_n = grid;
        }
    }
    ws->gcoef_ready = false;
    // PNA under a degree promise: the rows sorted into degree classes, right behind the tables on the prep stream
    ws->deg_ready = false;
    if (ws->desc.conv_type == GNNB_CONV_PNA && ws->max_degree > 0 && ws->max_degree <= GNNB_DEG_MAX && options().pna_classes &&
        ws->deg_perm && ws->large_g < 0 && ws->desc.fpx_w <= 0) {
        ws->deg_max_tiles = (num_nodes + 127) / 128 + GNNB_DEG_CLASSES;
        GNNB_HIP_TRY(launch_degree_classes(t, ws->max_degree, ws->deg_work, ws->deg_perm, ws->deg_tile_cls, ws->deg_max_tiles,
                                           (hipStream_t)stream));
        ws->deg_ready = num_nodes > 0; // (an empty batch has no class tables: launch_degree_classes returns before it writes any)
        ws->deg_delta = pna_delta;
    }
    // The GCN coefficient table (dinv_i dinv_j of the four inline sources; read by every layer-wise GCN aggregate) is
    // produced HERE, on the prep stream right behind the tables, whenever the batch can run layer by layer -- so that a
    // forward captured into a hipGraph contains no lazily launched table kernel and aggregates on other streams that are
    // ordered against the prep see a finished table.  Only a workspace whose whole batch is expected on the LDS-resident
    // stack kernels (promise set, no large segment) skips it; should that forward fall back after all, ensure_gcoef
    // launches the table kernel in front of the first aggregate (the one lazy case left).
    if (ws->desc.conv_type == GNNB_CONV_GCN && num_nodes > 0) {
        const bool stack_expected = options().fuse_gcn2 && ws->desc.num_layers >= 2 && ws->max_graph_nodes > 0 &&
                                    ws->large_g < 0 && ws->desc.fpx_w <= 0;
        if (!stack_expected) {
            GNNB_HIP_TRY(launch_gcn_coef(ws->t, (hipStream_t)stream));
            ws->gcoef_ready = true;
        }
    }
    return GNNB_OK;
}

// the GCN coefficient table of the prepared batch, once per batch, in front of the first layer-wise GCN aggregate
static int ensure_gcoef(gnnb_workspace *ws, void *stream)
{
    if (ws->gcoef_ready)
        return GNNB_OK;
    GNNB_HIP_TRY(launch_gcn_coef(ws->t, (hipStream_t)stream));
    ws->gcoef_ready = true;
    return GNNB_OK;
}

int gnnb_workspace_check(gnnb_workspace *ws, void *stream)
{
    if (!ws || !ws->prepared)
        return fail(GNNB_ERR_INVALID, "workspace has no prepared batch");
    int32_t err = 0;
    GNNB_HIP_TRY(hipMemcpyAsync(&err, ws->t.err, sizeof(err), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GNNB_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (err != 0) {
        (void)hipMemsetAsync(ws->t.err, 0, sizeof(int32_t), (hipStream_t)stream); // reset on read
        (void)hipStreamSynchronize((hipStream_t)stream);
    }
    if (ws->err_host)
        *(volatile int32_t *)ws->err_host = 0;
    if (err & ~GNNB_FLAG_RANGE)
        return fail(GNNB_ERR_GRAPH, "malformed batch (flags 0x%x): 1/2 ptr arrays not monotone/complete, 4 an edge leaves "
                                    "its graph, 8 a graph exceeds the max_graph_nodes promise, 16 the large-segment offsets "
                                    "disagree with node_ptr / edge_ptr, 32 a node exceeds the max_degree promise, 64 a reduced-precision "
                                    "kernel produced a non-finite value", err);
    if (err & GNNB_FLAG_RANGE)
        return fail(GNNB_ERR_RANGE, "a reduced-precision math mode (bf16x3 / f16x3) produced a non-finite value since the last check: an "
                                    "activation or a weight beyond fp16's range (65504), or non-finite inputs; the results of that forward "
                                    "are unspecified -- run the model with math = 0 (flag 0x40)");
    return GNNB_OK;
}

// CSR slots no row owns (edges dropped by graph prep -- explicit self loops on a GCN workspace -- leave a gap at the end of
// their graph's segment, whose slots hold stale data) read -1 in the host copies
static int mark_unused_slots(gnnb_workspace *ws, int32_t *slots, hipStream_t s)
{
    const int N = ws->t.num_nodes, E = ws->t.num_edges;
    if (!slots || E <= 0)
        return GNNB_OK;
    std::vector<int32_t> rec((size_t)std::max(N, 1) * 8);
    if (N > 0)
        GNNB_HIP_TRY(hipMemcpyAsync(rec.data(), ws->t.node_rec, (size_t)N * 32, hipMemcpyDeviceToHost, s));
    GNNB_HIP_TRY(hipStreamSynchronize(s));
    std::vector<char> used((size_t)E, 0);
    for (int v = 0; v < N; v++) {
        const long start = rec[(size_t)v * 8], deg = rec[(size_t)v * 8 + 1];
        for (long k = std::max(start, 0L); k < std::min(start + deg, (long)E); k++)
            used[(size_t)k] = 1;
    }
    for (int k = 0; k < E; k++)
        if (!used[(size_t)k])
            slots[k] = -1;
    return GNNB_OK;
}

int gnnb_graph_tables_to_host(gnnb_workspace *ws, int32_t *row_ptr, int32_t *col, int32_t *in_deg,
                              void *stream)
{
    if (!ws || !ws->prepared)
        return fail(GNNB_ERR_INVALID, "workspace has no prepared batch");
    hipStream_t s = (hipStream_t)stream;
    const int N = ws->t.num_nodes, E = ws->t.num_edges;
    std::vector<int32_t> rec(in_deg ? (size_t)N * 8 : 0); // node records {start, degree, j0, j1}{j2, j3, -, -}
    if (row_ptr)
        GNNB_HIP_TRY(hipMemcpyAsync(row_ptr, ws->t.row_ptr, ((size_t)N + 1) * 4, hipMemcpyDeviceToHost, s));
    if (in_deg && N > 0)
        GNNB_HIP_TRY(hipMemcpyAsync(rec.data(), ws->t.node_rec, (size_t)N * 32, hipMemcpyDeviceToHost, s));
    if (col && E > 0)
        GNNB_HIP_TRY(hipMemcpyAsync(col, ws->t.col, (size_t)E * 4, hipMemcpyDeviceToHost, s));
    GNNB_HIP_TRY(hipStreamSynchronize(s));
    if (in_deg)
        for (int i = 0; i < N; i++)
            in_deg[i] = rec[(size_t)i * 8 + 1];
    return mark_unused_slots(ws, col, s);
}

int gnnb_edge_index_table_to_host(gnnb_workspace *ws, int32_t *edge_index_table, void *stream)
{
    if (!ws || !ws->prepared || !edge_index_table)
        return fail(GNNB_ERR_INVALID, "gnnb_edge_index_table_to_host needs a prepared batch and an output array");
    if (ws->t.num_edges > 0)
        GNNB_HIP_TRY(hipMemcpyAsync(edge_index_table, ws->t.eid, (size_t)ws->t.num_edges * 4, hipMemcpyDeviceToHost,
                                    (hipStream_t)stream));
    GNNB_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return mark_unused_slots(ws, edge_index_table, (hipStream_t)stream);
}

int gnnb_aggregate_edges(gnnb_workspace *ws, const float *x_dev, const float *edge_term_dev, float *out_dev,
                         int width, float eps, void *stream)
{
    if (!ws || !ws->prepared)
        return fail(GNNB_ERR_INVALID, "gnnb_aggregate_edges needs a prepared batch (gnnb_graph_prep)");
    if (!x_dev || !out_dev || width < 1 || (ws->t.num_edges > 0 && !edge_term_dev))
        return fail(GNNB_ERR_INVALID, "bad argument to gnnb_aggregate_edges");
    GNNB_HIP_TRY(launch_aggregate_edges(ws->t, x_dev, edge_term_dev, out_dev, width, eps, (hipStream_t)stream));
    return GNNB_OK;
}

int gnnb_pna_product_aggregate(gnnb_workspace *ws, const float *x_dev, const float *wb_dev, int ldw, float *out_dev, int width,
                               void *stream)
{
    if (!ws || !ws->prepared)
        return fail(GNNB_ERR_INVALID, "gnnb_pna_product_aggregate needs a prepared batch (gnnb_graph_prep)");
    MathScope math_scope(ws->desc.math);
    if (!x_dev || !wb_dev || !out_dev || width < 1 || ldw < width)
        return fail(GNNB_ERR_INVALID, "bad argument to gnnb_pna_product_aggregate");
    hipError_t he = launch_pna_pagg(ws->t, x_dev, width, wb_dev, ldw, out_dev, (hipStream_t)stream);
    if (he == hipErrorNotSupported)
        return fail(GNNB_ERR_INVALID, "gnnb_pna_product_aggregate takes widths 128 / 64 / 32, 16-byte aligned operands and a workspace "
                                      "whose max_graph_nodes promise fits a 64-row stage (promise + tile rows - 1 <= 64), without a large "
                                      "segment; the option pna_pagg must be on (it is %s)", options().pna_pagg ? "on" : "OFF");
    GNNB_HIP_TRY(he);
    return GNNB_OK;
}

int gnnb_aggregate(gnnb_workspace *ws, int agg_kind, const float *x_dev, const float *self_dev,
                   float *out_dev, int width, float eps, void *stream)
{
    if (!ws || !ws->prepared)
        return fail(GNNB_ERR_INVALID, "gnnb_aggregate needs a prepared batch (gnnb_graph_prep)");
    if (!x_dev || !out_dev || width < 1)
        return fail(GNNB_ERR_INVALID, "bad argument to gnnb_aggregate");
    if (agg_kind < GNNB_AGG_GCN || agg_kind > GNNB_AGG_COPY)
        return fail(GNNB_ERR_INVALID, "unknown aggregate kind %d", agg_kind);
    // (PNA with self_dev == NULL: no destination term -- the statistics of p_j alone, what the degree-class form aggregates)
    if (ws->t.num_nodes == 0)
        return GNNB_OK;
    if (agg_kind == GNNB_AGG_GCN) {
        int rc = ensure_gcoef(ws, stream);
        if (rc != GNNB_OK)
            return rc;
    }
    GNNB_HIP_TRY(launch_aggregate(ws->t, agg_kind, x_dev, self_dev, out_dev, width, eps,
                                  (hipStream_t)stream));
    return GNNB_OK;
}

static int build_gemm(GemmArgs &g, const gnnb_gemm_seg *segs, int num_segs, const float *w, int ldw)
{
    if (num_segs < 1 || num_segs > 4 || !segs)
        return fail(GNNB_ERR_INVALID, "gnnb_linear takes 1..4 segments");
    memset(&g, 0, sizeof(g));
    g.nseg = num_segs;
    int koff = 0;
    g.cpre[0] = 0;
    for (int s = 0; s < 4; s++) {
        if (s < num_segs) {
            if (!segs[s].a_dev || segs[s].k < 1 || segs[s].lda < segs[s].k)
                return fail(GNNB_ERR_INVALID, "bad GEMM segment %d", s);
            g.a[s] = segs[s].a_dev;
            g.rs[s] = segs[s].rowscale_dev;
            g.lda[s] = segs[s].lda;
            g.k[s] = segs[s].k;
            g.koff[s] = koff;
            g.avec[s] = (segs[s].k % 4 == 0) && (segs[s].lda % 4 == 0) && (((uintptr_t)segs[s].a_dev & 15) == 0);
            g.wvec[s] = (segs[s].k % 4 == 0) && (ldw % 4 == 0) && (koff % 4 == 0) && (((uintptr_t)w & 15) == 0);
            g.cpre[s + 1] = g.cpre[s] + (segs[s].k + 31) / 32;
            koff += segs[s].k;
        } else {
            g.cpre[s + 1] = g.cpre[s];
        }
    }
    if (koff > ldw)
        return fail(GNNB_ERR_INVALID, "segments span %d columns but ldw = %d", koff, ldw);
    return GNNB_OK;
}

// sk_owned: the calling workspace's stream-K scratch (nullptr: the standalone entry -- one per (device, stream), never under capture)
static int linear_segs(const StreamK *sk_owned, const gnnb_gemm_seg *segs, int num_segs, const float *w_dev, int ldw,
                       const float *bias_dev, const float *skip_dev, float *y_dev, int M, int N, int act, void *stream)
{
    if (!w_dev || !y_dev || M < 0 || N < 1)
        return fail(GNNB_ERR_INVALID, "bad argument to gnnb_linear");
    if (act < 0 || act > GNNB_ACT_NONE)
        return fail(GNNB_ERR_INVALID, "unknown activation %d", act);
    GemmArgs g;
    int rc = build_gemm(g, segs, num_segs, w_dev, ldw);
    if (rc != GNNB_OK)
        return rc;
    GNNB_HIP_TRY(launch_linear(g, w_dev, ldw, bias_dev, skip_dev, y_dev, M, N, act, (hipStream_t)stream, nullptr, nullptr, sk_owned));
    return GNNB_OK;
}

int gnnb_linear(const gnnb_gemm_seg *segs, int num_segs, const float *w_dev, int ldw,
                const float *bias_dev, const float *skip_dev, float *y_dev, int M, int N, int act,
                void *stream)
{
    return linear_segs(nullptr, segs, num_segs, w_dev, ldw, bias_dev, skip_dev, y_dev, M, N, act, stream);
}

int gnnb_debug_stream_k_guard(gnnb_workspace *ws, void *stream)
{
    // (a workspace whose model has no K >= 1024 layer owns no scratch: said so, instead of silently checking the stand-alone
    // scratch of (device, stream) or nothing at all -- round-5 advisor finding)
    if (ws && !ws->sk.part)
        return fail(GNNB_ERR_INVALID, "this workspace owns no stream-K scratch (no layer of its model has K >= 1024): nothing to check; "
                                      "pass ws = NULL for the stand-alone gnnb_linear scratch of (device, stream)");
    const int ok = stream_k_guard_intact(ws ? &ws->sk : nullptr, (hipStream_t)stream);
    if (ok < 0)
        return fail(GNNB_ERR_HIP, "reading the stream-K scratch back failed");
    if (ok == 0)
        return fail(GNNB_ERR_INVALID, "stream-K scratch: an arrival counter was left non-zero or the guard region behind the counters was written");
    return GNNB_OK;
}

int gnnb_global_pool(gnnb_workspace *ws, const float *x_dev, int d, const int32_t *pools,
                     int num_pools, float *out_dev, void *stream)
{
    if (!ws || !ws->prepared)
        return fail(GNNB_ERR_INVALID, "gnnb_global_pool needs a prepared batch");
    if (!x_dev || !out_dev || !pools || d < 1 || num_pools < 1 || num_pools > 3)
        return fail(GNNB_ERR_INVALID, "bad argument to gnnb_global_pool");
    for (int i = 0; i < num_pools; i++)
        if (pools[i] < 0 || pools[i] > GNNB_POOL_MAX)
            return fail(GNNB_ERR_INVALID, "unsupported pooling %d", pools[i]);
    GNNB_HIP_TRY(launch_global_pool(x_dev, ws->t.graph_ptr, ws->t.num_graphs, d, pools, num_pools,
                                    out_dev, (hipStream_t)stream));
    return GNNB_OK;
}

// ---------------------------------------------------------------------------------------
static int linear1(const float *a, int lda, int k, const float *w, int ldw, const float *bias,
                   const float *skip, float *y, int M, int N, int act, void *stream)
{
    gnnb_gemm_seg seg = {a, nullptr, lda, k};
    return gnnb_linear(&seg, 1, w, ldw, bias, skip, y, M, N, act, stream);
}


// Middle layers of a GCN stack for the fused kernel: every one hidden -> hidden, weights / biases at one constant
// stride in the model blob (it is laid out layer by layer, so they are -- checked, not assumed).  nl = 0: not eligible.
static G2Deep gcn_stack_middle_layers(const gnnb_model *model)
{
    const gnnb_model_desc &d = model->desc;
    G2Deep g;
    g.nl = 0;
    const int L = d.num_layers;
    if (d.conv_type == GNNB_CONV_GIN && L >= 2 && L <= GNNB_MAX_LAYERS && model->gin_w && model->gin_b) {
        // (the execution-order copy made at upload: one stride by construction, the last layer padded to hidden x hidden)
        g.wmid = model->gin_w;
        g.bmid = model->gin_b;
        g.mid_stride = (long)d.hidden_dim * d.hidden_dim;
        g.bmid_stride = (long)d.hidden_dim;
        g.gin = 1;
        g.eps = d.gin_eps;
        g.skip = d.skip ? 1 : 0;
        g.nl = L;
        return g;
    }
    if (d.conv_type != GNNB_CONV_GCN || L < 2 || L > GNNB_MAX_LAYERS)
        return g;
    if (L > 2) {
        g.wmid = model->conv[1][0];
        g.bmid = model->conv[1][1];
        if (L > 3) {
            g.mid_stride = (long)(model->conv[2][0] - model->conv[1][0]);
            g.bmid_stride = (long)(model->conv[2][1] - model->conv[1][1]);
        }
        for (int l = 1; l + 1 < L; l++)
            if (model->conv[l][0] != g.wmid + (long)(l - 1) * g.mid_stride || model->conv[l][1] != g.bmid + (long)(l - 1) * g.bmid_stride)
                return g;
    }
    g.skip = d.skip ? 1 : 0;
    g.nl = L;
    return g;
}

// The conv layers one by one (gather-aggregate + GEMM kernels) on the node rows [row_lo, N) of the prepared batch:
// row_lo = 0 is the whole batch; row_lo > 0 the caller's large segment (gnnb_workspace_set_large_segment), whose first
// node tile is tile_lo.  Aggregations index the batch-global buffers (sources are batch-global ids) and walk the tiles
// from tile_lo; the GEMMs take the row range as a pointer offset.  *out_cur = the last layer's output matrix ([N, width],
// rows below row_lo untouched).
static int run_conv_layers(const gnnb_model *model, gnnb_workspace *ws, const float *x_dev, int row_lo, int tile_lo,
                           const float **out_cur, void *stream, bool *pooled_in_epilogue = nullptr)
{
    if (pooled_in_epilogue)
        *pooled_in_epilogue = false;
    const gnnb_model_desc &d = model->desc;
    const int N = ws->t.num_nodes, M = N - row_lo;
    int rc;
    const bool fpx = d.fpx_w > 0;
    auto quant = [&](float *buf, size_t n) -> int {
        if (!fpx)
            return GNNB_OK;
        GNNB_HIP_TRY(launch_quantize(buf, buf, n, d.fpx_w, d.fpx_i, (hipStream_t)stream));
        return GNNB_OK;
    };
    BatchTables tv = ws->t;
    tv.tile_lo = tile_lo;
    const StreamK *const sko = ws->sk.part ? &ws->sk : nullptr; // this workspace's stream-K scratch
    auto gnnb_linear = [&](const gnnb_gemm_seg *segs, int num_segs, const float *w_dev, int ldw, const float *bias_dev, const float *skip_dev,
                           float *y_dev, int M_, int N_, int act, void *st) -> int {
        return linear_segs(sko, segs, num_segs, w_dev, ldw, bias_dev, skip_dev, y_dev, M_, N_, act, st);
    };
    auto linear1 = [&](const float *a, int lda, int k, const float *w, int ldw, const float *bias, const float *skip_, float *y, int M_, int N_,
                       int act, void *st) -> int {
        gnnb_gemm_seg seg = {a, nullptr, lda, k};
        return linear_segs(sko, &seg, 1, w, ldw, bias, skip_, y, M_, N_, act, st);
    };
    auto aggregate = [&](int kind, const float *x, const float *selfq, float *out, int w, float eps) -> int {
        if (M <= 0)
            return GNNB_OK;
        if (kind == GNNB_AGG_GCN) {
            int rc2 = ensure_gcoef(ws, stream);
            if (rc2 != GNNB_OK)
                return rc2;
        }
        GNNB_HIP_TRY(launch_aggregate(tv, kind, x, selfq, out, w, eps, (hipStream_t)stream));
        return GNNB_OK;
    };
    auto R = [&](const float *p, int width) { return p ? p + (size_t)row_lo * width : p; }; // row range of a [N, width] matrix
    auto Rw = [&](float *p, int width) { return p + (size_t)row_lo * width; };
    const bool whole = row_lo == 0;
    const float *cur = x_dev;
    int which = 0;
    bool mean_ready = false; // GraphSAGE: ws->agg already holds mean_j of the current layer's input rows (k_sage_first_mean)
    for (int l = 0; l < d.num_layers; l++) {
        const LayerDims ld = layer_dims(d, l);
        const int fi = ld.fin, fo = ld.fout;
        const std::vector<const float *> &p = model->conv[l];
        // skip connection on middle layers only (models.py:562-564); fused into the GEMM epilogue
        const float *skip = (d.skip && l != 0 && l != d.num_layers - 1) ? cur : nullptr;
        // (GraphSAGE / PNA: derived weight slots of such a layer carry the skip connection as + I on x's own weights: gnnb_model_create)
        const bool skip_fold = skip != nullptr && fi == fo && !fpx;
        float *nxt = ws->act[which];
        if ((const float *)nxt == cur) { // never write the buffer being read
            which ^= 1;
            nxt = ws->act[which];
        }
        switch (d.conv_type) {
        case GNNB_CONV_GCN:
            // aggregate at the input width, then transform (the reference's order, lib:1346-1379)
            if (whole && options().fuse_narrow && fi <= 32) {
                hipError_t he = launch_conv_gather(ws->t, GNNB_AGG_GCN, 0.f, cur, fi, fi, p[0], fi, p[1], skip, nxt,
                                                   fo, d.activation, (hipStream_t)stream);
                if (he == hipSuccess)
                    break;
                if (he != hipErrorNotSupported)
                    return fail(GNNB_ERR_HIP, "fused narrow conv launch failed: %s", hipGetErrorString(he));
            }
            if ((rc = aggregate(GNNB_AGG_GCN, cur, nullptr, ws->agg, fi, 0.f)))
                return rc;
            if ((rc = linear1(R(ws->agg, fi), fi, fi, p[0], fi, p[1], R(skip, fi), Rw(nxt, fo), M, fo, d.activation, stream)))
                return rc;
            break;
        case GNNB_CONV_GIN: {
            bool fused = false;
            if (whole && options().fuse_narrow && fi <= 32) {
                hipError_t he = launch_conv_gather(ws->t, GNNB_AGG_SUM, d.gin_eps, cur, fi, fi, p[0], fi, p[1], nullptr,
                                                   ws->tmp0, fo, GNNB_ACT_RELU, (hipStream_t)stream);
                if (he == hipSuccess)
                    fused = true;
                else if (he != hipErrorNotSupported)
                    return fail(GNNB_ERR_HIP, "fused narrow conv launch failed: %s", hipGetErrorString(he));
            }
            if (!fused) {
                if ((rc = aggregate(GNNB_AGG_SUM, cur, nullptr, ws->agg, fi, d.gin_eps)))
                    return rc;
                if ((rc = linear1(R(ws->agg, fi), fi, fi, p[0], fi, p[1], nullptr, Rw(ws->tmp0, fo), M, fo, GNNB_ACT_RELU, stream)))
                    return rc;
            }
            if ((rc = linear1(R(ws->tmp0, fo), fo, fo, p[2], fo, p[3], R(skip, fo), Rw(nxt, fo), M, fo, d.activation, stream)))
                return rc;
            break;
        }
        case GNNB_CONV_SAGE: {
            if (whole && options().fuse_narrow && 2 * fi <= 32 && l + 1 < d.num_layers && skip == nullptr && !fpx) {
                // narrow input AND a layer behind it: the stage's output rows stay in LDS and the next layer's mean aggregate is
                // taken from there -- its aggregate kernel (a full read and write of [N, fo]) is not run
                hipError_t he = launch_sage_first_mean(ws->t, cur, fi, p[0], 2 * fi, p[1], nxt, ws->agg, fo, d.activation, (hipStream_t)stream);
                if (he == hipSuccess) {
                    mean_ready = true;
                    break;
                }
                if (he != hipErrorNotSupported)
                    return fail(GNNB_ERR_HIP, "first-layer + mean launch failed: %s", hipGetErrorString(he));
            }
            if (whole && options().fuse_narrow && 2 * fi <= 32) {
                // narrow input: [mean_j x_j | x_i] is produced inside the GEMM's A stage (K = 2 F_in)
                hipError_t he = launch_conv_gather(ws->t, GNNB_AGG_MEAN, 0.f, cur, fi, 2 * fi, p[0], 2 * fi, p[1], skip, nxt,
                                                   fo, d.activation, (hipStream_t)stream, fi);
                if (he == hipSuccess)
                    break;
                if (he != hipErrorNotSupported)
                    return fail(GNNB_ERR_HIP, "fused narrow conv launch failed: %s", hipGetErrorString(he));
            }
            if (!mean_ready && (rc = aggregate(GNNB_AGG_MEAN, cur, nullptr, ws->agg, fi, 0.f)))
                return rc;
            mean_ready = false;
            gnnb_gemm_seg segs[2] = {{R(ws->agg, fi), nullptr, fi, fi}, {R(cur, fi), nullptr, fi, fi}};
            // the LAST layer of a whole-batch run: global pooling in the GEMM's epilogue -- its [N, out] output is never
            // written and the separate pooling pass (a full read of it) disappears (reference: compute_gnn_head's last
            // layer + compute_global_graph_pooling, templates/model.cpp.jinja:151-449).  Falls back when the GEMM shape
            // has no such epilogue.
            if (pooled_in_epilogue && whole && l == d.num_layers - 1 && !fpx && options().fuse_pool && ws->t.node_graph && ws->pool_part) {
                GemmArgs g;
                if ((rc = build_gemm(g, segs, 2, p[0], 2 * fi)))
                    return rc;
                PoolEpilogue pe;
                pe.node_graph = ws->t.node_graph;
                pe.graph_ptr = ws->t.graph_ptr;
                pe.pooled = ws->pooled;
                pe.part = ws->pool_part;
                pe.num_graphs = ws->t.num_graphs;
                pe.np = d.num_pools;
                for (int k = 0; k < 3; k++)
                    pe.pools[k] = k < d.num_pools ? d.pools[k] : 0;
                hipError_t he = launch_linear(g, p[0], 2 * fi, p[1], nullptr, nxt, M, fo, d.activation, (hipStream_t)stream, &pe);
                if (he == hipSuccess) {
                    GNNB_HIP_TRY(launch_pool_combine(pe, M, fo, (hipStream_t)stream));
                    *pooled_in_epilogue = true;
                    break;
                }
                if (he != hipErrorNotSupported)
                    return fail(GNNB_ERR_HIP, "pooling GEMM launch failed: %s", hipGetErrorString(he));
            }
            if (skip_fold && p.size() >= 3 && options().fold_skip) { // (slot 2: [Wl | Wr + I])
                if ((rc = gnnb_linear(segs, 2, p[2], 2 * fi, p[1], nullptr, Rw(nxt, fo), M, fo, d.activation, stream)))
                    return rc;
                break;
            }
            if ((rc = gnnb_linear(segs, 2, p[0], 2 * fi, p[1], R(skip, fi), Rw(nxt, fo), M, fo, d.activation, stream)))
                return rc;
            break;
        }
        case GNNB_CONV_PNA: {
            // a narrow input (the first layer): the whole layer in one kernel, whole graphs staged in LDS (k_pna_first.hip)
            if (whole && fi <= 12 && skip == nullptr && !fpx && p.size() >= 8 && options().pna_fold_lin && ws->prep_delta == d.pna_delta) {
                hipError_t he = launch_pna_first(ws->t, cur, fi, p[0], p[1], p[6], 13 * fi, p[7], nxt, fo, d.activation, (hipStream_t)stream);
                if (he == hipSuccess)
                    break;
                if (he != hipErrorNotSupported)
                    return fail(GNNB_ERR_HIP, "narrow PNA layer launch failed: %s", hipGetErrorString(he));
            }
            // h_ij = Wpre [x_i || x_j] + b  ==  (Wpre[:, :F] x_i + b) + Wpre[:, F:] x_j
            float *q = ws->tmp0, *pp = ws->tmp1;
            // degree-class form (gnnb_workspace_set_max_degree; decided here: it folds the destination's pre-NN term into x's
            // class weights, so q is not computed and the aggregate runs without a destination term)
            const bool classes = p.size() >= 10 && options().pna_fold_lin && options().pna_classes && whole && ws->deg_ready && M > 0 && !fpx &&
                                 ws->deg_delta == model->desc.pna_delta && fo > 32;
            if (!classes && (rc = linear1(R(cur, fi), fi, fi, p[0], 2 * fi, p[1], nullptr, Rw(q, fi), M, fi, GNNB_ACT_NONE, stream)))
                return rc;
            // the source half p = x . Wb^T and its aggregate: in one kernel, p on chip, where the degree-class form (no destination
            // term) and the max_graph_nodes promise (whole graphs in a stage) allow; else GEMM -> [N, F] -> aggregate
            bool pagg = false;
            if (classes) {
                hipError_t he = launch_pna_pagg(ws->t, cur, fi, p[0] + fi, 2 * fi, ws->agg, (hipStream_t)stream);
                if (he == hipSuccess)
                    pagg = true;
                else if (he != hipErrorNotSupported)
                    return fail(GNNB_ERR_HIP, "PNA product + aggregate launch failed: %s", hipGetErrorString(he));
            }
            if (!pagg) {
                if ((rc = linear1(R(cur, fi), fi, fi, p[0] + fi, 2 * fi, nullptr, nullptr, Rw(pp, fi), M, fi, GNNB_ACT_NONE, stream)))
                    return rc;
                if ((rc = aggregate(GNNB_AGG_PNA, pp, classes ? nullptr : q, ws->agg, fi, 0.f)))
                    return rc;
            }
            // [x | A | amp.A | att.A] . Wpost^T without materialising the 13F concat
            gnnb_gemm_seg segs[4] = {{R(cur, fi), nullptr, fi, fi},
                                     {R(ws->agg, 4 * fi), nullptr, 4 * fi, 4 * fi},
                                     {R(ws->agg, 4 * fi), ws->t.amp + row_lo, 4 * fi, 4 * fi},
                                     {R(ws->agg, 4 * fi), ws->t.att + row_lo, 4 * fi, 4 * fi}};
            if (classes) {
                // degree-class form (gnnb_workspace_set_max_degree): [x | A] . W_class^T over the class-sorted rows, written to
                // the rows' own places; skip + activation in the epilogue (the last layer pools in the pass behind)
                gnnb_gemm_seg s2[2] = {{cur, nullptr, fi, fi}, {ws->agg, nullptr, 4 * fi, 4 * fi}};
                GemmArgs g;
                if ((rc = build_gemm(g, s2, 2, p[8], 5 * fi)))
                    return rc;
                RowClasses rcl;
                rcl.perm = ws->deg_perm;
                rcl.tile_cls = ws->deg_tile_cls;
                rcl.w_stride = (long)fo * 5 * fi;
                rcl.bias_stride = fo;
                hipError_t he = launch_linear(g, p[8], 5 * fi, p[9], skip_fold ? nullptr : skip, nxt, ws->deg_max_tiles * 128, fo, d.activation,
                                              (hipStream_t)stream, nullptr, &rcl, sko);
                if (he == hipSuccess)
                    break;
                // (no way back from here: the aggregate above ran without the destination term)
                return fail(GNNB_ERR_HIP, "degree-class GEMM launch failed: %s", hipGetErrorString(he));
            }
            if (p.size() >= 8 && options().pna_fold_lin) {
                // `lin` folded into the post-NN at upload (gnnb_model_create): one GEMM, skip + activation in its epilogue;
                // the last layer of a whole-batch run pools there too (as GraphSAGE's)
                if (pooled_in_epilogue && whole && l == d.num_layers - 1 && !fpx && options().fuse_pool && ws->t.node_graph && ws->pool_part && skip == nullptr) {
                    GemmArgs g;
                    if ((rc = build_gemm(g, segs, 4, p[6], 13 * fi)))
                        return rc;
                    PoolEpilogue pe;
                    pe.node_graph = ws->t.node_graph;
                    pe.graph_ptr = ws->t.graph_ptr;
                    pe.pooled = ws->pooled;
                    pe.part = ws->pool_part;
                    pe.num_graphs = ws->t.num_graphs;
                    pe.np = d.num_pools;
                    for (int k = 0; k < 3; k++)
                        pe.pools[k] = k < d.num_pools ? d.pools[k] : 0;
                    hipError_t he = launch_linear(g, p[6], 13 * fi, p[7], nullptr, nxt, M, fo, d.activation, (hipStream_t)stream, &pe);
                    if (he == hipSuccess) {
                        GNNB_HIP_TRY(launch_pool_combine(pe, M, fo, (hipStream_t)stream));
                        *pooled_in_epilogue = true;
                        break;
                    }
                    if (he != hipErrorNotSupported)
                        return fail(GNNB_ERR_HIP, "pooling GEMM launch failed: %s", hipGetErrorString(he));
                }
                if ((rc = gnnb_linear(segs, 4, p[6], 13 * fi, p[7], skip_fold ? nullptr : R(skip, fo), Rw(nxt, fo), M, fo, d.activation, stream)))
                    return rc;
                break;
            }
            float *hid = ws->tmp0; // q is dead after the aggregate
            if ((rc = gnnb_linear(segs, 4, p[2], 13 * fi, p[3], nullptr, Rw(hid, fo), M, fo, GNNB_ACT_NONE, stream)))
                return rc;
            if ((rc = linear1(R(hid, fo), fo, fo, p[4], fo, p[5], R(skip, fo), Rw(nxt, fo), M, fo, d.activation, stream)))
                return rc;
            break;
        }
        }
        if ((rc = quant(Rw(nxt, fo), (size_t)M * fo)))
            return rc;
        cur = nxt;
        which ^= 1;
    }

    *out_cur = cur;
    return GNNB_OK;
}

// The LDS-resident conv stack + pooling for this model on the prepared batch -> ws->pooled.  hipErrorNotSupported when
// no stack kernel takes the model / batch (the caller runs layer by layer); *path says which kernel ran.
// The large segment through the small-footprint per-layer kernel (k_conv_rows) + pooling, all on stream `s`; fills
// ws->pooled rows [large_g, B).  hipErrorNotSupported (nothing launched) when a layer does not suit that kernel.
static hipError_t large_segment_small(const gnnb_model *model, gnnb_workspace *ws, const float *x_dev, hipStream_t s)
{
    const gnnb_model_desc &d = model->desc;
    if (d.conv_type != GNNB_CONV_GCN && d.conv_type != GNNB_CONV_GIN)
        return hipErrorNotSupported;
    for (int l = 0; l < d.num_layers; l++) {
        const LayerDims ld = layer_dims(d, l);
        if (ld.fin > 128 || ld.fout > 128)
            return hipErrorNotSupported;
    }
    if ((d.in_dim & 3) == 0 && (((uintptr_t)x_dev) & 15))
        return hipErrorNotSupported;
    const float *cur = x_dev;
    int which = 0;
    for (int l = 0; l < d.num_layers; l++) {
        const LayerDims ld = layer_dims(d, l);
        const std::vector<const float *> &p = model->conv[l];
        const float *skip = (d.skip && l != 0 && l != d.num_layers - 1) ? cur : nullptr;
        float *nxt = ws->act[which];
        if ((const float *)nxt == cur) {
            which ^= 1;
            nxt = ws->act[which];
        }
        const bool gin = d.conv_type == GNNB_CONV_GIN;
        hipError_t he = launch_conv_rows(ws->t, d.conv_type, cur, ld.fin, p[0], p[1], gin ? p[2] : nullptr, gin ? p[3] : nullptr,
                                         ld.fout, skip, nxt, ws->large_n, d.activation, d.gin_eps, s);
        if (he != hipSuccess)
            return he; // (NotSupported can only come from the first layer's checks above: nothing is half done)
        cur = nxt;
        which ^= 1;
    }
    const int gw = gnn_out_width(d), B = ws->t.num_graphs;
    return launch_global_pool(cur, ws->t.graph_ptr + ws->large_g, B - ws->large_g, gw, d.pools, d.num_pools,
                              ws->pooled + (size_t)ws->large_g * d.num_pools * gw, s);
}

// The batch tables restricted to the graphs the max_graph_nodes promise covers: everything, or -- with a large segment --
// graphs [0, large_g) = nodes [0, large_n) = edges [0, large_e).  The stack kernels clamp every table entry to these
// counts, so a tile that begins in the small segment ends at its last node.
static BatchTables small_segment(const gnnb_workspace *ws)
{
    BatchTables t = ws->t;
    if (ws->large_g >= 0 && ws->large_g < t.num_graphs) {
        t.num_graphs = ws->large_g;
        t.num_nodes = ws->large_n;
        t.num_edges = ws->large_e;
        t.num_tiles = (t.num_nodes + t.tile_rows - 1) / t.tile_rows;
    }
    return t;
}

// head_out != nullptr: the stack kernel may run the MLP head on the graphs of `t` as well (k_gcn2_zf does when the head's
// activation is the conv stack's and its shape suits: *head_fused); out rows [0, t.num_graphs) are then complete
static hipError_t launch_conv_stack(const gnnb_model *model, gnnb_workspace *ws, const BatchTables &t, const float *x_dev,
                                    const G2Deep &deep, hipStream_t s, int *path, float *head_out = nullptr, bool *head_fused = nullptr)
{
    const gnnb_model_desc &d = model->desc;
    const int L = d.num_layers;
    hipError_t he = hipErrorNotSupported;
    if (head_fused)
        *head_fused = false;
    if (!deep.gin && L == 2) { // two GCN layers, fp32: the transform-first form with 96-row stages (k_stack_zf.hip)
        const bool offer = head_out != nullptr && model->head_dev != nullptr && d.mlp_num_linear <= 8 && d.mlp_activation == d.activation;
        const HeadArgs head = model_head_args(model);
        he = launch_gcn2_zf(t, x_dev, d.in_dim, model->conv[0][0], model->conv[0][1], d.hidden_dim, model->conv[1][0],
                            model->conv[1][1], d.out_dim, d.activation, d.pools, d.num_pools, ws->pooled, s, model->zf_w1f,
                            offer ? &head : nullptr, offer ? model->head_dev : nullptr, offer ? head_out : nullptr, head_fused);
    }
    *path = GNNB_PATH_STACK_ZF;
    if (he == hipErrorNotSupported) {
        *path = GNNB_PATH_STACK;
        he = launch_gcn2_fused(t, x_dev, d.in_dim, model->conv[0][0], model->conv[0][1], d.hidden_dim,
                               model->conv[L - 1][0], model->conv[L - 1][1], d.out_dim, d.activation, d.pools,
                               d.num_pools, ws->pooled, s, deep);
    }
    return he;
}

static int forward_prepared_body(const gnnb_model *model, gnnb_workspace *ws, const float *x_dev, float *out_dev,
                                 void *stream);

// The side stream and its fork / join events, only for large_fork = 1 (the default, 2, never uses them): created on first
// use -- all three or none; a partial failure destroys what was created and the large segment stays on the caller's stream.
static bool ensure_side_stream(gnnb_workspace *ws)
{
    if (ws->side)
        return true;
    hipStream_t st = nullptr;
    hipEvent_t ef = nullptr, ej = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
        hipEventCreateWithFlags(&ef, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&ej, hipEventDisableTiming) == hipSuccess) {
        ws->side = st;
        ws->ev_fork = ef;
        ws->ev_join = ej;
        return true;
    }
    (void)hipGetLastError();
    if (ej)
        (void)hipEventDestroy(ej);
    if (ef)
        (void)hipEventDestroy(ef);
    if (st)
        (void)hipStreamDestroy(st);
    return false;
}

int gnnb_forward_prepared(const gnnb_model *model, gnnb_workspace *ws, const float *x_dev,
                          float *out_dev, void *stream)
{
    // every launch below runs in the MODEL's math mode; the reduced modes' kernels flag this workspace
    MathScope math_scope(model ? model->desc.math : -1, ws ? ws->t.err : nullptr, ws ? ws->t.err_host_dev : nullptr);
    int rc = forward_prepared_body(model, ws, x_dev, out_dev, stream);
    if (rc != GNNB_OK)
        return rc;
    // output_activation(dim=-1) over every graph's output row (models.py:572-573)
    if (model->desc.output_activation != GNNB_OUT_NONE)
        GNNB_HIP_TRY(launch_output_activation(out_dev, ws->t.num_graphs, model->desc.mlp_out, model->desc.output_activation,
                                              (hipStream_t)stream));
    return GNNB_OK;
}

static int forward_prepared_body(const gnnb_model *model, gnnb_workspace *ws, const float *x_dev, float *out_dev,
                                 void *stream)
{
    if (!model || !ws || !x_dev || !out_dev)
        return fail(GNNB_ERR_INVALID, "null argument to gnnb_forward");
    if (!ws->prepared)
        return fail(GNNB_ERR_INVALID, "workspace has no prepared batch");
    if (memcmp(&model->desc, &ws->desc, sizeof(gnnb_model_desc)) != 0)
        return fail(GNNB_ERR_INVALID, "workspace was created for a different model");
    const gnnb_model_desc &d = model->desc;
    const int N = ws->t.num_nodes, B = ws->t.num_graphs;
    int rc;
    const bool fpx = d.fpx_w > 0;
    auto quant = [&](float *buf, size_t n) -> int { // (fixed-point emulation only: put a finished tensor on the grid)
        if (!fpx)
            return GNNB_OK;
        GNNB_HIP_TRY(launch_quantize(buf, buf, n, d.fpx_w, d.fpx_i, (hipStream_t)stream));
        return GNNB_OK;
    };
    if (fpx) { // the input features enter as F_TYPE values: a quantised copy (the caller's buffer is not written)
        GNNB_HIP_TRY(launch_quantize(x_dev, ws->act[1], (size_t)N * d.in_dim, d.fpx_w, d.fpx_i, (hipStream_t)stream));
        x_dev = ws->act[1]; // (the layer loop never writes the buffer it reads)
    }

    // ---- fused path: the whole GCN stack (two or more layers) + pooling in one persistent kernel, then the MLP head
    const G2Deep deep = gcn_stack_middle_layers(model);
    // With a large segment (graphs the promise does not cover, ordered last by the caller) the stack runs on the graphs
    // in front of it and the large ones go layer by layer into the same pooled matrix: one oversized molecule no longer
    // sends the whole batch down the layer-by-layer path.  (large_g = 0: every graph is large -> layer by layer below.)
    const bool seg = ws->large_g >= 0 && ws->large_g < B;
    if (!fpx && deep.nl >= 2 && d.mlp_num_linear <= 8 && !(seg && ws->large_g == 0)) {
        // The large segment first, FORKED: its kernels are built to run beside the stack kernel (k_conv_rows.hip), so they
        // go on the workspace's side stream behind an event on the caller's stream and are joined in front of the readout.
        // (Capturable: the side stream joins a capture through the event and is joined back.)
        bool forked = false, side_forked = false;
        if (seg && options().large_fork == 1 && ensure_side_stream(ws)) {
            GNNB_HIP_TRY(hipEventRecord(ws->ev_fork, (hipStream_t)stream));
            GNNB_HIP_TRY(hipStreamWaitEvent(ws->side, ws->ev_fork, 0));
            side_forked = true;
            hipError_t hl = large_segment_small(model, ws, x_dev, ws->side);
            // (joined whether or not anything ran on the side stream -- also in front of the error return: a side stream
            // left forked would invalidate a capture in progress)
            const hipError_t hj = hipEventRecord(ws->ev_join, ws->side);
            if (hj != hipSuccess || (hl != hipSuccess && hl != hipErrorNotSupported)) {
                if (hj == hipSuccess)
                    (void)hipStreamWaitEvent((hipStream_t)stream, ws->ev_join, 0);
                return fail(GNNB_ERR_HIP, "large-segment launch failed: %s", hipGetErrorString(hl != hipSuccess ? hl : hj));
            }
            forked = hl == hipSuccess;
        }
        bool head_fused = false;
        hipError_t he = launch_conv_stack(model, ws, small_segment(ws), x_dev, deep, (hipStream_t)stream, &ws->last_path, out_dev, &head_fused);
        if (side_forked)
            GNNB_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, ws->ev_join, 0));
        if (he == hipSuccess) {
            if (seg && !forked && options().large_fork == 2) { // the small kernels on the caller's stream, behind the stack
                hipError_t hl = large_segment_small(model, ws, x_dev, (hipStream_t)stream);
                if (hl != hipSuccess && hl != hipErrorNotSupported)
                    return fail(GNNB_ERR_HIP, "large-segment launch failed: %s", hipGetErrorString(hl));
                forked = hl == hipSuccess;
            }
            if (seg && forked) {
                ws->last_path |= GNNB_PATH_LARGE_LAYERWISE;
            } else if (seg) {
                const float *lcur = nullptr;
                if ((rc = run_conv_layers(model, ws, x_dev, ws->large_n, ws->large_n / ws->t.tile_rows, &lcur, stream)))
                    return rc;
                const int gwl = gnn_out_width(d);
                GNNB_HIP_TRY(launch_global_pool(lcur, ws->t.graph_ptr + ws->large_g, B - ws->large_g, gwl, d.pools, d.num_pools,
                                                ws->pooled + (size_t)ws->large_g * d.num_pools * gwl, (hipStream_t)stream));
                ws->last_path |= GNNB_PATH_LARGE_LAYERWISE;
            }
            const HeadArgs head = model_head_args(model);
            // (the stack kernel ran the head on its own graphs: what is left are the graphs of the large segment, if any)
            const int hg0 = head_fused ? (seg ? ws->large_g : B) : 0;
            if (hg0 >= B)
                return GNNB_OK;
            const size_t pw = (size_t)d.num_pools * d.out_dim;
            he = launch_pool_mlp(nullptr, ws->t.graph_ptr + hg0, B - hg0, d.out_dim, d.pools, d.num_pools, head, d.mlp_activation,
                                 out_dev + (size_t)hg0 * d.mlp_out, (hipStream_t)stream, ws->pooled + (size_t)hg0 * pw);
            if (he == hipSuccess)
                return GNNB_OK;
            if (he != hipErrorNotSupported)
                return fail(GNNB_ERR_HIP, "readout launch failed: %s", hipGetErrorString(he));
            // head too large for the fused readout: plain GEMM chain on the pooled matrix
            const float *h = ws->pooled + (size_t)hg0 * pw;
            for (int i = 0; i < d.mlp_num_linear; i++) {
                int din, dout;
                mlp_dims(d, i, &din, &dout);
                const bool last = (i == d.mlp_num_linear - 1);
                float *y = last ? out_dev + (size_t)hg0 * d.mlp_out : ws->mlp[i & 1];
                if ((rc = linear1(h, din, din, model->head_w[i], din, model->head_b[i], nullptr, y, B - hg0, dout,
                                  last ? GNNB_ACT_NONE : d.mlp_activation, stream)))
                    return rc;
                h = y;
            }
            return GNNB_OK;
        }
        if (he != hipErrorNotSupported)
            return fail(GNNB_ERR_HIP, "fused GCN stack launch failed: %s", hipGetErrorString(he));
    }

    ws->last_path = GNNB_PATH_LAYERWISE;
    const float *cur = nullptr;
    bool pooled_done = false;
    if ((rc = run_conv_layers(model, ws, x_dev, 0, 0, &cur, stream, &pooled_done)))
        return rc;

    const int gw = gnn_out_width(d);
    if (pooled_done) {
        // (the last layer's GEMM pooled in its epilogue: ws->pooled is complete, the readout takes it as the stack path does)
        HeadArgs head;
        memset(&head, 0, sizeof(head));
        head.nlin = d.mlp_num_linear;
        if (head.nlin <= 8) {
            for (int i = 0; i < head.nlin; i++) {
                int din, dout;
                mlp_dims(d, i, &din, &dout);
                head.w[i] = model->head_w[i];
                head.b[i] = model->head_b[i];
                head.dims[i] = din;
                head.dims[i + 1] = dout;
            }
            hipError_t he = launch_pool_mlp(nullptr, ws->t.graph_ptr, B, gw, d.pools, d.num_pools, head, d.mlp_activation, out_dev,
                                            (hipStream_t)stream, ws->pooled);
            if (he == hipSuccess)
                return GNNB_OK;
            if (he != hipErrorNotSupported)
                return fail(GNNB_ERR_HIP, "readout launch failed: %s", hipGetErrorString(he));
        }
    } else if (!fpx) {
        // fused readout (pooling + whole MLP head, one launch) when the head fits LDS
        HeadArgs head;
        memset(&head, 0, sizeof(head));
        head.nlin = d.mlp_num_linear;
        if (head.nlin <= 8) {
            for (int i = 0; i < head.nlin; i++) {
                int din, dout;
                mlp_dims(d, i, &din, &dout);
                head.w[i] = model->head_w[i];
                head.b[i] = model->head_b[i];
                head.dims[i] = din;
                head.dims[i + 1] = dout;
            }
            hipError_t he = hipErrorNotSupported;
            if (options().head_split) {
                // pooling pass (HBM-bound, every CU) + the small readout on the pooled matrix: neither needs the
                // 119 KB of LDS of the one-launch form, so both share CUs with other batches' kernels
                if ((rc = gnnb_global_pool(ws, cur, gw, d.pools, d.num_pools, ws->pooled, stream)))
                    return rc;
                he = launch_pool_mlp(nullptr, ws->t.graph_ptr, B, gw, d.pools, d.num_pools, head, d.mlp_activation, out_dev,
                                     (hipStream_t)stream, ws->pooled);
            } else {
                he = launch_pool_mlp(cur, ws->t.graph_ptr, B, gw, d.pools, d.num_pools, head, d.mlp_activation, out_dev,
                                     (hipStream_t)stream);
            }
            if (he == hipSuccess)
                return GNNB_OK;
            if (he != hipErrorNotSupported)
                return fail(GNNB_ERR_HIP, "fused readout launch failed: %s", hipGetErrorString(he));
            // The head's weights do not fit LDS (SAGE d = 256 with three pools: 768 x 64 floats): pooling pass, then
            // the small readout that takes its weights from L2 as MFMA operands -- one launch over B / 16 workgroups
            // instead of a chain of GEMMs with M = B rows (64 workgroups of the 128-row tile at B = 8192: 51 us)
            if (!options().head_split && options().head_small) {
                if ((rc = gnnb_global_pool(ws, cur, gw, d.pools, d.num_pools, ws->pooled, stream)))
                    return rc;
                he = launch_pool_mlp(nullptr, ws->t.graph_ptr, B, gw, d.pools, d.num_pools, head, d.mlp_activation, out_dev,
                                     (hipStream_t)stream, ws->pooled);
                if (he == hipSuccess)
                    return GNNB_OK;
                if (he != hipErrorNotSupported)
                    return fail(GNNB_ERR_HIP, "readout launch failed: %s", hipGetErrorString(he));
                pooled_done = true;
            }
        }
    }
    if (!pooled_done && (rc = gnnb_global_pool(ws, cur, gw, d.pools, d.num_pools, ws->pooled, stream)))
        return rc;
    if ((rc = quant(ws->pooled, (size_t)B * d.num_pools * gw)))
        return rc;

    const float *h = ws->pooled;
    for (int i = 0; i < d.mlp_num_linear; i++) {
        int din, dout;
        mlp_dims(d, i, &din, &dout);
        const bool last = (i == d.mlp_num_linear - 1);
        float *y = last ? out_dev : ws->mlp[i & 1];
        if ((rc = linear1(h, din, din, model->head_w[i], din, model->head_b[i], nullptr, y, B, dout,
                          last ? GNNB_ACT_NONE : d.mlp_activation, stream)))
            return rc;
        if ((rc = quant(y, (size_t)B * dout)))
            return rc;
        h = y;
    }
    return GNNB_OK;
}

int gnnb_forward_batched(const gnnb_model *model, gnnb_workspace *ws, const float *x_dev,
                         const int32_t *coo_dev, const int32_t *node_ptr_dev,
                         const int32_t *edge_ptr_dev, int num_graphs, int num_nodes, int num_edges,
                         float *out_dev, void *stream)
{
    if (!model || !ws)
        return fail(GNNB_ERR_INVALID, "null argument to gnnb_forward_batched");
    int rc = gnnb_graph_prep(ws, coo_dev, node_ptr_dev, edge_ptr_dev, num_graphs, num_nodes,
                             num_edges, model->desc.pna_delta, stream);
    if (rc != GNNB_OK)
        return rc;
    return gnnb_forward_prepared(model, ws, x_dev, out_dev, stream);
}

// gnnb_forward_prepared(model, ws, ...) followed by gnnb_graph_prep(ws_next, ...) on the same stream -- with the prep of ws_next
// run INSIDE the forward's readout kernel where that exists (k_head_small, GUEST: extra workgroups): the software-pipelined form
// of gnnb_forward_batched for a stream of batches over two alternating workspaces.
int gnnb_forward_prepared_prep_next(const gnnb_model *model, gnnb_workspace *ws, const float *x_dev, float *out_dev,
                                    gnnb_workspace *ws_next, const int32_t *coo_dev, const int32_t *node_ptr_dev,
                                    const int32_t *edge_ptr_dev, int num_graphs, int num_nodes, int num_edges, void *stream)
{
    if (!model || !ws || !ws_next)
        return fail(GNNB_ERR_INVALID, "null argument to gnnb_forward_prepared_prep_next");
    if (ws == ws_next)
        return fail(GNNB_ERR_INVALID, "gnnb_forward_prepared_prep_next: the next batch needs a workspace of its own (the forward reads "
                                      "the tables the prep writes)");
    if (!ws->prepared)
        return fail(GNNB_ERR_INVALID, "workspace has no prepared batch");
    if (!guest_prep_eligible(ws_next, num_nodes)) {
        const int rc = gnnb_forward_prepared(model, ws, x_dev, out_dev, stream);
        if (rc != GNNB_OK)
            return rc;
        return gnnb_graph_prep(ws_next, coo_dev, node_ptr_dev, edge_ptr_dev, num_graphs, num_nodes, num_edges, model->desc.pna_delta, stream);
    }
    PrepParams pp;
    int rc = graph_prep_impl(ws_next, coo_dev, node_ptr_dev, edge_ptr_dev, num_graphs, num_nodes, num_edges, model->desc.pna_delta, stream, &pp);
    if (rc != GNNB_OK)
        return rc; // (nothing was enqueued)
    ws_next->prepared = false; // (until its prep is enqueued)
    GuestPrep offer{&pp, false};
    struct Offer { // (the slot never outlives this call)
        GuestPrep *prev;
        explicit Offer(GuestPrep *g) : prev(guest_prep_slot()) { guest_prep_slot() = g; }
        ~Offer() { guest_prep_slot() = prev; }
    };
    {
        Offer scope(&offer);
        rc = gnnb_forward_prepared(model, ws, x_dev, out_dev, stream);
    }
    if (rc != GNNB_OK && !offer.taken)
        return rc; // (ws_next stays unprepared)
    if (!offer.taken) // the forward ran another readout than the one that hosts a prep: the prep as a launch of its own
        GNNB_HIP_TRY(launch_graph_prep(pp, (hipStream_t)stream));
    ws_next->prepared = true;
    return rc;
}

int gnnb_forward_batched_host(const gnnb_model *model, gnnb_workspace *ws, const float *x,
                              const int32_t *coo, const int32_t *node_ptr, const int32_t *edge_ptr,
                              int num_graphs, int num_nodes, int num_edges, float *out)
{
    if (!model || !ws || !x || !node_ptr || !edge_ptr || !out || (num_edges > 0 && !coo))
        return fail(GNNB_ERR_INVALID, "null argument to gnnb_forward_batched_host");
    if (num_graphs > ws->max_graphs || num_nodes > ws->max_nodes || num_edges > ws->max_edges)
        return fail(GNNB_ERR_CAPACITY,
                    "batch (%d graphs, %d nodes, %d edges) exceeds workspace (%d, %d, %d)",
                    num_graphs, num_nodes, num_edges, ws->max_graphs, ws->max_nodes, ws->max_edges);
    const gnnb_model_desc &d = model->desc;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // staging buffers live in the workspace, sized once for its capacities: the reference's <name>_top is called once
    // per graph (model_tb.cpp.jinja:189-205), and a hipMalloc / hipFree pair per call would dominate it
    const size_t cx = up((size_t)ws->max_nodes * d.in_dim * 4), cc = up((size_t)std::max(ws->max_edges, 1) * 8),
                 cp = up(((size_t)ws->max_graphs + 1) * 4), co = up((size_t)ws->max_graphs * d.mlp_out * 4);
    if (!ws->stage) {
        GNNB_HIP_TRY(hipMalloc((void **)&ws->stage, cx + cc + 2 * cp + co));
        ws->stage_bytes = cx + cc + 2 * cp + co;
    }
    float *dx = (float *)ws->stage;
    int32_t *dc = (int32_t *)(ws->stage + cx);
    int32_t *dn = (int32_t *)(ws->stage + cx + cc);
    int32_t *de = (int32_t *)(ws->stage + cx + cc + cp);
    float *dout = (float *)(ws->stage + cx + cc + 2 * cp);
    const size_t bx = (size_t)num_nodes * d.in_dim * 4, bc = (size_t)num_edges * 8,
                 bp = ((size_t)num_graphs + 1) * 4, bo = (size_t)num_graphs * d.mlp_out * 4;
    hipStream_t s0 = nullptr;
    if (bx)
        GNNB_HIP_TRY(hipMemcpyAsync(dx, x, bx, hipMemcpyHostToDevice, s0));
    if (bc)
        GNNB_HIP_TRY(hipMemcpyAsync(dc, coo, bc, hipMemcpyHostToDevice, s0));
    GNNB_HIP_TRY(hipMemcpyAsync(dn, node_ptr, bp, hipMemcpyHostToDevice, s0));
    GNNB_HIP_TRY(hipMemcpyAsync(de, edge_ptr, bp, hipMemcpyHostToDevice, s0));
    int rc = gnnb_forward_batched(model, ws, dx, dc, dn, de, num_graphs, num_nodes, num_edges, dout, nullptr);
    if (rc == GNNB_OK && bo)
        GNNB_HIP_TRY(hipMemcpyAsync(out, dout, bo, hipMemcpyDeviceToHost, s0));
    if (rc == GNNB_OK)
        rc = gnnb_workspace_check(ws, nullptr); // one synchronisation: the validation word and `out` are both back
    else
        (void)hipStreamSynchronize(s0);
    return rc;
}

// ---------------------------------------------------------------------------------------
int gnnb_aggregate_timed(gnnb_workspace *ws, int agg_kind, const float *const *x_dev_list,
                         const float *self_dev, float *const *out_dev_list, int nbuf, int width,
                         float eps, int iters, void *stream, float *out_us_per_launch)
{
    if (!x_dev_list || !out_dev_list || nbuf < 1 || iters < 1 || !out_us_per_launch)
        return fail(GNNB_ERR_INVALID, "bad argument to gnnb_aggregate_timed");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    GNNB_HIP_TRY(hipEventCreate(&e0));
    GNNB_HIP_TRY(hipEventCreate(&e1));
    int rc = GNNB_OK;
    for (int i = 0; i < nbuf && rc == GNNB_OK; i++) // warm-up, also touches every buffer
        rc = gnnb_aggregate(ws, agg_kind, x_dev_list[i], self_dev, out_dev_list[i], width, eps, stream);
    if (rc == GNNB_OK) {
        GNNB_HIP_TRY(hipStreamSynchronize(s));
        GNNB_HIP_TRY(hipEventRecord(e0, s));
        for (int i = 0; i < iters && rc == GNNB_OK; i++)
            rc = gnnb_aggregate(ws, agg_kind, x_dev_list[i % nbuf], self_dev, out_dev_list[i % nbuf], width,
                                eps, stream);
        GNNB_HIP_TRY(hipEventRecord(e1, s));
        GNNB_HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        GNNB_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *out_us_per_launch = ms * 1000.0f / (float)iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int gnnb_gcn_stack_timed(const gnnb_model *model, gnnb_workspace *ws, const float *x_dev, int iters,
                         void *stream, float *out_us_per_launch)
{
    if (!model || !ws || !x_dev || iters < 1 || !out_us_per_launch)
        return fail(GNNB_ERR_INVALID, "bad argument to gnnb_gcn_stack_timed");
    MathScope math_scope(model->desc.math, ws->t.err, ws->t.err_host_dev);
    if (!ws->prepared)
        return fail(GNNB_ERR_INVALID, "workspace has no prepared batch");
    const G2Deep deep = gcn_stack_middle_layers(model);
    if (deep.nl < 2)
        return fail(GNNB_ERR_INVALID, "the fused stack exists for GCN / GIN models of two or more layers");
    hipStream_t s = (hipStream_t)stream;
    // (as the forward launches it: with the MLP head inside where k_gcn2_zf takes it; its output goes to a workspace buffer)
    bool head_fused = false;
    auto launch = [&]() { return launch_conv_stack(model, ws, small_segment(ws), x_dev, deep, s, &ws->last_path, ws->mlp[0], &head_fused); };
    hipEvent_t e0, e1;
    GNNB_HIP_TRY(hipEventCreate(&e0));
    GNNB_HIP_TRY(hipEventCreate(&e1));
    int rc = GNNB_OK;
    hipError_t he = hipSuccess;
    for (int i = 0; i < 3 && he == hipSuccess; i++)
        he = launch();
    if (he == hipErrorNotSupported)
        rc = fail(GNNB_ERR_INVALID, "fused stack not eligible (shape, or no max_graph_nodes promise)");
    else if (he != hipSuccess)
        rc = fail(GNNB_ERR_HIP, "fused GCN stack launch failed: %s", hipGetErrorString(he));
    if (rc == GNNB_OK) {
        GNNB_HIP_TRY(hipStreamSynchronize(s));
        GNNB_HIP_TRY(hipEventRecord(e0, s));
        for (int i = 0; i < iters; i++)
            (void)launch();
        GNNB_HIP_TRY(hipEventRecord(e1, s));
        GNNB_HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        GNNB_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *out_us_per_launch = ms * 1000.0f / (float)iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int gnnb_linear_timed(const float *a_dev, int lda, int k, const float *w_dev, int ldw,
                      const float *bias_dev, float *y_dev, int M, int N, int act, int iters,
                      void *stream, float *out_us_per_launch)
{
    if (iters < 1 || !out_us_per_launch)
        return fail(GNNB_ERR_INVALID, "bad argument to gnnb_linear_timed");
    hipStream_t s = (hipStream_t)stream;
    gnnb_gemm_seg seg = {a_dev, nullptr, lda, k};
    hipEvent_t e0, e1;
    GNNB_HIP_TRY(hipEventCreate(&e0));
    GNNB_HIP_TRY(hipEventCreate(&e1));
    int rc = GNNB_OK;
    for (int i = 0; i < 3 && rc == GNNB_OK; i++)
        rc = gnnb_linear(&seg, 1, w_dev, ldw, bias_dev, nullptr, y_dev, M, N, act, stream);
    if (rc == GNNB_OK) {
        GNNB_HIP_TRY(hipStreamSynchronize(s));
        GNNB_HIP_TRY(hipEventRecord(e0, s));
        for (int i = 0; i < iters && rc == GNNB_OK; i++)
            rc = gnnb_linear(&seg, 1, w_dev, ldw, bias_dev, nullptr, y_dev, M, N, act, stream);
        GNNB_HIP_TRY(hipEventRecord(e1, s));
        GNNB_HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        GNNB_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *out_us_per_launch = ms * 1000.0f / (float)iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// ---------------------------------------------------------------------------------------
int gnnb_event_create(void **out_event)
{
    if (!out_event)
        return fail(GNNB_ERR_INVALID, "null out_event");
    hipEvent_t ev;
    GNNB_HIP_TRY(hipEventCreate(&ev));
    *out_event = (void *)ev;
    return GNNB_OK;
}

int gnnb_event_record(void *event, void *stream)
{
    GNNB_HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return GNNB_OK;
}

int gnnb_event_elapsed_ms(void *start, void *stop, float *out_ms)
{
    if (!out_ms)
        return fail(GNNB_ERR_INVALID, "null out_ms");
    GNNB_HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    GNNB_HIP_TRY(hipEventElapsedTime(out_ms, (hipEvent_t)start, (hipEvent_t)stop));
    return GNNB_OK;
}

void gnnb_event_destroy(void *event)
{
    if (event)
        (void)hipEventDestroy((hipEvent_t)event);
}

int gnnb_malloc(void **out_dev, size_t bytes)
{
    if (!out_dev)
        return fail(GNNB_ERR_INVALID, "null out_dev");
    GNNB_HIP_TRY(hipMalloc(out_dev, bytes ? bytes : 4));
    return GNNB_OK;
}

void gnnb_free(void *dev)
{
    if (dev)
        (void)hipFree(dev);
}

int gnnb_memcpy_h2d(void *dst_dev, const void *src, size_t bytes, void *stream)
{
    GNNB_HIP_TRY(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return GNNB_OK;
}

int gnnb_memcpy_d2h(void *dst, const void *src_dev, size_t bytes, void *stream)
{
    GNNB_HIP_TRY(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    GNNB_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return GNNB_OK;
}

} // extern "C"
