/*
 * equidock_hip.h -- C ABI of libequidock_hip.so: the MI355X (gfx950) implementation of EquiDock's
 * IEGMN forward/backward hot path.
 *
 * The reference (octavian-ganea/equidock_public) is pure Python: it has NO FFI/plugin boundary
 * of its own.  Its "operator surface" for this path is three nn.Modules
 * (src/model/rigid_docking_model.py: IEGMN_Layer :82, IEGMN :360, Rigid_Body_Docking_Net :611)
 * whose arithmetic runs in PyTorch + DGL kernels.  This header is the boundary a maintainer
 * binds with ctypes; each entry point cites the reference lines whose
 * arithmetic it replaces.  The Python drop-in modules in equidock_public_amd/model.py call
 * exactly these functions.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless stated otherwise; fp32 is `float`,
 *     indices are int32; row-major, leading dimension given where it is not the width;
 *   - ownership: the caller owns every buffer including workspaces; the library allocates
 *     nothing persistent and keeps no global mutable state except a thread-local error string (and the
 *     optional launch profiler below);
 *   - every function enqueues its work on `stream` (a hipStream_t passed as void*) and returns
 *     without synchronising; return value 0 = EQD_OK, otherwise an EQD_ERR_* code, and
 *     eqd_last_error() describes it.  Nothing throws or exits across this boundary (the
 *     reference sys.exit()s on an unstable SVD, rigid_docking_model.py:582-584; here the
 *     per-pair status word reports it);
 *   - node order: all ligand nodes of all pairs, then all receptor nodes; edges are
 *     destination-sorted (ll edges then rr edges), endpoints are global node ids.
 */
#ifndef EQUIDOCK_HIP_H
#define EQUIDOCK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EQD_ABI_VERSION 9
#define EQD_TILE_EDGES 32   /* edges per node-aligned tile (max supported in-degree) */
#define EQD_ATT_BLOCK 32    /* nodes per cross-attention work item */
#define EQD_MAX_SRC 6
#define EQD_PARAMS_PER_LAYER 19
#define EQD_GLOBAL_PARAMS 5

enum {
    EQD_OK = 0,
    EQD_ERR_NULL = 1,         /* a required pointer is NULL */
    EQD_ERR_SHAPE = 2,        /* inconsistent sizes */
    EQD_ERR_UNSUPPORTED = 3,  /* configuration outside the HIP path (see eqd_model_check) */
    EQD_ERR_WORKSPACE = 4,    /* workspace too small */
    EQD_ERR_LAUNCH = 5        /* a kernel launch failed */
};

int eqd_abi_version(void);
const char* eqd_last_error(void);
int eqd_tile_edges(void);
/* 1 when the library is the x86 host simulator built by tests/hostsim (never shipped), else 0 */
int eqd_is_simulator(void);
/* The EQD_* environment switches that select kernel forms for tests and A/B measurements (EQD_FUSE_FWD, EQD_FUSE_GATHER,
 * EQD_ATT_SPLIT, EQD_ATT_BWD_SPLIT, EQD_ATT_LB, EQD_ATT_LB_NB, EQD_ATT_DS, EQD_ATT_QDS_NB, EQD_ROWWAVE, EQD_ROW_TILES,
 * EQD_ROWRES_TPS, EQD_ROWCHAIN_OCC, EQD_ATB_WGS, EQD_ATB_XCD_ALIGN, EQD_KEYPOINT_MM, EQD_KEYPOINT_NC) are read ONCE per process, at their first use, so that the forward and the backward of a step always
 * agree on the forms they run.  A caller that changes one of them afterwards calls this to make the library forget its
 * snapshot (nothing in the reference corresponds to it). */
void eqd_tunables_reload(void);
/* Test aid: one 256-thread workgroup runs the library's cross-lane helpers (DPP moves, v_permlane{16,32}_swap) on in256
 * [256] beside the plain ds_bpermute forms, and the guard-free exponentials of the softmax kernels beside expf / exp2f;
 * mismatch [4] (device ints, zeroed by the caller) receives the number of differing (lane, check) pairs of the exchanges [0],
 * of exp_nooverflow vs expf [1] (bit for bit, except 2^-149 for 0 in (-103.98, -103.28)), of exp2_flush vs exp2f on normal results [2] and at the -1e30 sentinel [3];
 * out256 [256] a value that depends on every exchange. */
int eqd_selftest_lane_exchanges(const float* in256, int* mismatch, float* out256, void* stream);

/* ---- per-launch timing (measurement aid for bench.py; nothing in the reference corresponds to it) ----
 * Between eqd_profile_begin(stream, max) and eqd_profile_end() every kernel this library launches (from any thread:
 * torch runs the backward on its autograd thread) is followed by a hipEventRecord on `stream` (pass the stream the model runs on; hipMemsetAsync fills are
 * not recorded and count towards the next launch).  eqd_profile_end() synchronises the last event and returns the
 * number of launches seen (or -1); eqd_profile_name(i) / eqd_profile_us(i) give launch i's kernel name and the time
 * between the events before and after it - the kernel's duration when the stream was kept busy (enqueue the work
 * behind a long-running kernel so that the host is ahead of the GPU), plus one event-record of overhead that the
 * caller calibrates with an empty interval.  eqd_profile_mark(label) records an event under `label` for work that was
 * enqueued by someone else since the last event (e.g. the caller's loss kernels).  Not usable while the stream is being captured into a hipGraph. */
int eqd_profile_begin(void* stream, int max_launches);
int eqd_profile_end(void);
int eqd_profile_mark(const char* label);   /* closes an interval for work other libraries enqueued on the stream */
const char* eqd_profile_name(int i);
float eqd_profile_us(int i);

/* ---- batched pair graph: kernel-side view of what the reference passes as a batched DGL
 *      heterograph (src/utils/train_utils.py:61-100). Built by equidock_public_amd/graph.py. ---- */
typedef struct EqdGraph {
    int32_t n_pairs, n_lig, n_rec, n_nodes, n_edges, n_tiles, n_att_items, max_seg;
    /* max_seg = the node count of the LONGEST protein of the batch (max over seg_off differences).  Not only a scheduling
     * hint: the dS hand-off form of the attention backward (eqd_cross_attention_bwd_ds, and eqd_model_backward for large
     * batches) uses it as the row stride of its workspace and writes row[key - first node of the key's protein] without a
     * bound check - a smaller value corrupts the workspace.  The library can only reject values impossible for the node
     * counts (max_seg * n_pairs < max(n_lig, n_rec), or max_seg > max(n_lig, n_rec)); the packers (graph.py,
     * eqd_host_pack.cpp) compute it from the same counts as seg_off.  That form also assumes att_items[i].other_begin /
     * other_end are exactly the partner protein's segment bounds (what the packers emit). */
    const int32_t* seg_off;    /* [2*n_pairs+1] global node offsets: ligand segments then receptor segments */
    const int32_t* src;        /* [n_edges] */
    const int32_t* dst;        /* [n_edges] sorted ascending */
    const int32_t* rowptr;     /* [n_nodes+1] CSR by destination */
    const int32_t* csc_ptr;    /* [n_nodes+1] edges grouped by source ... */
    const int32_t* csc_eid;    /* [n_edges]   ... as edge ids */
    const int32_t* tile_node;  /* [n_tiles+1] node ranges; each tile has <= EQD_TILE_EDGES in-edges */
    const int32_t* att_items;  /* [n_att_items][4] = {blk_begin, blk_end, other_begin, other_end}; blk_begin == blk_end
                                * marks an empty (padding) item.  Workgroup b of a launch runs on XCD b % 8, so the packers
                                * place the items that stream the same partner rows at indices of one residue mod 8 and
                                * pad n_att_items to a multiple of 8 (graph.py: _xcd_interleave); any order is CORRECT,
                                * a count that is not a multiple of 8 merely disables the half-block kernels */
    const int32_t* res_id;     /* [n_nodes] residue type 0..20 */
    const float* mu_r_norm;    /* [n_nodes][5] */
    const float* he;           /* [n_edges][27] */
    const float* x0;           /* [n_nodes][3] ligand new_x rows then receptor x rows */
    const uint16_t* he_bf16;   /* [n_edges][32] bf16 copy of he (27 used, 5 zeros; 64-byte rows) - only read in
                                  bf16 mode (EqdEdgeParams.bf16 / EqdModelDesc.storage_bf16), may be NULL otherwise */
} EqdGraph;

/* ---- model configuration: the `args` keys the reference's modules consume
 *      (rigid_docking_model.py:95-110, 366-379, 617-623) ---- */
typedef struct EqdModelDesc {
    int32_t n_layers;               /* iegmn_n_lays */
    int32_t d_emb;                  /* residue_emb_dim (64) */
    int32_t d_hid;                  /* iegmn_lay_hid_dim (64) */
    int32_t use_mean_node_features; /* layer-0 width d0 = d_emb + 5 */
    int32_t edge_feats;             /* input_edge_feats_dim (27) */
    int32_t n_heads;                /* num_att_heads (50) */
    int32_t cross_msgs, use_dist_in_layers, use_edge_features;
    float skip_weight_h, x_connection_init, lrelu_slope, ln_eps;
    int32_t svd_seed;               /* seed of the counter-based draws used if the SVD guard fires */
    int32_t storage_bf16;           /* 1: every GEMM of the path on the bf16 MFMA (inputs rounded to bf16, fp32 accumulate):
                                       edge kernels (EqdEdgeParams.bf16), node-level Linears and their weight gradients
                                       (EqdLinJob.bf16, EqdAtbJob.bf16), attention (eqd_cross_attention_*_bf16); default 0 */
} EqdModelDesc;

/* nn.Dropout in training mode (args['dropout'] > 0; src/utils/args.py:240 draws 0 or 0.25 for the published family): the
 * keep masks of ONE forward, drawn by the caller with torch's generator in the reference's consumption order (per layer:
 * edge_mlp.1 on the ll then the rr edges, coors_mlp.1 ll / rr, node_mlp.1 ligand / receptor; then mlp_h_mean_ROT.1 per pair,
 * receptor before ligand - rigid_docking_model.py:236-237, 263-265, 319-337, 524-529), re-ordered to the library's node /
 * edge order.  The same struct must be passed to the backward of that forward.  NULL = no dropout (eval mode, p = 0). */
typedef struct EqdDropout {
    float p;                   /* kept elements are scaled by 1 / (1 - p) */
    const uint32_t* edge_z1;   /* [n_layers][n_edges][2] bit f of the 64-bit pair = feature f of edge_mlp.1 is kept */
    const uint32_t* edge_ch;   /* [n_layers][n_edges][2] the same for coors_mlp.1 */
    const float* node;         /* layer 0: [n_nodes][d0], then layers 1..: [n_nodes][64] each; entries 0 or 1 / (1 - p) */
    const float* head;         /* [n_nodes][64] mlp_h_mean_ROT.1; entries 0 or 1 / (1 - p) */
} EqdDropout;

/* Packs the edge-level keep masks of one layer: factors_ll [e_ll][64], factors_rr [n_edges - e_ll][64] are what
 * nn.Dropout multiplies the ligand-edge / receptor-edge activations by (0 or 1 / (1 - p); torch's dropout applied to ones in
 * the reference's order, ligand edges first: rigid_docking_model.py:236-237, 263-265), in the RAW edge order of the two
 * graphs; perm [n_edges] (device, int64 - a torch index tensor; NULL = identity): library edge i = raw edge perm[i], raw edges numbered ligand graph first.
 * words [n_edges][2]: one layer of EqdDropout.edge_z1 / edge_ch.  Factor tensors 16-byte aligned. */
/* Library-drawn masks - the alternative to masks drawn by the caller with torch's generator (above): fills the four arrays
 * of an EqdDropout for model `m` on graph `g` in ONE launch, counter-based (Philox4x32-10 keyed by *seed, a DEVICE word the
 * caller draws from its own generator: the step stays capturable in a hipGraph and every replay sees fresh masks).  A Philox
 * call yields eight 16-bit draws (word 0 low half, word 0 high half, word 1 low half, ...); an element is kept when its draw
 * >= round(p 2^16).  Float arrays (a = 2 node, 3 head; kept entries = 1 / (1 - p)): element i takes draw (i mod 8) of
 * philox(counter = (i / 8 [lo, hi], 0, a), key = *seed); bit b of packed word w of the edge arrays (a = 0 edge_z1, 1 edge_ch)
 * takes draw (b mod 8) of philox(counter = (w [lo, hi], b / 8, a)).  The Bernoulli(1 - p) law of nn.Dropout (p quantised to
 * 2^-16), NOT torch's random stream: parity with the reference is statistical for these masks and exact for any masks passed
 * in (tests: the masks drawn here, handed to the torch restatement).  No [E][64] tensor is materialised. */
int eqd_dropout_draw(const EqdModelDesc* m, const EqdGraph* g, float p, const uint64_t* seed, uint32_t* edge_z1,
                     uint32_t* edge_ch, float* node, float* head, void* stream);

int eqd_dropout_pack_edges(int n_edges, int e_ll, const float* factors_ll, const float* factors_rr,
                           const int64_t* perm, uint32_t* words, void* stream);

/* Parameter table: device pointers in this fixed order.  Per layer i (base = 19*i):
 *   0 edge_mlp.0.weight [64, 2*d_in+42]   1 edge_mlp.0.bias [64]
 *   2 edge_mlp.3.weight [64] (LayerNorm)  3 edge_mlp.3.bias [64]
 *   4 edge_mlp.4.weight [64,64]           5 edge_mlp.4.bias [64]
 *   6 att_mlp_Q.0.weight [d_in,d_in]      7 att_mlp_K.0.weight   8 att_mlp_V.0.weight
 *   9 node_mlp.0.weight [d_in, d0+2*d_in+64]  10 node_mlp.0.bias [d_in]
 *  11 node_mlp.3.weight [d_in]           12 node_mlp.3.bias [d_in]
 *  13 node_mlp.4.weight [64, d_in]       14 node_mlp.4.bias [64]
 *  15 coors_mlp.0.weight [64,64]         16 coors_mlp.0.bias [64]
 *  17 coors_mlp.4.weight [1,64]          18 coors_mlp.4.bias [1]
 * then (base = 19*n_layers): residue_emb_layer.weight [21,64], att_mlp_key_ROT.0.weight [K*64,64],
 * att_mlp_query_ROT.0.weight [K*64,64], mlp_h_mean_ROT.0.weight [64,64], mlp_h_mean_ROT.0.bias [64]
 * (names: rigid_docking_model.py:119-159, 382, 427-438).  With shared layers the same pointers
 * repeat for layers 1..L-1. */

/* Optional execution context: two auxiliary HIP streams + events that let eqd_model_forward/backward
 * overlap the independent branches of a layer (edge messages || cross attention || weight-gradient
 * GEMMs).  Created and destroyed by the caller (one per device); all auxiliary work is joined back
 * into the caller's stream before the functions return, so stream-ordered allocators stay correct.
 * Passing ctx = NULL runs everything on the caller's stream. */
int eqd_ctx_create(void** ctx);
int eqd_ctx_destroy(void* ctx);

/* workspace sizes (bytes) for a graph of the given dimensions */
size_t eqd_model_saved_bytes(const EqdModelDesc* m, const EqdGraph* g);    /* forward -> backward state */
size_t eqd_model_scratch_bytes(const EqdModelDesc* m, const EqdGraph* g);  /* transient, either pass */
/* Layout of the saved-state buffer under the current EQD_* switches: bit 0 = bf16 storage, bit 1 = q / k / v of the 64-wide
 * layers saved as bf16, bit 2 = per-edge state of the edge messages saved (EqdEdgeParams.xh_save).  A forward and the backward
 * that consumes its buffer must see the same value: do not call eqd_tunables_reload between them (-1: bad arguments). */
int eqd_model_saved_layout(const EqdModelDesc* m, const EqdGraph* g);
int eqd_model_check(const EqdModelDesc* m, const EqdGraph* g);

/* Test / debug aid: pointers to the node features h [n_nodes][*h_width] and coordinates x [n_nodes][3] after `layer`
 * layers (0 = embedding output and input coordinates, n_layers = what the keypoint head consumes; the reference stores the
 * latter as 'hv_iegmn_out' / 'x_iegmn_out', rigid_docking_model.py:507-510) inside the `saved` buffer of a forward. */
int eqd_model_layer_state(const EqdModelDesc* m, const EqdGraph* g, const void* saved, size_t saved_bytes, int layer,
                          const float** h, int* h_width, const float** x);

/* Test / debug aid: the LeakyReLU branch decisions of the forward whose state is in `saved`, one byte per element
 * (1 = pre-activation > 0, i.e. derivative 1; 0 = derivative lrelu_slope) - exactly the masks eqd_model_backward applies.
 * layer in [0, n_layers): edge_z1 / edge_ch [n_edges][64] = edge_mlp.0 / coors_mlp.0 outputs (rigid_docking_model.py:119-125,
 * 153-159; recomputed per edge tile like the backward does, same bits), node [n_nodes][d_in] = node_mlp.0 output (:142-148),
 * q / k [n_nodes][d_in] = att_mlp_Q / att_mlp_K outputs (:130-137; skipped without cross_msgs).  layer == n_layers: node
 * [n_nodes][64] = mlp_h_mean_ROT (:434-438).  Any output may be NULL (edge_z1 and edge_ch: both or neither).  A CPU oracle
 * evaluated with these slopes has ONE gradient to compare against, whatever the fp32 summation order did to
 * pre-activations within rounding of 0 (tests/parity_common.py). */
int eqd_model_lrelu_signs(const EqdModelDesc* m, const EqdGraph* g, const float* const* params, const EqdDropout* drop,
                          const void* saved, size_t saved_bytes, int layer, unsigned char* edge_z1, unsigned char* edge_ch,
                          unsigned char* node, unsigned char* q, unsigned char* k, void* stream);

/* Rigid_Body_Docking_Net.forward (rigid_docking_model.py:642-692) for the single-stage model:
 * embedding + L IEGMN layers + keypoint attention + Kabsch + rigid apply.
 * Outputs: lig_out [n_lig][3], Y_lig / Y_rec [n_pairs][n_heads][3], T [n_pairs][9], b [n_pairs][3],
 * svd_status [n_pairs] int32 (number of guard perturbations, 11 = "consistently unstable").
 * `svd_draws` may be NULL, or [n_pairs][10][3] diagonal perturbations to use when the guard
 * (:574) fires.  `saved` may be NULL for inference (no backward possible; the state then lives in `scratch`).  `drop`: the
 * dropout masks of a training-mode forward (EqdDropout), NULL otherwise.
 * bf16 storage mode (EqdModelDesc.storage_bf16): a forward that SAVES state also needs `scratch` (eqd_model_scratch_bytes) -
 * the fp32 tensors the forward itself reads but the backward only needs rounded to bf16 (the node features of the inner
 * layers, aggr_msg) are transients there, their saved form is bf16; EQD_ERR_WORKSPACE without it.  The same buffer may be
 * handed to the backward afterwards. */
int eqd_model_forward(const EqdModelDesc* m, const EqdGraph* g, const float* const* params, const EqdDropout* drop,
                      const float* svd_draws,
                      float* lig_out, float* Y_lig, float* Y_rec, float* T, float* b, int32_t* svd_status,
                      void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream, void* ctx);

/* Backward of the above (the autograd the reference gets from loss.backward(), src/train.py:154).
 * d_* are gradients w.r.t. the five outputs (any may be NULL = zero).  d_h_last [n_nodes][64] / d_x_last [n_nodes][3]
 * (usually NULL) are ADDED to the gradient w.r.t. the state after the last IEGMN layer (what the reference keeps as
 * 'hv_iegmn_out' / 'x_iegmn_out', rigid_docking_model.py:507-510): a loss on those node data, and the tests' way of
 * driving the backward of the layer stack with a fixed gradient, without the keypoint / Kabsch head.  Parameter gradients are
 * ACCUMULATED into grad_flat at grad_offsets[i] (in floats, same order as the parameter table;
 * shared entries share offsets); the caller zeroes grad_flat. */
int eqd_model_backward(const EqdModelDesc* m, const EqdGraph* g, const float* const* params, const EqdDropout* drop,
                       const float* d_lig, const float* d_Ylig, const float* d_Yrec, const float* d_T,
                       const float* d_b, const float* d_h_last, const float* d_x_last,
                       float* grad_flat, const int64_t* grad_offsets,
                       const void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream,
                       void* ctx);

/* The first stage of the backward on its own: the keypoint / Kabsch head (rigid_docking_model.py:521-600 and the rigid apply
 * :665) differentiated from the state a forward left in `saved`.  d_h_L [n_nodes][64] and d_x_L [n_nodes][3] RECEIVE the
 * gradient w.r.t. the state after the last IEGMN layer; the gradients of att_mlp_key_ROT / att_mlp_query_ROT /
 * mlp_h_mean_ROT are accumulated into grad_flat (the layer parameters' entries are not touched).  Together with
 * eqd_model_backward's d_h_last / d_x_last this splits the whole-model gradient at (h_L, x_L): the head is fp32 in every
 * mode, so the tests compare it plainly in bf16 mode too (tests/parity_common.py: check_head_backward). */
int eqd_model_head_backward(const EqdModelDesc* m, const EqdGraph* g, const float* const* params, const EqdDropout* drop,
                            const float* d_lig, const float* d_Ylig, const float* d_Yrec, const float* d_T,
                            const float* d_b, float* grad_flat, const int64_t* grad_offsets,
                            const void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                            float* d_h_L, float* d_x_L, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Operator-level entry points (used by the model functions above; exported for unit parity
 * tests and for callers that want a single IEGMN_Layer).
 * ------------------------------------------------------------------------------------------- */

/* Generic node-level building block: Y = alpha * f(sum_s X_s W_s^T + bias) + beta * R with
 * f = [LeakyReLU] [-> LayerNorm]; used for every nn.Linear of the path
 * (rigid_docking_model.py:119-159, 427-438) and, with transposed weight strides, for dX = dY W. */
typedef struct EqdLinSrc {
    const float* X;     /* [rows][ldx] */
    const float* mask;  /* optional: X is multiplied by LeakyReLU'(mask) (same layout as X) */
    const float* W;     /* element (m, k) at W[m * w_rs + k * w_cs] */
    int32_t ldx, K, w_rs, w_cs;
} EqdLinSrc;
typedef struct EqdLinJob {
    EqdLinSrc s[EQD_MAX_SRC];
    int32_t nsrc, M, act, rows;
    const float* bias;
    const float* ln_g; const float* ln_b; float* pre_ln; int32_t ld_pre;
    const float* R; int32_t ldr;
    float alpha, beta, slope, ln_eps;
    float* Y; int32_t ldy;
    int32_t bf16;   /* 1: the products run on v_mfma_f32_16x16x16_bf16 (X and W rounded to bf16 when the MFMA operands are
                       formed, fp32 accumulate; bias, activation, LayerNorm, residual fp32).  One mode per eqd_linear call
                       (the first job's). */
    const float* mul; int32_t ld_mul;   /* optional [rows][ld_mul]: the activation output is multiplied element-wise by it
                       before the LayerNorm - nn.Dropout between a Linear and its LeakyReLU in training mode
                       (rigid_docking_model.py:129, 435: LeakyReLU(keep * s * z) = keep * s * LeakyReLU(z)), entries are
                       0 or 1 / (1 - p), drawn by the caller.  NULL = none.  Jobs with `mul` run on the four-wave kernels. */
    int32_t pad_to;   /* > M: columns M .. pad_to - 1 of every output row are written as zeros (the zero padding of a
                         69-wide layer's 80-float attention rows, without a separate fill); 0 = none.  Honoured by the
                         general (M % 4 != 0) epilogue - the only place such widths occur. */
    uint16_t* Yb; int32_t ldyb;   /* optional bf16 COPY of the output, [rows][ldyb] (round to nearest even of the fp32
                         result; ldyb >= M).  Y may be NULL when only the copy is wanted.  bf16 storage of the saved state
                         (EqdModelDesc.storage_bf16): a tensor whose every later use rounds it to bf16 anyway - a GEMM
                         operand of the bf16 mode - is kept as bf16, with the same bits the consumer would have formed. */
} EqdLinJob;
int eqd_linear(const EqdLinJob* jobs /* host */, int njobs, void* stream);

/* dW[m][n] (+)= sum_rows (X * LeakyReLU'(xmask))[row][m] * Y[row][n]; optional column sums of X
 * (bias gradients).  Deterministic two-stage reduction through `partial`. */
typedef struct EqdAtbJob {
    const float* X; const float* xmask; int32_t ldx, M;
    const float* Y; int32_t ldy, N;
    int32_t rows;
    float* out; int32_t o_rs, o_cs;
    float* bias_out;
    float slope;
    float scale;   /* multiplier applied to both results; 0 means 1 */
    int32_t bf16;  /* 1: X^T Y on the bf16 MFMA (inputs rounded, fp32 accumulate); column sums stay fp32 */
    int32_t y_bf16; /* 1: Y points at bf16 rows (uint16_t [rows][ldy], ldy in elements, a multiple of 4, rows 8-byte aligned;
                       columns N .. round_up(N, 4) - 1 must be readable) - a saved tensor of the bf16 storage mode; with
                       bf16 = 1 the products round Y to bf16 anyway: the same bits as with the fp32 tensor */
} EqdAtbJob;
size_t eqd_atb_partial_bytes(const EqdAtbJob* jobs /* host */, int njobs);
int eqd_atb(const EqdAtbJob* jobs /* host */, int njobs, void* partial, size_t partial_bytes, void* stream);

/* Edge message + coordinate update of one IEGMN layer for both graphs of all pairs
 * (rigid_docking_model.py:204-237, 263-292): gathers, 15 RBFs, edge_mlp, coors_mlp, per-destination
 * means (DGL copy_edge + mean), x' = eta x0 + (1-eta) x + mean(x_rel * coef). */
typedef struct EqdEdgeParams {
    const float* W1; int32_t ldw1;  /* edge_mlp.0.weight [64][2*d_in+42]; columns >= 2*d_in are used here */
    int32_t d_in;
    const float* ln_g; const float* ln_b;
    const float* W2; const float* b2;
    const float* Wc1; const float* bc1; const float* wc2; const float* bc2;
    float slope, ln_eps, eta;
    int32_t use_dist, use_he;
    int32_t bf16;   /* 1: he from EqdGraph.he_bf16, GEMM inputs rounded to bf16, fp32 accumulate (bf16 MFMA) */
    /* nn.Dropout of edge_mlp.1 / coors_mlp.1 (rigid_docking_model.py:121,154) in training mode: keep masks drawn by the
     * caller, bit-packed [n_edges][2] uint32 in the graph's edge order (bit f of the 64-bit pair = feature f is kept),
     * kept elements are scaled by drop_scale = 1 / (1 - p).  Both NULL = no dropout (eval mode / p = 0). */
    const uint32_t* drop_z1; const uint32_t* drop_ch; float drop_scale;
    uint16_t* aggr_bf16;   /* optional (bf16 = 1 only): a bf16 copy of aggr_msg, [n_nodes][64], written by the forward beside
                              the fp32 result - the saved copy of the bf16 storage mode (the backward's weight-gradient
                              GEMM rounds aggr_msg to bf16 anyway); NULL = none */
    /* Saved per-edge state (round 6, optional; all three or none).  eqd_edge_message_fwd WRITES, for every edge of the graph's
     * edge order, the LayerNorm-normalised hidden row of edge_mlp (before the affine; fp32 [n_edges][64]), its 1 / std
     * ([n_edges]) and the sign bits of edge_mlp.0's output ([n_edges][4] uint16: bit 4 mb + r of word g = feature
     * 16 mb + 4 g + r is positive); eqd_edge_message_bwd READS them instead of gathering P[src] + Q[dst] and recomputing the
     * first Linear, the LeakyReLU and the LayerNorm statistics (268 B per edge in one contiguous row against 512 B from two
     * random rows): same bits either way.  NULL: the backward recomputes (and P, Q must be the forward's).  The model calls
     * leave it off by default (EQD_EDGE_SAVE=1 turns it on): measured, the forward's stores cost more than the backward gains
     * (DESIGN.md section 7). */
    float* xh_save; float* rstd_save; uint16_t* zpos_save;
} EqdEdgeParams;
/* P = h W1[:, :d_in]^T, Q = h W1[:, d_in:2 d_in]^T + b1 are node-level inputs ([n_nodes][64]). */
int eqd_edge_message_fwd(const EqdGraph* g, const EqdEdgeParams* p, const float* P, const float* Q,
                         const float* x, float* aggr_msg, float* x_new, void* stream);
/* Backward: recomputes the tile forward; outputs dP, dQ [n_nodes][64], dx [n_nodes][3]
 * (= (1-eta) d_xnew + geometric terms), and the parameter gradients (accumulated).  The weight-gradient
 * GEMMs (dW2, dWc1, dW1[:, 2d:]) are fused into the kernel: per-edge operands never reach HBM. */
typedef struct EqdEdgeGrads {
    float* dW1; int32_t ldw1;   /* only columns >= 2*d_in are written here */
    float* db1_unused;
    float* dln_g; float* dln_b; float* dW2; float* db2; float* dWc1; float* dbc1; float* dwc2; float* dbc2;
} EqdEdgeGrads;
size_t eqd_edge_message_bwd_workspace_bytes(const EqdGraph* g);
int eqd_edge_message_bwd(const EqdGraph* g, const EqdEdgeParams* p, const float* P, const float* Q,
                         const float* x, const float* d_aggr_msg, const float* d_xnew,
                         float* dP, float* dQ, float* dx, const EqdEdgeGrads* grads,
                         void* workspace, size_t ws_bytes, void* stream);

/* Profiling aid: only the per-edge backward kernel (k_edge_bwd) of eqd_edge_message_bwd. */
int eqd_edge_message_bwd_kernel_only(const EqdGraph* g, const EqdEdgeParams* p, const float* P, const float* Q,
                                     const float* x, const float* d_aggr_msg, const float* d_xnew,
                                     float* dQ, float* dx, void* workspace, size_t ws_bytes, void* stream);

/* Block-diagonal cross attention, both directions (rigid_docking_model.py:46-64, 244-256):
 * out_i = sum_j softmax_j(q_i . k_j) v_j over the partner protein of the same pair (no 1/sqrt(d)).
 * q, k, v, out: [n_nodes][d]; lse: [n_nodes]. */
int eqd_cross_attention_fwd(const EqdGraph* g, int d, const float* q, const float* k, const float* v,
                            float* out, float* lse, void* stream);
int eqd_cross_attention_bwd(const EqdGraph* g, int d, const float* q, const float* k, const float* v,
                            const float* out, const float* lse, const float* d_out,
                            float* dq, float* dk, float* dv, float* delta /* [n_nodes] scratch */, void* stream);
/* The same backward in the form eqd_model_backward takes for large batches (more attention work items than CUs; fp32,
 * d = 64): the key / value pass also writes its dS = P o (dP - delta) tiles into `ws` ([n_nodes][stride] floats, stride =
 * the longest protein rounded up to 32) and the dq pass is ONE contraction over them instead of a recompute of S and dP -
 * 5 GEMM units executed per (query tile, key tile) instead of 7, for 4 as written.  Same results up to fp32 summation
 * order.  Needs 16-byte aligned operands and an att_items count that is a multiple of 8 (the packers' interleaved list). */
size_t eqd_cross_attention_bwd_ds_workspace_bytes(const EqdGraph* g);
int eqd_cross_attention_bwd_ds(const EqdGraph* g, int d, const float* q, const float* k, const float* v,
                               const float* out, const float* lse, const float* d_out,
                               float* dq, float* dk, float* dv, void* ws, size_t ws_bytes, void* stream);

/* bf16 mode of the two calls above: every contraction (Q K^T, P V; dS K, dS^T Q, P^T dO, dO V^T) on
 * v_mfma_f32_16x16x16_bf16 - inputs rounded to bf16 (round to nearest even) when the MFMA operands are formed, fp32
 * accumulate; logits, softmax statistics, exponentials and outputs fp32.  d = 64 or 80 (zero-padded), 16-byte aligned
 * operands (what the model presents); otherwise EQD_ERR_UNSUPPORTED. */
int eqd_cross_attention_fwd_bf16(const EqdGraph* g, int d, const float* q, const float* k, const float* v,
                                 float* out, float* lse, void* stream);
int eqd_cross_attention_bwd_bf16(const EqdGraph* g, int d, const float* q, const float* k, const float* v,
                                 const float* out, const float* lse, const float* d_out,
                                 float* dq, float* dk, float* dv, float* delta, void* stream);

/* Node update of one IEGMN layer (rigid_docking_model.py:319-337; node_mlp = Linear, Dropout, LeakyReLU, LayerNorm,
 * Linear - :142-148 - and the skip connection :332-337):
 *   a1n = LayerNorm(mul * LeakyReLU([h | aggr_msg | aggr_cross | h0] Wn1^T + bn1)),  u = a1n Wn2^T + bn2,
 *   h_out = s u + (1 - s) h  when d_in == d_out, else u.
 * One row-chain launch forward; backward = one row chain (d a1n, LayerNorm / LeakyReLU backward, the four input gradients)
 * + the weight-gradient GEMMs (eqd_atb) + one deterministic reduction.  The same jobs eqd_model_forward / _backward
 * enqueue per layer. */
typedef struct EqdNodeUpdateParams {
    int32_t d_in;          /* width of h: 64, or d_emb + 5 = 69 for the first layer (4..80) */
    int32_t d0;            /* width of h0 = orig_h_feats_dim (4..80) */
    int32_t d_out;         /* out_feats_dim: 64 */
    int32_t ld_cross;      /* row stride of aggr_cross / d_aggr_cross (>= d_in; the attention operators' 16-float blocks: 80
                              for a 69-wide layer); not read when aggr_cross is NULL */
    const float* Wn1;      /* node_mlp.0.weight [d_in][d_in + 64 + d_in + d0], column blocks [h | aggr_msg | aggr_cross | h0] */
    const float* bn1;      /* node_mlp.0.bias [d_in] */
    const float* ln_g; const float* ln_b;   /* node_mlp.3 (LayerNorm) weight / bias [d_in] */
    const float* Wn2;      /* node_mlp.4.weight [d_out][d_in] */
    const float* bn2;      /* node_mlp.4.bias [d_out] */
    float skip_weight_h, slope, ln_eps;
    int32_t bf16;          /* 1: GEMM inputs rounded to bf16 (EqdLinJob.bf16 / EqdAtbJob.bf16) */
    const float* drop_mul; /* [rows][d_in] nn.Dropout factors of node_mlp.1 in training mode (0 or 1 / (1 - p)), or NULL */
} EqdNodeUpdateParams;
typedef struct EqdNodeUpdateGrads {   /* ACCUMULATED into (the caller zeroes them); shapes as the parameters */
    float* dWn1; float* dbn1; float* dln_g; float* dln_b; float* dWn2; float* dbn2;
} EqdNodeUpdateGrads;
/* h [rows][d_in], aggr_msg [rows][64], aggr_cross [rows][ld_cross] (NULL: no cross messages, the block is skipped),
 * h0 [rows][d0] -> h_out [rows][d_out]; y_act [rows][d_in] (the LayerNorm's input) and a1n [rows][d_in] (its output) are
 * the state the backward needs. */
int eqd_node_update_fwd(int rows, const EqdNodeUpdateParams* p, const float* h, const float* aggr_msg,
                        const float* aggr_cross, const float* h0, float* h_out, float* y_act, float* a1n, void* stream);
size_t eqd_node_update_bwd_workspace_bytes(int rows, const EqdNodeUpdateParams* p);
/* d_h_out [rows][d_out] -> d_h [rows][d_in], d_aggr_msg [rows][64], d_aggr_cross [rows][ld_cross] (columns d_in ..
 * ld_cross - 1 written as zeros when d_in is not a multiple of 4 - the 69-wide layer's padding -, else untouched),
 * d_h0 [rows][d0] (all written), parameter gradients accumulated. */
int eqd_node_update_bwd(int rows, const EqdNodeUpdateParams* p, const float* h, const float* aggr_msg,
                        const float* aggr_cross, const float* h0, const float* y_act, const float* a1n,
                        const float* d_h_out, float* d_h, float* d_aggr_msg, float* d_aggr_cross, float* d_h0,
                        const EqdNodeUpdateGrads* grads, void* workspace, size_t ws_bytes, void* stream);

/* K-head attention keypoint pooling (rigid_docking_model.py:521-560) with collapsed heads:
 * u[s][k] = W_K^(k)T (W_Q^(k) qmean[partner(s)]) / sqrt(d);  scores = H u^T; softmax over the
 * segment's nodes; Y = att^T Z.  qmean: [2B][64]; H [n_nodes][64]; Z [n_nodes][3];
 * outputs Y [2B][K][3] (ligand segments first), scores [n_nodes][K], lse [2B][K], qp [2B][K][64], u [2B][K][64]. */
int eqd_keypoint_pool_fwd(const EqdGraph* g, int n_heads, const float* Wk, const float* Wq, const float* qmean,
                          const float* H, const float* Z, float* Y, float* scores, float* lse,
                          float* qp, float* u, void* stream);

/* Backward of eqd_keypoint_pool_fwd (autograd of rigid_docking_model.py:521-560).  Inputs: the forward's operands and
 * saved outputs (scores, lse, qp, u) and dY [2B][K][3].  Outputs: dH [n_nodes][64], dZ [n_nodes][3] (written);
 * d_hm [n_nodes][64] (written) = gradient w.r.t. the ROWS whose per-segment means are `qmean`, i.e.
 * d qmean[seg(i)] / n_seg(i) for node i (the partner's keypoints consume qmean[seg]); dWk, dWq [K*64][64] are
 * ACCUMULATED (the caller zeroes them).  Deterministic (no atomics). */
size_t eqd_keypoint_pool_bwd_workspace_bytes(const EqdGraph* g, int n_heads);
int eqd_keypoint_pool_bwd(const EqdGraph* g, int n_heads, const float* Wk, const float* Wq, const float* qmean,
                          const float* qp, const float* u, const float* H, const float* Z, const float* scores,
                          const float* lse, const float* dY, float* dH, float* dZ, float* dWk, float* dWq, float* d_hm,
                          void* workspace, size_t ws_bytes, void* stream);

/* Kabsch / 3x3 SVD (rigid_docking_model.py:563-589): per pair A = (Yr - mean)^T (Yl - mean),
 * T = U diag(1,1,sign det A) V^T, b = mean_r - T mean_l.  Y: [2B][K][3].  A_out [B][9] is the
 * (possibly perturbed) matrix that was decomposed. */
int eqd_kabsch_fwd(int n_pairs, int n_heads, const float* Y, const float* svd_draws, int svd_seed,
                   float* T, float* b, float* A_out, int32_t* status, void* stream);
/* Closed-form backward. dY: [2B][K][3] is OVERWRITTEN with dL/dY. */
int eqd_kabsch_bwd(int n_pairs, int n_heads, const float* Y, const float* A, const float* T,
                   const float* dT, const float* db, float* dY, void* stream);

/* lig' = (T x^T)^T + b per pair (rigid_docking_model.py:665). */
int eqd_rigid_apply_fwd(const EqdGraph* g, const float* T, const float* b, float* lig_out, void* stream);
/* dT[b] += sum_i d_lig_i x_i^T, db[b] += sum_i d_lig_i (dT, db must be initialised by the caller). */
int eqd_rigid_apply_bwd(const EqdGraph* g, const float* d_lig, float* dT, float* db, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Loss terms right after the hot path, batched over the pairs of g instead of the
 * reference's Python loop over the minibatch (src/train.py:112-133):
 *   mse[p]   = nn.MSELoss(reduction='mean')(lig_pred_p, lig_target_p)                       (src/train.py:114, 274)
 *   inter[p] = compute_body_intersection_loss(lig_pred_p, rec_p, sigma, surface_ct)          (src/train.py:41-49)
 * lig_pred / lig_target: [n_lig][3] in the graph's ligand node order; rec: [n_rec][3] bound receptor coordinates in
 * the graph's receptor node order.  s_lig [n_lig] / s_rec [n_rec] receive the Gaussian sums of every node and are the
 * only state the backward needs.  The backward writes d lig_pred = d_mse[p] dmse/da + d_inter[p] dinter/da
 * (d_mse / d_inter: [n_pairs], NULL = zero); no other input carries a gradient in the reference either.
 * The pocket OT term is eqd_pocket_ot_* below (its exact solver runs on the host).
 * ------------------------------------------------------------------------------------------- */
int eqd_pair_losses_fwd(const EqdGraph* g, const float* lig_pred, const float* lig_target, const float* rec,
                        float sigma, float surface_ct, float* mse, float* inter, float* s_lig, float* s_rec,
                        void* stream);
int eqd_pair_losses_bwd(const EqdGraph* g, const float* lig_pred, const float* lig_target, const float* rec,
                        float sigma, float surface_ct, const float* s_lig, const float* s_rec, const float* d_mse,
                        const float* d_inter, float* d_lig_pred, void* stream);

/* Per-item random rigid augmentation of the ligand, for the whole batch (src/utils/db5_data.py:195-204): for pair p
 *   new_x_i = R_p (x_i - mean_p(x)) + t_p   for the pair's ligand nodes (x_lig / new_x: [n_lig][3], ligand node order),
 * and the same map for the pair's ligand pocket coordinates when pocket_off [n_pairs + 1] (device) is not NULL
 * (pocket_in / pocket_out [sum n_pocket][3]).  R [n_pairs][9] row-major, t [n_pairs][3]: drawn by the caller (the
 * reference draws them with scipy / numpy on the host, src/utils/protein_utils.py:15-23). */
int eqd_rigid_augment(const EqdGraph* g, const float* x_lig, const float* R, const float* t, float* new_x,
                      const int32_t* pocket_off, const float* pocket_in, float* pocket_out, void* stream);

/* Graph construction / featurisation (compute_dig_kNN_graph, src/utils/protein_utils.py:311-397), one protein per call,
 * fp64 like the reference's numpy code.  atoms [A][3] fp32 + atom_off [n + 1]: the all-atom coordinates of each residue;
 * x, n_i, u_i, v_i [n][3] fp64: (aligned) representative locations and local frames.
 *   eqd_protein_graph_distances: D [n][n] fp64 = mean all-atom distance between residues (inf on the diagonal)
 *   eqd_protein_graph_select   : per residue i its sources: the j with D[i][j] < cutoff in index order, or the
 *                                max_neighbor nearest in ascending distance when more qualify (np.argsort, :339-343) ->
 *                                nbr / nbr_dist [n][max_neighbor], deg [n]; and mu_r_norm [n][5] fp32 (:351-359)
 *   eqd_protein_graph_edges    : edge_off [n + 1] = exclusive prefix sum of deg (by the caller) -> destination-major
 *                                src / dst [E] int32 and he [E][27] fp32 = 15 RBFs exp(-d^2 / 1.5^k) (:71-86) followed by
 *                                the orientation features p, q, k, t in the destination's frame (:370-387) */
int eqd_protein_graph_distances(int n, const float* atoms, const int32_t* atom_off, double* D, void* stream);
int eqd_protein_graph_select(int n, int max_neighbor, double cutoff, const double* D, const double* x, int32_t* nbr,
                             double* nbr_dist, int32_t* deg, float* mu_r_norm, void* stream);
int eqd_protein_graph_edges(int n, int max_neighbor, const int32_t* edge_off, const int32_t* nbr, const double* nbr_dist,
                            const double* x, const double* n_i, const double* u_i, const double* v_i, int32_t* src,
                            int32_t* dst, float* he, void* stream);

/* Clash removal after docking (src/inference_rigid.py:207-234): gradient descent on 3 Euler angles (roll, yaw, pitch;
 * R = RZ RY RX, :46-73) and a translation applied to the docked ligand's atoms `lig0` [n_lig][3] against the receptor
 * atoms `rec` [n_rec][3] under compute_body_intersection_loss (reference: sigma = 8, surface_ct = 8), while
 * loss > loss_stop (0.5) and it < max_it (2000), with the reference's step sizes.  All state lives on the device:
 * `state` (zero-initialised by the caller: angles, translation, iteration counter, last loss, stop flag).  One call
 * enqueues n_iter iterations (4 launches each; they return immediately once the flag is up) - the caller reads
 * state->done every few dozen iterations.  Like the reference's loop, the iteration whose loss evaluates <= loss_stop
 * still applies its step (state->euler / trans are one step past that evaluation and state->it counts it; the final
 * ligand is R(euler) lig0 + trans); state->loss is the last EVALUATED loss.  The ligand positions of the last evaluated
 * iteration are the first n_lig * 3 floats of `workspace`.  float32 like the reference. */
typedef struct EqdClashState {
    float euler[3];
    float trans[3];
    float loss;
    int32_t it;
    int32_t done;
    int32_t reserved;
} EqdClashState;
size_t eqd_clash_workspace_bytes(int n_lig, int n_rec);
int eqd_clash_iterations(int n_iter, int n_lig, int n_rec, const float* lig0, const float* rec, float sigma, float surface_ct,
                         float loss_stop, int max_it, EqdClashState* state, void* workspace, size_t ws_bytes, void* stream);

/* Pocket optimal-transport term of the loss (src/train.py:117-129, src/utils/ot_utils.py:5-29), device side.  Pocket
 * rows of all pairs are stored one after the other: pocket_lig / pocket_rec [sum n_pocket][3] (matched rows: row i of
 * both is the same binding-pocket contact), pocket_off [n_pairs + 1] (device, int32); Y_* [n_pairs][n_heads][3].
 *   eqd_pocket_ot_cost: cost [sum n_pocket][n_heads] = compute_sq_dist_mat(pocket_lig_p, Y_lig_p) + (... rec ...)
 *   (the caller copies `cost` to the host, solves the exact transport problems with uniform marginals there -
 *    libequidock_host.so: eqd_host_emd_uniform, replacing POT's ot.emd - and copies `plan` back)
 *   eqd_pocket_ot_fwd : ot[p] = sum(plan_p * cost_p)
 *   eqd_pocket_ot_bwd : dY_lig / dY_rec (written) = d_ot[p] * d ot[p] / d Y, the plan being a constant as in the
 *                       reference (ot_mat_attached has requires_grad=False) */
int eqd_pocket_ot_cost(int n_pairs, int n_heads, const int32_t* pocket_off, const float* pocket_lig,
                       const float* pocket_rec, const float* Y_lig, const float* Y_rec, float* cost, void* stream);
int eqd_pocket_ot_fwd(int n_pairs, int n_heads, const int32_t* pocket_off, const float* plan, const float* cost,
                      float* ot, void* stream);
int eqd_pocket_ot_bwd(int n_pairs, int n_heads, const int32_t* pocket_off, const float* pocket_lig,
                      const float* pocket_rec, const float* Y_lig, const float* Y_rec, const float* plan,
                      const float* d_ot, float* dY_lig, float* dY_rec, void* stream);

/* Fixed scalar loss of the measurement harness (bench.py, __graft_entry__.smoke):
 *   loss = sum over pairs of mean(lig_p^2) + mean(Y_lig_p^2) + mean(Y_rec_p^2)
 * together with its gradients w.r.t. the three model outputs, in one launch (deterministic).  lig [n_lig][3],
 * Y_* [n_pairs][n_heads][3]; pair_loss [n_pairs] scratch; counter: one int32 that must be 0 before the first call
 * (the kernel resets it).  Nothing in the reference corresponds to it: its training losses are eqd_pair_losses_*. */
int eqd_scalar_loss(const EqdGraph* g, int n_heads, const float* lig, const float* Y_lig, const float* Y_rec,
                    float* d_lig, float* d_Ylig, float* d_Yrec, float* pair_loss, float* loss, int32_t* counter,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif
