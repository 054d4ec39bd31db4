// Model-level plans: the whole forward / backward of the DiffPool encoders as one launch sequence.
//
//   SoftPoolingGcnEncoder.forward   encoders.py:1231-1300   (num_pooling >= 1)
//   GcnEncoderGraph.forward         encoders.py:1083-1122   (num_pooling == 0, readout 0)
//   GcnSet2SetEncoder.forward       encoders.py:1144-1157   (num_pooling == 0, readout 1)
//
// Dataflow per pooling level j (n_j nodes, adjacency A_j, features X_j):
//   joint GCN stacks on A_j:  embed_j(X_j) -> Z_j   and, if j < P,  assign_j(Xa_j) -> Za_j
//       (the two stacks share every pass over A_j: their X·W products sit side by side in one
//        [B, n_j, Ce+Ca] operand — the reference runs them as separate bmm's, encoders.py:1254,1269)
//   readout_j = max_n Z_j                                        (:1257 / :1287)
//   S_j = softmax(Linear_j(Za_j)) * mask (j == 0)                 (:1273-1275)
//   X_{j+1} = S_j^T Z_j,  A_{j+1} = S_j^T A_j S_j                 (:1278-1279)
// ypred = pred_model(cat_j readout_j)                             (:1295-1299)
#include "dp_common.h"

namespace dp {

void set2set_fwd(Seq& q, const float* emb, int lde, const float* w_ih, const float* w_hh, const float* b_ih,
                 const float* b_hh, const float* Wp, const float* bp, float* out, int B, int n, int d, void* save);
void set2set_bwd(Seq& q, const float* emb, int lde, const float* w_ih, const float* w_hh, const float* b_ih,
                 const float* b_hh, const float* Wp, const float* bp, const float* out, const float* dout,
                 float* demb, int ldde, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, float* dWp,
                 float* dbp, int B, int n, int d, const void* save);
size_t set2set_save_bytes(int B, int n, int d);

namespace {

struct LevelInfo {
    int j;         // pooling level
    int n, G, L;
    const dp_stack_cfg* e;
    const dp_stack_cfg* a;
    int D, Da, K;
    int ctot[DP_MAX_LAYERS];
    int coff_e[DP_MAX_LAYERS], coff_a[DP_MAX_LAYERS];
    int cmax;
};

LevelInfo level_info(const dp_encoder_cfg& c, int j) {
    LevelInfo li{};
    li.j = j;
    li.n = c.n_nodes[j];
    li.e = &c.embed[j];
    li.L = li.e->n_layers;
    li.G = (j < c.num_pooling) ? 2 : 1;
    li.a = li.G == 2 ? &c.assign[j] : nullptr;
    li.K = li.G == 2 ? c.n_nodes[j + 1] : 0;
    li.D = li.Da = li.cmax = 0;
    for (int l = 0; l < li.L; ++l) {
        li.coff_e[l] = li.D;
        li.D += li.e->dims[l + 1];
        li.ctot[l] = li.e->dims[l + 1];
        if (li.a) {
            li.coff_a[l] = li.Da;
            li.Da += li.a->dims[l + 1];
            li.ctot[l] += li.a->dims[l + 1];
        }
        if (li.ctot[l] > li.cmax) li.cmax = li.ctot[l];
    }
    return li;
}

// Association of layer l's GraphConv product.  The reference computes (A x) W (encoders.py:966-968); the plan
// normally runs A (x W), which is cheaper when a layer narrows (DD: 89 -> 20) and lets the two stacks share a pass.
// A layer that WIDENS a lot — the assign stack's last layer at big pooling sizes (20 -> 256: 276 joint output columns
// against 40 input columns) — is run in the reference's own order instead: the adjacency pass carries the 40 input
// columns (HBM-bound on A) rather than 276 (MFMA-bound), forward and backward.  Level 0 only (no adjacency gradient),
// layers >= 1 (the input is the previous layer's output), no dropout mask on the layer.
int layer_cin(const LevelInfo& li, int l) { return li.e->dims[l] + (li.a ? li.a->dims[l] : 0); }
bool layer_agg_first(const LevelInfo& li, int l) {
    if (knobs().no_agg_first || li.j != 0 || l < 1) return false;
    if (li.e->drop_off[l] >= 0 || (li.a && li.a->drop_off[l] >= 0)) return false;
    const int cin = layer_cin(li, l), ct = li.ctot[l];
    return ct > 128 && ct >= 2 * cin;
}

// A bump allocator over the caller's save buffer (dry when base == nullptr).
struct Bump {
    char* base;
    size_t off;
    template <typename T>
    T* take(size_t n) {
        T* p = base ? (T*)(base + off) : nullptr;
        off += align256(n * sizeof(T));
        return p;
    }
};

struct LayerSave {
    float* Y;      // normalised pre-ReLU output of a non-last layer, joint [B, n, ctot]
    float* invn;   // [B, n, G]
    float* stats;  // [n, G, 2] (mu, rstd)
    float* Uin;    // aggregate-first layers: A [x_e | x_a] (+ x), [B, n, cin] — the left operand of dW
};
struct LevelSave {
    float* Ze;
    float* Za;
    LayerSave layer[DP_MAX_LAYERS];
    float* S;
    float* T;     // Tt = A^T S, [B, n, K]
    float* Xn;
    float* An;
    int* argmax;
};
struct SaveLayout {
    unsigned short* pkA;     // bf16 copy of the level-0 adjacency [B, N, pk_ld] (null: not packed)
    unsigned short* pkAt;    // ... of its transpose
    int* pk_flag;
    int pk_ld;
    LevelSave lv[DP_MAX_LEVELS + 1];
    float* feat;
    float* hid[DP_MAX_PRED + 2];
    float* Zm;
    void* s2s;
    size_t total;
};

int readout_width(const dp_encoder_cfg& c, const LevelInfo& li) {
    return (c.flags & DP_F_LAST_ONLY) ? li.e->dims[li.L] : li.D;
}

bool level0_persistent(const dp_encoder_cfg& c);

SaveLayout layout_save(const dp_encoder_cfg& c, void* base) {
    SaveLayout s{};
    Bump b{(char*)base, 0};
    const size_t B = c.B;
    // the bf16 copies of A / A^T: for the packed aggregation kernels (N >= 128) and for the persistent level-0 pair
    // (which keeps them for its backward at any N it takes)
    if (adj_pack_supported(c.N, 1) || level0_persistent(c)) {
        s.pk_ld = adj_pack_ld(c.N);
        s.pkA = b.take<unsigned short>(B * c.N * s.pk_ld);
        s.pkAt = b.take<unsigned short>(B * c.N * s.pk_ld);
        s.pk_flag = b.take<int>(64);
        if (!base) s.pkA = s.pkAt = reinterpret_cast<unsigned short*>(1);   // dry run: "packing enabled" marker
    }
    for (int j = 0; j <= c.num_pooling; ++j) {
        const LevelInfo li = level_info(c, j);
        LevelSave& lv = s.lv[j];
        lv.Ze = b.take<float>(B * li.n * li.D);
        lv.Za = li.a ? b.take<float>(B * li.n * li.Da) : nullptr;
        for (int l = 0; l < li.L; ++l) {
            lv.layer[l].Y = (l < li.L - 1) ? b.take<float>(B * li.n * li.ctot[l]) : nullptr;
            lv.layer[l].invn = b.take<float>(B * li.n * li.G);
            lv.layer[l].stats = (l < li.L - 1) ? b.take<float>((size_t)li.n * li.G * 2) : nullptr;
            lv.layer[l].Uin = layer_agg_first(li, l) ? b.take<float>(B * li.n * layer_cin(li, l)) : nullptr;
        }
        if (li.a) {
            lv.S = b.take<float>(B * li.n * li.K);
            lv.T = b.take<float>(B * li.K * li.n);
            lv.Xn = b.take<float>(B * li.K * li.D);
            lv.An = b.take<float>(B * li.K * li.K);
        }
        lv.argmax = b.take<int>(B * readout_width(c, li));
    }
    s.feat = b.take<float>(B * c.pred_dims[0]);
    s.hid[0] = s.feat;
    for (int i = 1; i < c.n_pred; ++i) s.hid[i] = b.take<float>(B * c.pred_dims[i]);
    if (c.readout == 1) {
        const LevelInfo li = level_info(c, 0);
        s.Zm = b.take<float>(B * li.n * li.D);
        s.s2s = b.take<char>(set2set_save_bytes(c.B, li.n, li.D));
    }
    s.total = b.off;
    return s;
}

int validate(const dp_encoder_cfg* c) {
    DP_CHECK_ARG(c != nullptr, "cfg is NULL");
    DP_CHECK_ARG(c->B > 0 && c->N > 0, "B=%d N=%d must be positive", c->B, c->N);
    DP_CHECK_ARG(c->num_pooling >= 0 && c->num_pooling <= DP_MAX_LEVELS, "num_pooling=%d out of range [0,%d]",
                 c->num_pooling, DP_MAX_LEVELS);
    DP_CHECK_ARG(c->n_nodes[0] == c->N, "n_nodes[0]=%d must equal N=%d", c->n_nodes[0], c->N);
    DP_CHECK_ARG(c->n_pred >= 1 && c->n_pred <= DP_MAX_PRED + 1, "n_pred=%d out of range", c->n_pred);
    for (int j = 0; j <= c->num_pooling; ++j) {
        const dp_stack_cfg& e = c->embed[j];
        DP_CHECK_ARG(e.n_layers >= 1 && e.n_layers <= DP_MAX_LAYERS, "embed[%d].n_layers=%d out of range", j,
                     e.n_layers);
        DP_CHECK_ARG(c->n_nodes[j] > 0, "n_nodes[%d]=%d must be positive (assign_ratio too small?)", j,
                     c->n_nodes[j]);
        for (int l = 0; l <= e.n_layers; ++l) DP_CHECK_ARG(e.dims[l] > 0, "embed[%d].dims[%d] must be > 0", j, l);
        DP_CHECK_ARG(e.drop_off[0] < 0 && (j >= c->num_pooling || c->assign[j].drop_off[0] < 0),
                     "level %d: layer 0 (conv_first) takes no dropout mask (encoders.py:1010-1012)", j);
        if (j < c->num_pooling) {
            const dp_stack_cfg& a = c->assign[j];
            DP_CHECK_ARG(a.n_layers == e.n_layers,
                         "assign[%d].n_layers=%d must equal embed n_layers=%d (the reference's assign_pred "
                         "input width assumes it, encoders.py:1209)", j, a.n_layers, e.n_layers);
            DP_CHECK_ARG(a.dims[a.n_layers] == c->n_nodes[j + 1], "assign[%d] output width %d != n_nodes[%d]=%d", j,
                         a.dims[a.n_layers], j + 1, c->n_nodes[j + 1]);
            for (int l = 0; l <= a.n_layers; ++l)
                DP_CHECK_ARG(a.dims[l] > 0, "assign[%d].dims[%d] must be > 0", j, l);
        }
    }
    if (c->readout == 1) DP_CHECK_ARG(c->num_pooling == 0, "Set2Set readout requires num_pooling == 0");
    {
        int feat = 0;
        for (int j = 0; j <= c->num_pooling; ++j) {
            const LevelInfo li = level_info(*c, j);
            if (j >= 1) {
                const LevelInfo lp = level_info(*c, j - 1);
                DP_CHECK_ARG(c->embed[j].dims[0] == lp.D, "embed[%d] input width %d != pooled feature width %d", j,
                             c->embed[j].dims[0], lp.D);
                if (j < c->num_pooling)
                    DP_CHECK_ARG(c->assign[j].dims[0] == lp.D, "assign[%d] input width %d != pooled feature width %d",
                                 j, c->assign[j].dims[0], lp.D);
            }
            feat += readout_width(*c, li);
        }
        DP_CHECK_ARG(c->pred_dims[0] == feat, "pred_dims[0]=%d != readout width %d", c->pred_dims[0], feat);
    }
    DP_CHECK_ARG(c->n_graph_params >= 0 && c->n_graph_params <= c->n_params, "n_graph_params out of range");
    DP_CHECK_ARG(c->bn_world >= 0 && c->bn_world <= 1024, "bn_world=%d out of range", c->bn_world);
    DP_CHECK_ARG(c->bn_world <= 1 || c->exchange, "bn_world=%d needs the exchange callback", c->bn_world);
    DP_CHECK_ARG(!(c->bn_world > 1 || (c->bn_world == 1 && c->exchange)) || c->readout == 0,
                 "sync-BN is implemented for the max-readout encoders");
    return DP_OK;
}

inline const float* PW(const float* params, long off) { return off >= 0 ? params + off : nullptr; }
inline float* PWm(float* params, long off) { return off >= 0 ? params + off : nullptr; }

RowGroups groups_of(const LevelInfo& li, int l) {
    RowGroups g{};
    g.G = li.G;
    g.c0[0] = 0;
    g.w[0] = li.e->dims[l + 1];
    g.c0[1] = li.e->dims[l + 1];
    g.w[1] = li.a ? li.a->dims[l + 1] : 0;
    return g;
}

// Split-K factor of the contractions over the node index (and the slab-row multiplier): small batches of
// big graphs need it to fill the chip, large batches do not.
int node_ksplit(const dp_encoder_cfg& c) {
    if (c.B >= 64) return 1;
    const int force = knobs().node_ksplit;   // tuning knob
    if (force >= 1 && force <= 8) return force;
    int ks = (c.N + 127) / 128;
    return ks < 1 ? 1 : (ks > 8 ? 8 : ks);
}

// The backward pass's zero-initialised accumulators — level gradient buffers and the per-graph parameter-gradient
// slabs (atomic bias sums; unused split-K rows) — are ONE block at the very start of the workspace in BOTH walks, so a
// training forward can clear it on the side of its adjacency-pack kernel and the backward pass starts without a
// zero-fill launch.
struct LevelGrad {
    float* dZe;    // [B, n, D]
    float* dZa;    // [B, n, Da]
    float* dX0;    // [B, n, dims_e[0]] gradient w.r.t. the level input (levels >= 1), accumulated
    float* dAdj;   // [B, n, n] (levels >= 1), accumulated
};
struct BwdZero {
    LevelGrad gr[DP_MAX_LEVELS + 1];
    float* slabs;
    int* bar;          // grid-barrier tickets of the whole-level backward kernels: 64 ints per level
    size_t begin, end;
};
BwdZero alloc_bwd_zero(Seq& q, const dp_encoder_cfg& c) {
    BwdZero z{};
    z.begin = q.ws_off;
    for (int j = 0; j <= c.num_pooling; ++j) {
        const LevelInfo li = level_info(c, j);
        const size_t rows = (size_t)c.B * li.n;
        z.gr[j].dZe = q.alloc<float>(rows * li.D);
        z.gr[j].dX0 = j >= 1 ? q.alloc<float>(rows * li.e->dims[0]) : nullptr;
        z.gr[j].dAdj = j >= 1 ? q.alloc<float>(rows * li.n) : nullptr;
    }
    z.slabs = q.alloc<float>((size_t)c.B * node_ksplit(c) * (c.n_graph_params > 0 ? c.n_graph_params : 1));
    z.bar = q.alloc<int>(64 * (DP_MAX_LEVELS + 1));
    z.end = q.ws_off;
    return z;
}

// sync-BN (cfg.bn_world > 1): every BatchNorm site gathers its row partials from all ranks before combining them.
// The one-workgroup-per-graph kernels and the whole-level kernels combine partials inside the launch, so they are
// not used in this mode: all levels run on the generic per-layer kernels.
inline int bn_world(const dp_encoder_cfg& c) { return c.bn_world > 1 ? c.bn_world : 1; }
// bn_world == 1 WITH an exchange callback is sync-BN over a single rank: the same launch sequence and the same
// callbacks as W > 1 (what a one-GPU box can execute of the RCCL path), statistics over the local batch
inline bool bn_sync(const dp_encoder_cfg& c) { return c.bn_world > 1 || (c.bn_world == 1 && c.exchange != nullptr); }
// local [B, n, G, 2] block -> gathered [world * B, n, G, 2]; returns the pointer the consumer should read
const float* bn_exchange(Seq& q, const dp_encoder_cfg& c, const float* local, float* gathered, size_t floats) {
    if (!bn_sync(c)) return local;
    if (!q.ok()) return gathered;
    const int rc = c.exchange(c.exchange_user, local, gathered, floats * sizeof(float), (void*)q.stream);
    if (rc != 0) {
        set_error("sync-BN exchange callback failed (code %d)", rc);
        q.err = DP_ERR_INVALID_ARG;
    }
    return gathered;
}

bool level_is_small(int B, const LevelInfo& li) {
    if (li.G != 1) return false;
    for (int l = 0; l < li.L; ++l)
        if (!small_level_supported(B, li.n, li.e->dims[l], li.e->dims[l + 1])) return false;
    return true;
}

// the level-0 adjacency pack, run right AFTER the first transform GEMM of the level: that GEMM does not read the
// adjacency and clears the pack flag on the side (Seq::fold_zero_p), so the flag needs no launch of its own
struct PackJob {
    const float* adj;
    unsigned short *pkA, *pkAt;
    int* flag;
    int ld;
    void* zero_p;        // backward accumulators to clear on the side (training forward), or null
    size_t zero_bytes;
};

struct LevelIO {
    const PackJob* pack = nullptr;
    const float* x0e;  // embed stack input [B, n, dims_e[0]]
    const float* x0a;  // assign stack input [B, n, dims_a[0]]
    const float* adj;  // [B, n, n]
    const float* drop = nullptr;   // dropout mask buffer (training with dropout > 0) or null
    float* xm[2] = {nullptr, nullptr};   // scratch [B, n, din] per stack: the masked layer input
};

// mask of stack gi's layer-l input, or null (GraphConv dropout, encoders.py:962-964)
const float* drop_mask(const LevelInfo& li, const LevelIO& io, int gi, int l) {
    const dp_stack_cfg* st = gi == 0 ? li.e : li.a;
    return (io.drop && st && st->drop_off[l] >= 0) ? io.drop + st->drop_off[l] : nullptr;
}
bool level_has_dropout(const LevelInfo& li, const LevelIO& io) {
    for (int gi = 0; gi < li.G; ++gi)
        for (int l = 0; l < li.L; ++l)
            if (drop_mask(li, io, gi, l)) return true;
    return false;
}

// every layer of a small level in one launch per direction (dp_small.hip, whole-level kernels)
bool level_is_fused(const dp_encoder_cfg& c, const LevelInfo& li, const LevelIO& io, bool dadj) {
    return level_is_small(c.B, li) && !level_has_dropout(li, io) &&
           small_level_fused_ok(c.B, li.n, li.e->dims, li.L, dadj);
}
// exchange scratch of the whole-level kernels over all levels that may use them
size_t level_part_floats(const dp_encoder_cfg& c) {
    size_t mx = 64;
    for (int j = 0; j <= c.num_pooling; ++j) {
        const LevelInfo li = level_info(c, j);
        if (li.G == 1 && li.n <= 64) {
            const size_t f = small_level_part_floats(c.B, li.n, li.L);
            mx = f > mx ? f : mx;
        }
    }
    return mx;
}
SmallLevelIO small_level_io(const LevelInfo& li, const LevelSave& lv, const LevelIO& io, const float* params,
                            float* part, int* bar) {
    SmallLevelIO s{};
    s.adj = io.adj;
    s.x0 = io.x0e;
    s.ldx0 = li.e->dims[0];
    s.params = params;
    for (int l = 0; l < li.L; ++l) {
        s.w_off[l] = li.e->w_off[l];
        s.b_off[l] = li.e->b_off[l];
        s.Y[l] = l < li.L - 1 ? lv.layer[l].Y : nullptr;
        s.ldY[l] = li.ctot[l];
        s.invn[l] = lv.layer[l].invn;
        s.stats[l] = lv.layer[l].stats;
        s.coff[l] = li.coff_e[l];
    }
    s.Ze = lv.Ze;
    s.ldz = li.D;
    s.part = part;
    s.bar = bar;
    return s;
}

// P = [x_e W_e | x_a W_a]  for layer l of level li  (one grouped launch)
void transform(Seq& q, const dp_encoder_cfg& c, const LevelInfo& li, const LevelSave& lv, const LevelIO& io,
               const float* params, int l, float* Pj, unsigned short* vs_split = nullptr) {
    const int ct = li.ctot[l];
    const int B = c.B, n = li.n;
    GemmDesc d[2];
    for (int g = 0; g < li.G; ++g) {
        const dp_stack_cfg* st = g == 0 ? li.e : li.a;
        const int din = st->dims[l], dout = st->dims[l + 1];
        const float* xin;
        int ldin;
        if (l == 0) {
            xin = g == 0 ? io.x0e : io.x0a;
            ldin = din;
        } else {
            xin = (g == 0 ? lv.Ze + li.coff_e[l - 1] : lv.Za + li.coff_a[l - 1]);
            ldin = g == 0 ? li.D : li.Da;
        }
        if (const float* m = drop_mask(li, io, g, l)) {        // x <- dropout(x) for this GraphConv only
            mask_mul(q, xin, ldin, m, io.xm[g], (long)B * n, din);
            xin = io.xm[g];
            ldin = din;
        }
        const int c0 = g == 0 ? 0 : li.e->dims[l + 1];
        d[g] = GemmDesc{xin, PW(params, st->w_off[l]), Pj + c0, nullptr, n, dout, din, ldin, dout, ct,
                        (long)n * ldin, 0, (long)n * ct, false, false, 1.f, 0.f, 0, 0, 0,
                        vs_split, (ct + 15) / 16, ((n + 31) / 32) * 4, c0};
    }
    bgemm_group(q, d, li.G, B);
}

void level_forward(Seq& q, const dp_encoder_cfg& c, const LevelInfo& li, const LevelSave& lv, const LevelIO& io,
                   const float* params, float* Pj, float* Uj, float* part, float* part_b, const PackedAdj* pk,
                   unsigned short* vs, float* lvl_part, int* bar /*zeroed in stream order, or null*/,
                   float* part_all /*[world * B, n, G, 2] under sync-BN*/) {
    const int B = c.B, n = li.n;
    const bool bn = c.flags & DP_F_BN;
    const bool add_self = c.flags & DP_F_ADD_SELF;
    const int W = bn_world(c);
    if (!bn_sync(c) && bar && level_is_fused(c, li, io, true)) {
        small_level_fwd(q, small_level_io(li, lv, io, params, lvl_part, bar), B, n, li.e->dims, li.L, add_self ? 1 : 0,
                        bn ? 1 : 0);
        return;
    }
    if (!bn_sync(c) && level_is_small(B, li) && !level_has_dropout(li, io)) {
        // pooled level (or tiny graphs): one launch per layer, one workgroup per graph (dp_small.hip)
        float* pbuf[2] = {part, part_b};
        for (int l = 0; l < li.L; ++l) {
            const bool last = l == li.L - 1;
            const int din = li.e->dims[l], dout = li.e->dims[l + 1];
            small_gcn_fwd(q, io.adj, l == 0 ? io.x0e : nullptr, din, l > 0 ? lv.layer[l - 1].Y : nullptr,
                          l > 0 ? li.ctot[l - 1] : 0, (l > 0 && bn) ? pbuf[(l - 1) & 1] : nullptr,
                          l > 0 ? lv.layer[l - 1].stats : nullptr, l > 0 ? lv.Ze + li.coff_e[l - 1] : nullptr, li.D,
                          PW(params, li.e->w_off[l]), PW(params, li.e->b_off[l]),
                          last ? lv.Ze + li.coff_e[l] : lv.layer[l].Y, last ? li.D : li.ctot[l], lv.layer[l].invn,
                          (!last && bn) ? pbuf[l & 1] : nullptr, B, n, din, dout, add_self ? 1 : 0,
                          (!last && bn) ? 1 : 0);
        }
        return;
    }
    bool transformed = false;      // layer l's P = x W was already written by the previous layer's BN launch
    for (int l = 0; l < li.L; ++l) {
        const int ct = li.ctot[l];
        const bool last = l == li.L - 1;
        // with a packed adjacency the transform GEMM also emits the 3-plane bf16 split the aggregation reads
        const bool presplit = pk && vs && aggregate_packed_usable(io.adj, n, ct);
        if (l == 0 && io.pack) {
            q.fold_zero_p = io.pack->flag;
            q.fold_zero_n16 = 16;
        }
        const bool agg_first = layer_agg_first(li, l);
        bool widen_fused = false;
        if (agg_first) {
            // (A [x_e | x_a]) W: gather the two stacks' inputs, ONE narrow pass over A, then the row-local products
            const int cin = layer_cin(li, l), de = li.e->dims[l], da = li.a ? li.a->dims[l] : 0;
            float* Uin = lv.layer[l].Uin;
            gather_cols(q, lv.Ze + li.coff_e[l - 1], li.D, de, li.a ? lv.Za + li.coff_a[l - 1] : nullptr, li.Da, da, Pj,
                        (long)B * n);
            aggregate(q, io.adj, Pj, cin, Uin, cin, B, n, cin, false, 0.f, pk, vs, false);
            if (add_self) axpy(q, Uin, Pj, 1.f, (long)B * n * cin);
            const int dins[2] = {de, da}, c0ins[2] = {0, de};
            widen_fused = last && widen_fwd_supported(groups_of(li, l), dins);
            if (!widen_fused) {
                GemmDesc d[2];
                for (int gi = 0; gi < li.G; ++gi) {
                    const dp_stack_cfg* st = gi == 0 ? li.e : li.a;
                    const int c0 = gi == 0 ? 0 : li.e->dims[l + 1], c0in = gi == 0 ? 0 : de;
                    d[gi] = GemmDesc{Uin + c0in, PW(params, st->w_off[l]), Pj + c0, nullptr, n, st->dims[l + 1],
                                     st->dims[l], cin, st->dims[l + 1], ct, (long)n * cin, 0, (long)n * ct, false, false,
                                     1.f, 0.f, 0};
                }
                bgemm_group(q, d, li.G, B);
            } else {
                // the row-local products and the GraphConv tail in one row kernel (the last layer has no statistics)
                const float* Wg[2] = {PW(params, li.e->w_off[l]), li.a ? PW(params, li.a->w_off[l]) : nullptr};
                GroupCPtrs wb{};
                wb.p[0] = PW(params, li.e->b_off[l]);
                wb.p[1] = li.a ? PW(params, li.a->b_off[l]) : nullptr;
                GroupPtrs wy{};
                wy.p[0] = lv.Ze + li.coff_e[l];
                wy.ld[0] = li.D;
                wy.p[1] = li.a ? lv.Za + li.coff_a[l] : nullptr;
                wy.ld[1] = li.Da;
                widen_fwd(q, Uin, cin, c0ins, dins, Wg, wb, groups_of(li, l), wy, lv.layer[l].invn, (long)B * n, 1);
            }
        } else if (!transformed) {
            transform(q, c, li, lv, io, params, l, Pj, presplit ? vs : nullptr);
        }
        transformed = false;
        if (l == 0 && io.pack) {
            const bool folded = q.fold_zero_p == nullptr;      // consumed by the GEMM launch
            q.fold_zero_p = nullptr;
            q.fold_zero_n16 = 0;
            adj_pack(q, io.pack->adj, io.pack->pkA, io.pack->pkAt, io.pack->flag, B, n, io.pack->ld, folded,
                     io.pack->zero_p, io.pack->zero_bytes);
        }
        RowGroups g = groups_of(li, l);
        GroupCPtrs bias{};
        bias.p[0] = PW(params, li.e->b_off[l]);
        bias.p[1] = li.a ? PW(params, li.a->b_off[l]) : nullptr;
        GroupPtrs yout{};
        if (last) {
            yout.p[0] = lv.Ze + li.coff_e[l];
            yout.ld[0] = li.D;
            yout.p[1] = li.a ? lv.Za + li.coff_a[l] : nullptr;
            yout.ld[1] = li.Da;
        } else {
            yout.p[0] = lv.layer[l].Y;
            yout.ld[0] = ct;
            yout.p[1] = lv.layer[l].Y + g.c0[1];
            yout.ld[1] = ct;
        }
        const int stats_mode = (!last && bn) ? 1 : 0;
        // aggregation + GraphConv tail in one launch when the panel kernel takes the shape
        if (widen_fused) {
            // (written by widen_fwd above)
        } else if (agg_first) {
            rownorm_fwd(q, Pj, ct, nullptr, bias, g, yout, lv.layer[l].invn, stats_mode ? part : nullptr, (long)B * n, 1,
                        stats_mode);
        } else if (!aggregate_rownorm_fwd(q, io.adj, Pj, ct, add_self ? Pj : nullptr, bias, g, yout, lv.layer[l].invn,
                                   stats_mode ? part : nullptr, B, n, 1, stats_mode, pk, vs, presplit)) {
            // wide / odd shapes: plain aggregation (panel kernel with column chunks, or the generic GEMM) + row pass
            aggregate(q, io.adj, Pj, ct, Uj, ct, B, n, ct, false, 0.f, pk, vs, presplit);
            rownorm_fwd(q, Uj, ct, add_self ? Pj : nullptr, bias, g, yout, lv.layer[l].invn,
                        stats_mode ? part : nullptr, (long)B * n, 1, stats_mode);
        }
        if (!last) {
            GroupPtrs xout{};
            xout.p[0] = lv.Ze + li.coff_e[l];
            xout.ld[0] = li.D;
            xout.p[1] = li.a ? lv.Za + li.coff_a[l] : nullptr;
            xout.ld[1] = li.Da;
            // apply_bn and the NEXT layer's transform in one launch when that transform is a plain row-local product
            const RowGroups gnext = groups_of(li, l + 1);
            const float* part_r = bn ? bn_exchange(q, c, part, part_all, (size_t)B * n * li.G * 2) : nullptr;
            bool plain_next = !bn_sync(c) && !knobs().no_level_fusion && bn_transform_supported(g, gnext, B) &&
                              !layer_agg_first(li, l + 1);
            for (int gi = 0; gi < li.G; ++gi) plain_next = plain_next && !drop_mask(li, io, gi, l + 1);
            if (plain_next) {
                const float* Wn[2] = {PW(params, li.e->w_off[l + 1]), li.a ? PW(params, li.a->w_off[l + 1]) : nullptr};
                const int ctn = li.ctot[l + 1];
                const bool presplit_n = pk && vs && aggregate_packed_usable(io.adj, n, ctn);
                bn_transform_fwd(q, lv.layer[l].Y, ct, bn ? part : nullptr, lv.layer[l].stats, g, xout, Wn, gnext, Pj,
                                 ctn, B, n, 1, presplit_n ? vs : nullptr);
                transformed = true;
            } else {
                bn_apply_fwd(q, lv.layer[l].Y, ct, part_r, lv.layer[l].stats, g, xout, B, n, 1, B * W);
            }
        }
    }
}

void level_backward(Seq& q, const dp_encoder_cfg& c, const LevelInfo& li, const LevelSave& lv, const LevelIO& io,
                    const float* params, const LevelGrad& gr, float* slabs, long slab_stride, int KS, float* Pj,
                    float* dUj, float* Gj, float* part, float* part_b, const PackedAdj* pk, unsigned short* vs,
                    float* const dxm[2], float* lvl_part, int* bar, float* part_all) {
    const int B = c.B, n = li.n;
    const int W = bn_world(c);
    const long gstride = slab_stride * KS;     // slab rows of one graph: KS split-K partials
    const int ks_level = n >= 256 ? KS : 1;
    const bool bn = c.flags & DP_F_BN;
    const bool add_self = c.flags & DP_F_ADD_SELF;
    if (!bn_sync(c) && lvl_part && bar && level_is_fused(c, li, io, true)) {
        small_level_bwd(q, small_level_io(li, lv, io, params, lvl_part, bar), gr.dZe, gr.dX0, gr.dAdj, slabs, gstride,
                        B, n, li.e->dims, li.L, add_self ? 1 : 0, bn ? 1 : 0);
        return;
    }
    if (!bn_sync(c) && level_is_small(B, li) && !level_has_dropout(li, io)) {
        float* pbuf[2] = {part, part_b};
        for (int l = li.L - 1; l >= 0; --l) {
            const bool last = l == li.L - 1;
            const bool has_bn = !last && bn;
            const int din = li.e->dims[l], dout = li.e->dims[l + 1];
            const float* xin = l == 0 ? io.x0e : lv.Ze + li.coff_e[l - 1];
            const int ldxin = l == 0 ? din : li.D;
            float* dxin = l > 0 ? gr.dZe + li.coff_e[l - 1] : gr.dX0;
            const int lddxin = l > 0 ? li.D : din;
            small_gcn_bwd(q, io.adj, xin, ldxin, PW(params, li.e->w_off[l]),
                          last ? lv.Ze + li.coff_e[l] : lv.layer[l].Y, last ? li.D : li.ctot[l],
                          has_bn ? lv.Ze + li.coff_e[l] : nullptr, li.D, lv.layer[l].invn, lv.layer[l].stats,
                          has_bn ? pbuf[l & 1] : nullptr, gr.dZe + li.coff_e[l], li.D, dxin, lddxin,
                          (l > 0 && bn && dxin) ? pbuf[(l - 1) & 1] : nullptr, gr.dAdj, slabs + li.e->w_off[l],
                          li.e->b_off[l] >= 0 ? slabs + li.e->b_off[l] : nullptr, gstride, B, n, din, dout,
                          add_self ? 1 : 0, has_bn ? 1 : 0, last ? 0 : 1);
        }
        return;
    }
    bool part_ready = false;       // layer l's BatchNorm-backward partials were written by layer l + 1's dX GEMM
    for (int l = li.L - 1; l >= 0; --l) {
        const int ct = li.ctot[l];
        const bool last = l == li.L - 1;
        RowGroups g = groups_of(li, l);
        GroupCPtrs dx{}, xhat{}, y{};
        dx.p[0] = gr.dZe + li.coff_e[l];
        dx.ld[0] = li.D;
        xhat.p[0] = lv.Ze + li.coff_e[l];
        xhat.ld[0] = li.D;
        if (li.a) {
            dx.p[1] = gr.dZa + li.coff_a[l];
            dx.ld[1] = li.Da;
            xhat.p[1] = lv.Za + li.coff_a[l];
            xhat.ld[1] = li.Da;
        }
        if (last) {
            y = xhat;
        } else {
            y.p[0] = lv.layer[l].Y;
            y.ld[0] = ct;
            y.p[1] = lv.layer[l].Y + g.c0[1];
            y.ld[1] = ct;
        }
        const bool has_bn = !last && bn;
        const float* part2 = part;
        if (has_bn) {
            if (!part_ready) bn_bwd_partials(q, dx, xhat, g, part, (long)B * n);
            part2 = bn_exchange(q, c, part, part_all, (size_t)B * n * li.G * 2);
        }
        part_ready = false;
        // bias gradients: column sums of dU go straight into each graph's (zeroed) slab row with float atomics
        GroupPtrs dbias{};
        dbias.p[0] = li.e->b_off[l] >= 0 ? slabs + li.e->b_off[l] : nullptr;
        dbias.p[1] = (li.a && li.a->b_off[l] >= 0) ? slabs + li.a->b_off[l] : nullptr;
        dbias.ld[0] = dbias.ld[1] = (int)gstride;
        const bool agg_first = layer_agg_first(li, l) && !gr.dAdj;
        const bool presplit = !agg_first && pk && vs && aggregate_packed_usable(io.adj, n, ct);
        bool mv_fused = false;       // d(A x) = dV W^T already written by the row kernel
        if (agg_first) {
            const int dins[2] = {li.e->dims[l], li.a ? li.a->dims[l] : 0}, c0ins[2] = {0, li.e->dims[l]};
            mv_fused = rownorm_bwd_mv_supported(g, dins, n);
            if (mv_fused) {
                const float* Wg[2] = {PW(params, li.e->w_off[l]), li.a ? PW(params, li.a->w_off[l]) : nullptr};
                rownorm_bwd_mv(q, dx, xhat, y, lv.layer[l].invn, lv.layer[l].stats, part2, g, dUj, ct, &dbias, B, n,
                               !last, has_bn, 1, B * W, Wg, dins, c0ins, Gj, layer_cin(li, l));
            }
        }
        if (!mv_fused)
            rownorm_bwd(q, dx, xhat, y, lv.layer[l].invn, lv.layer[l].stats, part2, g, dUj, ct, &dbias, B, n, !last,
                        has_bn, 1, presplit ? vs : nullptr, B * W);
        if (agg_first) {
            // y = (A x) W: dW = (A x)^T dV with the saved A x;  d(A x) = dV W^T (row-local);  dx += A^T d(A x), one
            // narrow pass over A
            const int cin = layer_cin(li, l), de = li.e->dims[l], da = li.a ? li.a->dims[l] : 0;
            const float* Uin = lv.layer[l].Uin;
            GemmDesc d[4];
            int nd = 0;
            for (int gi = 0; gi < li.G; ++gi) {
                const dp_stack_cfg* st = gi == 0 ? li.e : li.a;
                const int din = st->dims[l], dout = st->dims[l + 1], c0in = gi == 0 ? 0 : de;
                d[nd++] = GemmDesc{Uin + c0in, dUj + g.c0[gi], slabs + st->w_off[l], nullptr, din, dout, n, cin, ct, dout,
                                   (long)n * cin, (long)n * ct, gstride, true, false, 1.f, 0.f, 0, slab_stride, 0};
                if (!mv_fused)
                    d[nd++] = GemmDesc{dUj + g.c0[gi], PW(params, st->w_off[l]), Gj + c0in, nullptr, n, din, dout, ct,
                                       dout, cin, (long)n * ct, 0, (long)n * cin, false, true, 1.f, 0.f, 0, 0, 0, nullptr,
                                       0, 0, 0, 1};
            }
            bgemm_group(q, d, nd, B, ks_level);
            float* dxagg = dUj;                    // dV is consumed: its buffer takes A^T d(A x)
            aggregate(q, io.adj, Gj, cin, dxagg, cin, B, n, cin, true, 0.f, pk, vs, false);
            if (add_self) axpy(q, dxagg, Gj, 1.f, (long)B * n * cin);
            if (bn && !knobs().no_rowpart_hook && scatter_add_cols_part_supported(de, da)) {
                // ... and the BatchNorm-backward partials of layer l - 1, whose gradient these rows complete
                GroupPtrs dd{};
                GroupCPtrs xh{};
                dd.p[0] = gr.dZe + li.coff_e[l - 1];
                dd.ld[0] = li.D;
                xh.p[0] = lv.Ze + li.coff_e[l - 1];
                xh.ld[0] = li.D;
                if (li.a) {
                    dd.p[1] = gr.dZa + li.coff_a[l - 1];
                    dd.ld[1] = li.Da;
                    xh.p[1] = lv.Za + li.coff_a[l - 1];
                    xh.ld[1] = li.Da;
                }
                scatter_add_cols_part(q, dxagg, dd, xh, de, da, li.G, part, (long)B * n);
                part_ready = true;
            } else {
                scatter_add_cols(q, dxagg, gr.dZe + li.coff_e[l - 1], li.D, de,
                                 li.a ? gr.dZa + li.coff_a[l - 1] : nullptr, li.Da, da, (long)B * n);
            }
            continue;
        }
        // G = A^T dU (+ dU)
        aggregate(q, io.adj, dUj, ct, Gj, ct, B, n, ct, true, 0.f, pk, vs, presplit);
        if (add_self) axpy(q, Gj, dUj, 1.f, (long)B * n * ct);
        {
            GemmDesc d[4];
            int nd = 0;
            // The dX products of this layer complete the gradient of layer l - 1's output: when that layer has a
            // BatchNorm its backward partials (row sums of dx and dx * xhat) ride in the epilogue of those products
            // instead of a k_bn_bwd_partials launch (narrow layers, no dropout mask in the way).
            bool hook = l >= 1 && bn && !knobs().no_rowpart_hook;
            for (int gi = 0; gi < li.G && hook; ++gi) {
                const dp_stack_cfg* st = gi == 0 ? li.e : li.a;
                hook = gemm_rowpart_ok(st->dims[l]) && !drop_mask(li, io, gi, l);
            }
            float* masked_dst[2] = {nullptr, nullptr};
            const float* masked_m[2] = {nullptr, nullptr};
            int masked_ld[2] = {0, 0}, masked_w[2] = {0, 0};
            for (int gi = 0; gi < li.G; ++gi) {
                const dp_stack_cfg* st = gi == 0 ? li.e : li.a;
                const int din = st->dims[l], dout = st->dims[l + 1];
                const float* xin;
                int ldin;
                if (l == 0) {
                    xin = gi == 0 ? io.x0e : io.x0a;
                    ldin = din;
                } else {
                    xin = gi == 0 ? lv.Ze + li.coff_e[l - 1] : lv.Za + li.coff_a[l - 1];
                    ldin = gi == 0 ? li.D : li.Da;
                }
                const float* m = drop_mask(li, io, gi, l);
                if (m) {                                   // the GraphConv saw dropout(x): recompute it
                    mask_mul(q, xin, ldin, m, io.xm[gi], (long)B * n, din);
                    xin = io.xm[gi];
                    ldin = din;
                }
                // dW slab[b] = x_in[b]^T G[b]
                d[nd++] = GemmDesc{xin, Gj + g.c0[gi], slabs + st->w_off[l], nullptr, din, dout, n, ldin, ct, dout,
                                   (long)n * ldin, (long)n * ct, gstride, true, false, 1.f, 0.f, 0, slab_stride, 0};
                // gradient w.r.t. the layer input: G W^T (accumulated into the concat-gradient slice)
                float* dxin = nullptr;
                int lddx = 0;
                if (l > 0) {
                    dxin = gi == 0 ? gr.dZe + li.coff_e[l - 1] : gr.dZa + li.coff_a[l - 1];
                    lddx = gi == 0 ? li.D : li.Da;
                } else if (gr.dX0 && gi == 0) {
                    dxin = gr.dX0;
                    lddx = din;
                }
                if (dxin && m) {
                    // through the mask: G W^T into scratch (plain store, whole K per workgroup), masked add below
                    masked_dst[gi] = dxin;
                    masked_ld[gi] = lddx;
                    masked_m[gi] = m;
                    masked_w[gi] = din;
                    d[nd++] = GemmDesc{Gj + g.c0[gi], PW(params, st->w_off[l]), dxm[gi], nullptr, n, din, dout, ct, dout,
                                       din, (long)n * ct, 0, (long)n * din, false, true, 1.f, 0.f, 0, 0, 0, nullptr, 0, 0,
                                       0, 1};
                } else if (dxin && hook) {
                    // whole K in one workgroup, plain read-modify-write of the dZ slice, partials of the final values
                    GemmDesc h{Gj + g.c0[gi], PW(params, st->w_off[l]), dxin, nullptr, n, din, dout, ct, dout,
                               lddx, (long)n * ct, 0, (long)n * lddx, false, true, 1.f, 1.f, 0, 0, 0};
                    h.nosplit = 1;
                    h.rp_xhat = gi == 0 ? lv.Ze + li.coff_e[l - 1] : lv.Za + li.coff_a[l - 1];
                    h.rp_ldx = gi == 0 ? li.D : li.Da;
                    h.rp_part = part;
                    h.rp_G = li.G;
                    h.rp_g = gi;
                    d[nd++] = h;
                } else if (dxin)   // accumulates into dZ: atomic so split-K ranges (only > 1 for wide layers) cannot race
                    d[nd++] = GemmDesc{Gj + g.c0[gi], PW(params, st->w_off[l]), dxin, nullptr, n, din, dout, ct, dout,
                                       lddx, (long)n * ct, 0, (long)n * lddx, false, true, 1.f, 1.f, 0, 0,
                                       ks_level > 1 ? 1 : 0};
            }
            bgemm_group(q, d, nd, B, ks_level);
            part_ready = hook;
            for (int gi = 0; gi < li.G; ++gi)
                if (masked_dst[gi])
                    mask_axpy(q, masked_dst[gi], masked_ld[gi], dxm[gi], masked_m[gi], (long)B * n, masked_w[gi]);
            // both stacks of a pooled level read the same input X_j: the assign stack's share is added after
            if (l == 0 && gr.dX0 && li.G == 2) {
                const dp_stack_cfg* st = li.a;
                bgemm(q, Gj + g.c0[1], PW(params, st->w_off[0]), gr.dX0, nullptr, B, n, st->dims[0], st->dims[1], ct,
                      st->dims[1], st->dims[0], (long)n * ct, 0, (long)n * st->dims[0], false, true, 1.f, 1.f, 0);
            }
        }
        if (gr.dAdj) {
            // dA += dU P^T   (P = X W recomputed)
            transform(q, c, li, lv, io, params, l, Pj);
            bgemm(q, dUj, Pj, gr.dAdj, nullptr, B, n, n, ct, ct, ct, n, (long)n * ct, (long)n * ct, (long)n * n, false,
                  true, 1.f, 1.f, 0);
        }
    }
}

// floats of one masked-input scratch buffer (0 when no stack of the model has a dropout mask)
size_t dropout_scratch_floats(const dp_encoder_cfg& c) {
    size_t mx = 0;
    for (int j = 0; j <= c.num_pooling; ++j) {
        const LevelInfo li = level_info(c, j);
        for (int gi = 0; gi < li.G; ++gi) {
            const dp_stack_cfg* st = gi == 0 ? li.e : li.a;
            for (int l = 0; l < li.L; ++l)
                if (st->drop_off[l] >= 0) {
                    const size_t f = (size_t)c.B * li.n * st->dims[l];
                    mx = f > mx ? f : mx;
                }
        }
    }
    return mx;
}

struct Scratch {
    float* xm[2];
    float *Pj, *Uj, *part, *part_b, *logits, *lvl_part, *part_all;
    float *xpart, *mpart;    // persistent level-0 kernel: partial pooled products / max-readout partials,
    float* l0_part;          // BatchNorm partials per layer,
    unsigned short* l0_vs;   // split operands per pass (write-once exchange regions, dp_level0.hip)
    unsigned short* vs;      // 3-plane bf16 split of the current V operand (level 0, packed adjacency)
};
size_t vs_elems(const dp_encoder_cfg& c) {
    if (!adj_pack_supported(c.N, 1)) return 64;
    const LevelInfo li = level_info(c, 0);
    int cm = li.cmax > li.K ? li.cmax : li.K;
    if (cm > 320) cm = 320;     // widest operand the packed kernels take (dp_agg.hip AGGW_MAX_C)
    return split3_elems(c.B, c.N, cm) + 64;
}

// pred_model (+ the last level's readout when that level is a pooled, unmasked one) as one launch
HeadArgs head_args(const dp_encoder_cfg& c, const SaveLayout& sv, const float* params, float* ypred) {
    HeadArgs h{};
    h.params = params;
    h.n_pred = c.n_pred;
    h.B = c.B;
    for (int i = 0; i <= c.n_pred; ++i) h.dims[i] = c.pred_dims[i];
    for (int i = 0; i < c.n_pred; ++i) {
        h.w_off[i] = c.pred_w_off[i];
        h.b_off[i] = c.pred_b_off[i];
        h.hid[i] = sv.hid[i];
    }
    h.hid[c.n_pred] = ypred;
    return h;
}
bool head_usable(const dp_encoder_cfg& c) {
    if (c.readout != 0) return false;
    if (knobs().no_head_fusion) return false;
    HeadArgs h{};
    h.n_pred = c.n_pred;
    h.B = c.B;
    for (int i = 0; i <= c.n_pred; ++i) h.dims[i] = c.pred_dims[i];
    return head_supported(h);
}

// Level 0 as the persistent kernel sees it (dp_level0.hip).  Everything level0_persistent_ok() looks at comes from the
// cfg, so the sizing walk and the call decide alike; pointers are filled in by the caller.
Level0Fwd level0_desc(const dp_encoder_cfg& c) {
    const LevelInfo li = level_info(c, 0);
    Level0Fwd f{};
    f.B = c.B; f.N = c.N; f.L = li.L; f.G = li.G;
    f.bn = (c.flags & DP_F_BN) ? 1 : 0;
    f.do_max = c.readout == 0 ? 1 : 0;
    f.mask_readout = c.mask_readout ? 1 : 0;
    for (int g = 0; g < li.G; ++g) {
        const dp_stack_cfg* st = g == 0 ? li.e : li.a;
        for (int l = 0; l <= li.L; ++l) f.st[g].dims[l] = st->dims[l];
        for (int l = 0; l < li.L; ++l) {
            f.st[g].w_off[l] = st->w_off[l];
            f.st[g].b_off[l] = st->b_off[l];
            f.coff[g][l] = g == 0 ? li.coff_e[l] : li.coff_a[l];
        }
    }
    f.ldz[0] = li.D;
    f.ldz[1] = li.Da;
    f.K = li.K;
    if (li.G == 2) {
        f.wp_off = c.assign_pred_w_off[0];
        f.bp_off = c.assign_pred_b_off[0];
    }
    f.rw = readout_width(c, li);
    f.zoff = (c.flags & DP_F_LAST_ONLY) ? li.coff_e[li.L - 1] : 0;
    f.ldfeat = c.pred_dims[0];
    f.featoff = 0;
    f.pk_ld = adj_pack_ld(c.N);
    return f;
}
Level0Bwd level0_bwd_desc(const dp_encoder_cfg& c) {
    const Level0Fwd w = level0_desc(c);
    Level0Bwd f{};
    f.B = w.B; f.N = w.N; f.L = w.L; f.G = w.G; f.bn = w.bn;
    f.st[0] = w.st[0]; f.st[1] = w.st[1];
    f.ldz[0] = w.ldz[0]; f.ldz[1] = w.ldz[1];
    for (int g = 0; g < 2; ++g)
        for (int l = 0; l < DP_MAX_LAYERS; ++l) f.coff[g][l] = w.coff[g][l];
    f.K = w.K; f.wp_off = w.wp_off; f.bp_off = w.bp_off;
    f.pk_ld = w.pk_ld;
    return f;
}
bool level0_persistent(const dp_encoder_cfg& c);
bool level0_bwd_persistent(const dp_encoder_cfg& c) {
    return level0_persistent(c) && !knobs().no_l0_persist_bwd && level0_bwd_persistent_ok(level0_bwd_desc(c));
}
bool level0_persistent(const dp_encoder_cfg& c) {
    if (bn_sync(c) || (c.flags & DP_F_ADD_SELF)) return false;
    const LevelInfo li = level_info(c, 0);
    for (int l = 0; l < li.L; ++l)
        if (li.e->drop_off[l] >= 0 || (li.a && li.a->drop_off[l] >= 0) || layer_agg_first(li, l)) return false;
    return level0_persistent_ok(level0_desc(c));
}

// The tagged-entry exchange regions (BatchNorm partials of the persistent level-0 kernels, forward and backward, and of
// the pooled-level kernels).  Their readers recognise an entry by its tag, so NOTHING else may ever be written there:
// they sit right behind the barrier block, at the same offsets in the forward and the backward walk, and are never
// handed out as scratch — they only ever hold zeros (the workspace's one-time fill) or entries of earlier launches.
struct ExchangeRegions {
    float *l0_fwd, *l0_bwd, *lvl;
};
ExchangeRegions alloc_exchange(Seq& q, const dp_encoder_cfg& c) {
    ExchangeRegions x{};
    x.l0_fwd = q.alloc<float>(level0_persistent(c) ? level0_part_floats(level0_desc(c)) : 4);
    x.l0_bwd = q.alloc<float>(level0_bwd_persistent(c) ? level0_bwd_part_floats(level0_bwd_desc(c)) : 4);
    x.lvl = q.alloc<float>(level_part_floats(c));
    return x;
}

// shared allocation walk for the forward (also used for sizing)
Scratch fwd_scratch(Seq& q, const dp_encoder_cfg& c) {
    size_t maxPU = 0, maxPart = 0, maxLog = 0;
    for (int j = 0; j <= c.num_pooling; ++j) {
        const LevelInfo li = level_info(c, j);
        const size_t rows = (size_t)c.B * li.n;
        if (rows * li.cmax > maxPU) maxPU = rows * li.cmax;
        if (rows * li.G * 2 > maxPart) maxPart = rows * li.G * 2;
        if (li.a && rows * li.K > maxLog) maxLog = rows * li.K;
    }
    Scratch s{};
    s.Pj = q.alloc<float>(maxPU);
    s.Uj = q.alloc<float>(maxPU);
    s.part = q.alloc<float>(maxPart);
    s.part_b = q.alloc<float>(maxPart);
    s.logits = q.alloc<float>(maxLog > 0 ? maxLog : 1);
    s.part_all = bn_sync(c) ? q.alloc<float>(maxPart * bn_world(c)) : nullptr;
    s.vs = q.alloc<unsigned short>(vs_elems(c));
    if (level0_persistent(c)) {
        const Level0Fwd f = level0_desc(c);
        s.xpart = q.alloc<float>(level0_xpart_floats(f));
        s.mpart = q.alloc<float>(level0_mpart_floats(f));
        s.l0_vs = q.alloc<unsigned short>(level0_vs_elems(f));
    }
    const size_t dsf = dropout_scratch_floats(c);
    s.xm[0] = dsf ? q.alloc<float>(dsf) : nullptr;
    s.xm[1] = dsf ? q.alloc<float>(dsf) : nullptr;
    return s;
}

}  // namespace

int encoder_forward(Seq& q, const dp_encoder_cfg& c, const float* params, const float* x, const float* adj,
                    const float* assign_x, const int* num_nodes, const float* dropout, float* ypred,
                    float* assign_out, void* save, int mode, long long* labels_out, const PackedAdj* given) {
    SaveLayout sv = layout_save(c, save);
    if (given) {
        // the level-0 adjacency arrives as the packed bf16 pair (dp_build_batch_packed) instead of fp32: only the
        // persistent level-0 kernels multiply straight from it (every other plan wants the fp32 rows somewhere)
        if (!level0_persistent(c) || !level0_bwd_persistent(c)) {
            set_error("the packed-adjacency entry needs the persistent level-0 plan (N >= 64, N %% 4 == 0, B * ceil(N / RB) "
                      "<= CUs, no sync-BN); pass the fp32 adjacency to dp_encoder_forward for this configuration");
            return q.err = DP_ERR_UNSUPPORTED;
        }
        sv.pkA = const_cast<unsigned short*>(given->A);
        sv.pkAt = const_cast<unsigned short*>(given->At);
    }
    // the persistent level-0 kernel's barrier block: first thing in the workspace in BOTH walks; zero when the workspace
    // is first used (diffpool_hip.h), self-cleaning afterwards (dp_level0.hip)
    int* l0_bar = q.alloc<int>(level0_bar_ints(c.B));
    q.seq_word = level0_seq_word(l0_bar);
    const ExchangeRegions xr = alloc_exchange(q, c);
    const BwdZero bz = alloc_bwd_zero(q, c);      // same offsets as in encoder_backward: next block of the workspace
    Scratch sc = fwd_scratch(q, c);
    sc.lvl_part = xr.lvl;
    sc.l0_part = xr.l0_fwd;
    if (q.err) return q.err;
    const int B = c.B, P = c.num_pooling;
    const int ldfeat = c.pred_dims[0];
    const bool train = (mode & DP_MODE_TRAIN) != 0;
    // one pass over the level-0 adjacency: bf16 copies of A and A^T + exactness flag (used by every later pass); it
    // runs behind the level's first transform GEMM (see PackJob) and, in a training forward, clears the backward
    // pass's accumulators on the side
    PackedAdj pk0{sv.pkA, sv.pkAt, sv.pk_ld, sv.pk_flag};
    const PackedAdj* pkp = sv.pkA ? &pk0 : nullptr;
    PackJob pack{adj, sv.pkA, sv.pkAt, sv.pk_flag, sv.pk_ld, train && !q.dry ? q.ws + bz.begin : nullptr,
                 train ? bz.end - bz.begin : 0};
    const bool l0_persist = level0_persistent(c);
    const bool pack_in_level = pkp && !l0_persist && !(level_is_small(B, level_info(c, 0)) && !dropout);
    if (l0_persist) {
        // handled below (one launch for the whole level, pack included)
    } else if (pkp && !pack_in_level) adj_pack(q, adj, sv.pkA, sv.pkAt, sv.pk_flag, B, c.N, sv.pk_ld, false, pack.zero_p,
                                        pack.zero_bytes);
    else if (!pkp && train && !q.dry) zero_fill(q, q.ws + bz.begin, bz.end - bz.begin);
    int featoff = 0;
    const bool fused_head = head_usable(c);
    HeadArgs head = head_args(c, sv, params, ypred);
    head.labels = labels_out;
    // grid-barrier tickets of a whole-level kernel: cleared in stream order by the softmax launch of the level below
    // (together with its split-K tickets); a small level 0 has no such launch in front of it and clears its own
    int* level_bar = nullptr;
    const int* poison[DP_MAX_LEVELS + 1];   // error words of the whole-level kernels' barrier blocks (read by the head launch)
    int n_poison = 0;
    for (int j = 0; j <= P; ++j) {
        const LevelInfo li = level_info(c, j);
        const LevelSave& lv = sv.lv[j];
        if (j == 0 && li.G == 1 && li.n <= 64) {
            level_bar = q.alloc<int>(64);
            if (q.ok()) zero_small(q, level_bar, 64 * sizeof(int));
        }
        LevelIO io{};
        io.x0e = j == 0 ? x : sv.lv[j - 1].Xn;
        io.x0a = j == 0 ? assign_x : sv.lv[j - 1].Xn;
        io.adj = j == 0 ? adj : sv.lv[j - 1].An;
        io.drop = dropout;
        io.xm[0] = sc.xm[0];
        io.xm[1] = sc.xm[1];
        io.pack = (j == 0 && pack_in_level) ? &pack : nullptr;
        if (j == 0 && l0_persist) {
            // the whole level in ONE persistent launch: pack, GraphConv stacks, readout, assign head, pooling products
            Level0Fwd f = level0_desc(c);
            f.train = train ? 1 : 0;
            f.A = adj;
            f.x0[0] = x;
            f.x0[1] = assign_x;
            f.num_nodes = num_nodes;
            f.params = params;
            for (int l = 0; l < li.L; ++l) {
                f.Y[l] = lv.layer[l].Y;
                f.invn[l] = lv.layer[l].invn;
                f.stats[l] = lv.layer[l].stats;
            }
            f.Z[0] = lv.Ze;
            f.Z[1] = lv.Za;
            f.S = lv.S; f.S2 = assign_out; f.Tt = lv.T; f.Xn = lv.Xn; f.An = lv.An;
            f.feat = sv.feat;
            f.argmax = lv.argmax;
            f.pkA = sv.pkA; f.pkAt = sv.pkAt; f.pk_flag = sv.pk_flag;
            f.vs = sc.l0_vs; f.part = sc.l0_part; f.xpart = sc.xpart; f.mpart = sc.mpart;
            f.bar = l0_bar;
            f.zero_p = train && !q.dry ? q.ws + bz.begin : nullptr;
            f.zero_bytes = train ? bz.end - bz.begin : 0;
            if (j < P) {
                level_bar = q.alloc<int>(64);             // the pooled level's grid-barrier block, cleared by this launch
                f.next_bar = level_bar;
            }
            level0_forward(q, f);
            if (!q.dry) poison[n_poison++] = level0_error_word(l0_bar);
            if (c.readout == 0) featoff += f.rw;
            else {
                mask_rows(q, lv.Ze, li.D, sv.Zm, li.D, num_nodes, B, li.n, li.D);
                set2set_fwd(q, sv.Zm, li.D, PW(params, c.s2s_off[0]), PW(params, c.s2s_off[1]), PW(params, c.s2s_off[2]),
                            PW(params, c.s2s_off[3]), PW(params, c.s2s_off[4]), PW(params, c.s2s_off[5]), sv.feat, B,
                            li.n, li.D, sv.s2s);
            }
            continue;
        }
        level_forward(q, c, li, lv, io, params, sc.Pj, sc.Uj, sc.part, sc.part_b, j == 0 ? pkp : nullptr, sc.vs,
                      sc.lvl_part, level_bar, sc.part_all);
        if (level_bar && !q.dry && !bn_sync(c) && level_is_fused(c, li, io, true)) poison[n_poison++] = level_bar + 1;
        level_bar = nullptr;
        const int* nn_j = (j == 0) ? num_nodes : nullptr;
        if (c.readout == 0) {
            const int rw = readout_width(c, li);
            const float* zsrc = (c.flags & DP_F_LAST_ONLY) ? lv.Ze + li.coff_e[li.L - 1] : lv.Ze;
            if (fused_head && j == P && j > 0) {          // pooled levels are unmasked: the head kernel reads it
                head.Z = zsrc; head.ldz = li.D; head.n = li.n; head.rw = rw; head.featoff = featoff;
                head.argmax = lv.argmax; head.lda = rw;
            } else {
                masked_max_fwd(q, zsrc, li.D, c.mask_readout ? nn_j : nullptr, sv.feat + featoff, ldfeat, lv.argmax,
                               rw, B, li.n, rw);
            }
            featoff += rw;
        } else {
            mask_rows(q, lv.Ze, li.D, sv.Zm, li.D, nn_j, B, li.n, li.D);
            set2set_fwd(q, sv.Zm, li.D, PW(params, c.s2s_off[0]), PW(params, c.s2s_off[1]), PW(params, c.s2s_off[2]),
                        PW(params, c.s2s_off[3]), PW(params, c.s2s_off[4]), PW(params, c.s2s_off[5]), sv.feat, B,
                        li.n, li.D, sv.s2s);
        }
        if (j < P) {
            const int K = li.K, n = li.n;
            // S = softmax(Za Wp^T + bp) * mask
            bgemm(q, lv.Za, PW(params, c.assign_pred_w_off[j]), sc.logits, PW(params, c.assign_pred_b_off[j]), B, n, K,
                  li.Da, li.Da, li.Da, K, (long)n * li.Da, 0, (long)n * K, false, true, 1.f, 0.f, 0);
            // X' = S^T Z ;  Tt = A^T S  (= (S^T A)^T, [n x K]) ;  A' = Tt^T S.   X' and A' contract over the node
            // index (K = n): split-K with float atomics into the zeroed outputs when the level is large.  The softmax
            // launch also zero-fills those outputs and writes the bf16 split of S for the packed A^T S pass.
            const int ksn = n >= 256 ? node_ksplit(c) : 1;
            const bool s_split = j == 0 && pkp && aggregate_packed_usable(io.adj, n, K);
            // (deterministic split-K: partial tiles + tickets, the last range sums in range order -- float atomics
            // here made X' and A' vary in the last bit from run to run, enough to flip a near-tied max-readout
            // winner downstream and route its gradient to another row; the softmax launch zero-fills the tickets)
            const size_t cnt_ints = ksn > 1 ? gemm_fix_counters(B, K, li.D) + gemm_fix_counters(B, K, K) : 0;
            const size_t cnt_pad = (cnt_ints + 63) & ~size_t(63);
            int* sync_blk = q.alloc<int>(64 + cnt_pad);          // [level_bar of level j + 1 | split-K tickets]
            int* fix_cnt = ksn > 1 && sync_blk ? sync_blk + 64 : nullptr;
            float* fix_part = ksn > 1 ? q.alloc<float>((size_t)B * ksn * K * (li.D + K)) : nullptr;
            softmax_mask_fwd(q, sc.logits, K, lv.S, K, nn_j, B, n, K, j == 0 ? assign_out : nullptr,
                             s_split ? sc.vs : nullptr, sync_blk, (64 + cnt_pad) * sizeof(int));
            level_bar = sync_blk;
            aggregate(q, io.adj, lv.S, K, lv.T, K, B, n, K, true, 0.f, j == 0 ? pkp : nullptr, sc.vs, s_split);
            {
                // both pooled outputs in ONE launch (X' does not need T, but a launch of its own costs more than
                // waiting for the adjacency pass)
                GemmDesc d[2] = {
                    {lv.S, lv.Ze, lv.Xn, nullptr, K, li.D, n, K, li.D, li.D, (long)n * K, (long)n * li.D, (long)K * li.D,
                     true, false, 1.f, 0.f, 0, 0, 0},
                    {lv.T, lv.S, lv.An, nullptr, K, K, n, K, K, K, (long)n * K, (long)n * K, (long)K * K, true, false, 1.f,
                     0.f, 0, 0, 0}};
                if (ksn > 1) {
                    d[0].fix_part = fix_part;
                    d[0].fix_cnt = fix_cnt;
                    d[1].fix_part = fix_part + (size_t)B * ksn * K * li.D;
                    d[1].fix_cnt = fix_cnt + gemm_fix_counters(B, K, li.D);
                }
                bgemm_group(q, d, 2, B, ksn);
            }
        }
    }
    // pred_model
    if (fused_head) {
        for (int i = 0; i < n_poison; ++i) head.poison[i] = poison[i];
        head.n_poison = n_poison;
        head_fwd(q, head);
        return q.err;
    }
    for (int i = 0; i < c.n_pred; ++i) {
        const bool lastl = i == c.n_pred - 1;
        float* out = lastl ? ypred : sv.hid[i + 1];
        bgemm(q, sv.hid[i], PW(params, c.pred_w_off[i]), out, PW(params, c.pred_b_off[i]), 1, B, c.pred_dims[i + 1],
              c.pred_dims[i], c.pred_dims[i], c.pred_dims[i], c.pred_dims[i + 1], 0, 0, 0, false, true, 1.f, 0.f,
              lastl ? 0 : 1);
    }
    if (labels_out) argmax_rows(q, ypred, c.pred_dims[c.n_pred], labels_out, B, c.pred_dims[c.n_pred]);
    return q.err;
}

int encoder_backward(Seq& q, const dp_encoder_cfg& c, const float* params, const float* x, const float* adj,
                     const float* assign_x, const int* num_nodes, const float* dropout, const float* d_ypred,
                     const float* d_assign, float* grads, const void* save, int prezeroed, const PackedAdj* given) {
    SaveLayout sv = layout_save(c, (void*)save);
    if (given) {
        if (!level0_persistent(c) || !level0_bwd_persistent(c)) {
            set_error("the packed-adjacency entry needs the persistent level-0 plan (see dp_encoder_forward_packed)");
            return q.err = DP_ERR_UNSUPPORTED;
        }
        sv.pkA = const_cast<unsigned short*>(given->A);
        sv.pkAt = const_cast<unsigned short*>(given->At);
    }
    const int B = c.B, P = c.num_pooling;
    // ---- workspace walk
    size_t maxPU = 0, maxPart = 0, maxSK = 0, maxMeans = 0;
    // (the forward's barrier block, then) zero-initialised gradient accumulators + slabs: ONE block (alloc_bwd_zero)
    int* l0_bar = q.alloc<int>(level0_bar_ints(c.B));
    q.seq_word = level0_seq_word(l0_bar);
    const ExchangeRegions xr = alloc_exchange(q, c);
    const BwdZero bz = alloc_bwd_zero(q, c);
    const bool l0_persist = level0_bwd_persistent(c);
    unsigned short* l0_vs = nullptr;
    float *l0_part = nullptr, *l0_gpart = nullptr;
    if (l0_persist) {
        const Level0Bwd f0 = level0_bwd_desc(c);
        l0_vs = q.alloc<unsigned short>(level0_bwd_vs_elems(f0));
        l0_part = xr.l0_bwd;
        l0_gpart = q.alloc<float>(level0_bwd_gpart_floats(f0));
    }
    LevelGrad gr[DP_MAX_LEVELS + 1]{};
    for (int j = 0; j <= P; ++j) gr[j] = bz.gr[j];
    const int KS = node_ksplit(c);
    float* slabs = bz.slabs;
    const size_t zero_begin = bz.begin, zero_end = bz.end;
    for (int j = 0; j <= P; ++j) {
        const LevelInfo li = level_info(c, j);
        const size_t rows = (size_t)B * li.n;
        if (rows * li.cmax > maxPU) maxPU = rows * li.cmax;
        if (rows * li.G * 2 > maxPart) maxPart = rows * li.G * 2;
        if ((size_t)li.n * li.G * 2 > maxMeans) maxMeans = (size_t)li.n * li.G * 2;
        if (li.a && rows * li.K > maxSK) maxSK = rows * li.K;
        gr[j].dZa = li.a ? q.alloc<float>(rows * li.Da) : nullptr;
    }
    float* Pj = q.alloc<float>(maxPU);
    float* dUj = q.alloc<float>(maxPU);
    float* Gj = q.alloc<float>(maxPU);
    float* part = q.alloc<float>(maxPart);
    float* part_b = q.alloc<float>(maxPart);
    float* lvl_part = xr.lvl;
    float* part_all = bn_sync(c) ? q.alloc<float>(maxPart * bn_world(c)) : nullptr;
    unsigned short* vs = q.alloc<unsigned short>(vs_elems(c));
    float* dS = q.alloc<float>(maxSK ? maxSK : 1);
    float* dlog = q.alloc<float>(maxSK ? maxSK : 1);
    float* V = q.alloc<float>(maxSK ? maxSK : 1);
    float* V2 = q.alloc<float>(maxSK ? maxSK : 1);
    const size_t dsf = dropout_scratch_floats(c);
    float* xm[2] = {dsf ? q.alloc<float>(dsf) : nullptr, dsf ? q.alloc<float>(dsf) : nullptr};
    float* dxm[2] = {dsf ? q.alloc<float>(dsf) : nullptr, dsf ? q.alloc<float>(dsf) : nullptr};
    float* dh[DP_MAX_PRED + 2];
    for (int i = 0; i < c.n_pred; ++i) dh[i] = q.alloc<float>((size_t)B * c.pred_dims[i]);
    dh[c.n_pred] = const_cast<float*>(d_ypred);   // read only
    float* dZm = nullptr;
    if (c.readout == 1) {
        const LevelInfo li = level_info(c, 0);
        dZm = q.alloc<float>((size_t)B * li.n * li.D);
    }
    if (q.err) return q.err;
    const long slab_stride = c.n_graph_params;
    PackedAdj pk0{sv.pkA, sv.pkAt, sv.pk_ld, sv.pk_flag};
    const PackedAdj* pkp = sv.pkA ? &pk0 : nullptr;

    // every entry of `grads` is written below (weights: slab reduce / direct GEMM; biases: reduce_bias /
    // column sums), so no memset of it is needed
    // (one launch: level gradient accumulators + the slabs, which receive atomic adds and leave split-K rows unused)
    // (a training forward with this workspace already cleared it on the side of its adjacency pack: prezeroed)
    if (!prezeroed) zero_fill(q, q.ws + zero_begin, zero_end - zero_begin);
    const bool fused_head = head_usable(c);
    if (fused_head) {
        // pred_model backward + the max-readout scatter of every level in one launch
        HeadBwdArgs hb{};
        hb.h = head_args(c, sv, params, nullptr);
        hb.d_ypred = d_ypred;
        hb.grads = grads;
        int featoff = 0;
        for (int j = 0; j <= P; ++j) {
            const LevelInfo li = level_info(c, j);
            const int rw = readout_width(c, li);
            HeadBwdArgs::Level& s = hb.lv[hb.n_levels++];
            s.dZ = (c.flags & DP_F_LAST_ONLY) ? gr[j].dZe + li.coff_e[li.L - 1] : gr[j].dZe;
            s.argmax = sv.lv[j].argmax;
            s.lda = rw; s.n = li.n; s.ldz = li.D; s.rw = rw; s.featoff = featoff;
            featoff += rw;
        }
        head_bwd(q, hb);
    } else {
        // ---- pred_model backward
        for (int i = c.n_pred - 1; i >= 0; --i) {
            const int din = c.pred_dims[i], dout = c.pred_dims[i + 1];
            {
                GemmDesc d[2] = {
                    {dh[i + 1], sv.hid[i], grads + c.pred_w_off[i], nullptr, dout, din, B, dout, din, din, 0, 0, 0, true,
                     false, 1.f, 0.f, 0},
                    {dh[i + 1], PW(params, c.pred_w_off[i]), dh[i], nullptr, B, din, dout, dout, din, din, 0, 0, 0, false,
                     false, 1.f, 0.f, 0}};
                bgemm_group(q, d, 2, 1);
            }
            if (c.pred_b_off[i] >= 0) colsum_batched(q, dh[i + 1], dout, 0, B, dout, grads + c.pred_b_off[i], 0, 1);
            if (i > 0) relu_bwd_inplace(q, dh[i], sv.hid[i], (long)B * din);
        }
        const float* dfeat = dh[0];
        const int ldfeat = c.pred_dims[0];
        // ---- readout backward -> dZe_j
        int featoff = 0;
        for (int j = 0; j <= P; ++j) {
            const LevelInfo li = level_info(c, j);
            if (c.readout == 0) {
                const int rw = readout_width(c, li);
                float* dz = (c.flags & DP_F_LAST_ONLY) ? gr[j].dZe + li.coff_e[li.L - 1] : gr[j].dZe;
                masked_max_bwd(q, dfeat + featoff, ldfeat, sv.lv[j].argmax, rw, dz, li.D, B, li.n, rw);
                featoff += rw;
            } else {
                set2set_bwd(q, sv.Zm, li.D, PW(params, c.s2s_off[0]), PW(params, c.s2s_off[1]), PW(params, c.s2s_off[2]),
                            PW(params, c.s2s_off[3]), PW(params, c.s2s_off[4]), PW(params, c.s2s_off[5]), sv.feat, dfeat,
                            dZm, li.D, grads + c.s2s_off[0], grads + c.s2s_off[1], grads + c.s2s_off[2],
                            grads + c.s2s_off[3], grads + c.s2s_off[4], grads + c.s2s_off[5], B, li.n, li.D, sv.s2s);
                mask_rows(q, dZm, li.D, gr[j].dZe, li.D, num_nodes, B, li.n, li.D);
            }
        }
    }
    // ---- levels, top down
    for (int j = P; j >= 0; --j) {
        const LevelInfo li = level_info(c, j);
        const LevelSave& lv = sv.lv[j];
        LevelIO io{};
        io.x0e = j == 0 ? x : sv.lv[j - 1].Xn;
        io.x0a = j == 0 ? assign_x : sv.lv[j - 1].Xn;
        io.adj = j == 0 ? adj : sv.lv[j - 1].An;
        io.drop = dropout;
        io.xm[0] = xm[0];
        io.xm[1] = xm[1];
        const int n = li.n;
        if (j == 0 && l0_persist) {
            // the whole level in ONE persistent launch (dp_level0.hip): pooling products, A V, softmax / assign head
            // backward, every GraphConv layer; each graph's parameter gradients arrive as its first slab row
            Level0Bwd f = level0_bwd_desc(c);
            f.A = adj;
            f.x0[0] = x;
            f.x0[1] = assign_x;
            f.params = params;
            for (int l = 0; l < li.L; ++l) {
                f.Y[l] = lv.layer[l].Y;
                f.invn[l] = lv.layer[l].invn;
                f.stats[l] = lv.layer[l].stats;
            }
            f.Z[0] = lv.Ze;
            f.Z[1] = lv.Za;
            f.S = lv.S; f.Tt = lv.T;
            f.dXn = P >= 1 ? gr[1].dX0 : nullptr;
            f.dAn = P >= 1 ? gr[1].dAdj : nullptr;
            f.d_assign = d_assign;
            f.dZe = gr[0].dZe;
            f.pkA = sv.pkA; f.pkAt = sv.pkAt; f.pk_flag = sv.pk_flag;
            f.slabs = slabs;
            f.slab_gstride = slab_stride * KS;
            f.vs = l0_vs; f.part = l0_part; f.gpart = l0_gpart;
            f.bar = l0_bar;
            level0_backward(q, f);
            continue;
        }
        if (j < P) {
            const int K = li.K, D = li.D;
            bool vsplit_used = false;
            const float* dS2 = nullptr;
            const float* dXn = gr[j + 1].dX0;
            const float* dAn = gr[j + 1].dAdj;
            {
                // X' = S^T Z:  dZ += S dX',  dS = Z dX'^T ;  A' = (S^T A) S:  V = S dA'^T,  W = S dA'
                unsigned short* vsplit =
                    (j == 0 && pkp && aggregate_packed_usable(io.adj, n, K)) ? vs : nullptr;
                vsplit_used = vsplit != nullptr;
                GemmDesc d[4] = {
                    {lv.S, dXn, gr[j].dZe, nullptr, n, D, K, K, D, D, (long)n * K, (long)K * D, (long)n * D, false, false,
                     1.f, 1.f, 0},
                    {lv.Ze, dXn, dS, nullptr, n, K, D, D, D, K, (long)n * D, (long)K * D, (long)n * K, false, true, 1.f,
                     0.f, 0},
                    {lv.S, dAn, V, nullptr, n, K, K, K, K, K, (long)n * K, (long)K * K, (long)n * K, false, true, 1.f, 0.f,
                     0, 0, 0, vsplit, (K + 15) / 16, ((n + 31) / 32) * 4, 0},
                    {lv.S, dAn, V2, nullptr, n, K, K, K, K, K, (long)n * K, (long)K * K, (long)n * K, false, false, 1.f,
                     0.f, 0}};
                if (!gr[j].dAdj) {
                    // no dA_j wanted (level 0): the fourth slot carries T dA' instead, into the idle V2 buffer; the
                    // softmax backward adds it to dS (one launch less on the critical path)
                    d[3] = GemmDesc{lv.T, dAn, V2, nullptr, n, K, K, K, K, K, (long)n * K, (long)K * K, (long)n * K, false,
                                    false, 1.f, 0.f, 0};
                    dS2 = V2;
                }
                bgemm_group(q, d, 4, B);
            }
            // dS += T dA' ;  dS += A V
            if (gr[j].dAdj)
                bgemm(q, lv.T, dAn, dS, nullptr, B, n, K, K, K, K, K, (long)n * K, (long)K * K, (long)n * K, false, false,
                      1.f, 1.f, 0);
            aggregate(q, io.adj, V, K, dS, K, B, n, K, false, 1.f, j == 0 ? pkp : nullptr, vs, vsplit_used);
            if (gr[j].dAdj)   // dA_j += (S dA') S^T
                bgemm(q, V2, lv.S, gr[j].dAdj, nullptr, B, n, n, K, K, K, n, (long)n * K, (long)n * K, (long)n * n, false,
                      true, 1.f, 1.f, 0);
            if (j == 0 && d_assign) axpy(q, dS, d_assign, 1.f, (long)B * n * K);
            softmax_mask_bwd(q, lv.S, K, dS, K, j == 0 ? num_nodes : nullptr, dlog, K, B, n, K,
                             c.assign_pred_b_off[j] >= 0 ? slabs + c.assign_pred_b_off[j] : nullptr, slab_stride * KS,
                             dS2);
            // assign_pred: logits = Za Wp^T + bp
            {
                // weight gradient (split-K over the node index, into the slabs) and dZa (whole K per workgroup) in
                // one launch
                const int ksn = n >= 256 ? KS : 1;
                GemmDesc d2[2] = {
                    {dlog, lv.Za, slabs + c.assign_pred_w_off[j], nullptr, K, li.Da, n, K, li.Da, li.Da, (long)n * K,
                     (long)n * li.Da, slab_stride * KS, true, false, 1.f, 0.f, 0, slab_stride, 0},
                    {dlog, PW(params, c.assign_pred_w_off[j]), gr[j].dZa, nullptr, n, li.Da, K, K, li.Da, li.Da,
                     (long)n * K, 0, (long)n * li.Da, false, false, 1.f, 0.f, 0, 0, 0, nullptr, 0, 0, 0, 1}};
                bgemm_group(q, d2, 2, B, ksn);
            }
        }
        level_backward(q, c, li, lv, io, params, gr[j], slabs, slab_stride, KS, Pj, dUj, Gj, part, part_b,
                       j == 0 ? pkp : nullptr, vs, dxm, lvl_part, bz.bar + 64 * j, part_all);
    }
    reduce_slabs(q, slabs, slab_stride, B * KS, grads, c.n_graph_params, 0);
    return q.err;
}

size_t encoder_save_bytes(const dp_encoder_cfg& c) { return layout_save(c, nullptr).total; }

// byte offset / float count of one saved activation of level `level` inside the save buffer (read-back for logging
// and tests; -1 when the level has no such tensor)
int encoder_save_locate(const dp_encoder_cfg& c, int level, int field, size_t* offset, size_t* count) {
    char* const base = reinterpret_cast<char*>(size_t(1) << 20);       // any non-null base: only differences are used
    SaveLayout sv = layout_save(c, base);
    const LevelInfo li = level_info(c, level);
    const LevelSave& lv = sv.lv[level];
    const size_t B = c.B;
    const void* p = nullptr;
    size_t cnt = 0;
    switch (field) {
        case DP_SAVE_ARGMAX: p = lv.argmax; cnt = B * readout_width(c, li); break;
        case DP_SAVE_S: p = lv.S; cnt = B * li.n * li.K; break;
        case DP_SAVE_XPOOL: p = lv.Xn; cnt = B * li.K * li.D; break;
        case DP_SAVE_ADJPOOL: p = lv.An; cnt = B * li.K * li.K; break;
        case DP_SAVE_Z: p = lv.Ze; cnt = B * li.n * li.D; break;
        case DP_SAVE_ZASSIGN: p = lv.Za; cnt = B * li.n * li.Da; break;
        default: set_error("dp_encoder_save_locate: unknown field %d", field); return DP_ERR_INVALID_ARG;
    }
    if (!p) {
        set_error("dp_encoder_save_locate: level %d has no field %d", level, field);
        return DP_ERR_INVALID_ARG;
    }
    *offset = (size_t)(static_cast<const char*>(p) - base);
    *count = cnt;
    return DP_OK;
}

int encoder_validate(const dp_encoder_cfg* c) { return validate(c); }

}  // namespace dp
