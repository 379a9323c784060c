// Host-side (no GPU, no HIP) construction of the kernel layout of a batch of protein pairs: what
// equidock_public_amd/graph.py:PackedGraph.build computes with ~60 numpy calls, as one pass of plain loops for the
// DataLoader workers (the reference spends this time in per-pair DGL heterograph
// construction and dgl.batch, src/utils/train_utils.py:61-108).  Results are bit-identical to the numpy path
// (tests/test_abi_and_graph.py::test_native_pack_equals_numpy_pack); that path stays as the fallback when this
// library has not been built.
//
//   edges of each type arrive in block-local node ids; ligand nodes are global ids [0, n_lig), receptor nodes
//   [n_lig, n_lig + n_rec); edges are stably sorted by destination if they are not already; he rows follow.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

extern "C" {

enum { EQDH_OK = 0, EQDH_ERR_DEGREE = 1, EQDH_ERR_RANGE = 2, EQDH_ERR_SPACE = 3 };

struct EqdHostPackIn {
    int32_t n_pairs, n_lig, n_rec;
    const int64_t* lig_counts;   // [n_pairs]
    const int64_t* rec_counts;   // [n_pairs]
    int64_t e_ll, e_rr;
    const int32_t *src_ll, *dst_ll, *src_rr, *dst_rr;   // block-local ids
    const float *he_ll, *he_rr;                         // [e][27]
    int32_t tile_edges, tile_nodes, att_block;
    // optional (eqd_host_collate_pack): the he rows of each edge type as per-pair blocks instead of one array
    // (he_ll / he_rr are ignored when the tables are given); he_*_cum[p] = first edge of pair p, [n_pairs] = total
    const float* const* he_ll_parts;
    const float* const* he_rr_parts;
    const int64_t* he_ll_cum;
    const int64_t* he_rr_cum;
    int32_t n_threads;   // worker threads for the he pass (0 / 1 = caller's thread only)
};
struct EqdHostPackOut {
    // caller-allocated: int32 [n_pairs+1] x2, [E] x3 (src, dst, csc_eid), [N+1] x2, [N+2] tile_node, [items_cap][4],
    // [2 n_pairs + 1] seg_off; int64 [E] edge_perm; float [E][27] he; uint16 [max(E,1)][32] he_bf16
    int32_t *lig_off, *rec_off, *src, *dst, *rowptr, *csc_ptr, *csc_eid, *tile_node, *att_items, *seg_off;
    int64_t* edge_perm;
    float* he;
    uint16_t* he_bf16;
    int32_t items_cap;
    int32_t n_tiles, n_att_items, max_seg, max_degree;   // results
};

static inline uint16_t f2bf_rne(float f) {      // round to nearest even, like torch's float -> bfloat16
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);    // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

int eqd_host_pack(const EqdHostPackIn* in, EqdHostPackOut* out) {
    const int B = in->n_pairs, nl = in->n_lig, nr = in->n_rec, n = nl + nr;
    const int64_t E = in->e_ll + in->e_rr;
    // per-pair offsets and segments (B ligand segments, then B receptor segments, global ids)
    out->lig_off[0] = out->rec_off[0] = 0;
    int max_seg = 0;
    for (int b = 0; b < B; ++b) {
        out->lig_off[b + 1] = out->lig_off[b] + (int32_t)in->lig_counts[b];
        out->rec_off[b + 1] = out->rec_off[b] + (int32_t)in->rec_counts[b];
        max_seg = std::max<int>(max_seg, (int)std::max(in->lig_counts[b], in->rec_counts[b]));
    }
    if (out->lig_off[B] != nl || out->rec_off[B] != nr) return EQDH_ERR_RANGE;
    for (int b = 0; b < B; ++b) {
        out->seg_off[b] = out->lig_off[b];
        out->seg_off[B + b] = nl + out->rec_off[b];
    }
    out->seg_off[2 * B] = nl + nr;
    out->max_seg = max_seg;

    // edges: per type a stable sort by destination (identity when already sorted), then global ids
    int64_t eoff = 0;
    const int32_t* srcs[2] = {in->src_ll, in->src_rr};
    const int32_t* dsts[2] = {in->dst_ll, in->dst_rr};
    const int64_t es[2] = {in->e_ll, in->e_rr};
    const int bases[2] = {0, nl}, counts[2] = {nl, nr};
    std::vector<int64_t> perm;
    for (int t = 0; t < 2; ++t) {
        const int64_t e = es[t];
        const int32_t *s = srcs[t], *d = dsts[t];
        bool sorted = true;
        for (int64_t i = 0; i < e; ++i) {
            if (s[i] < 0 || d[i] < 0 || s[i] >= counts[t] || d[i] >= counts[t]) return EQDH_ERR_RANGE;
            if (i && d[i] < d[i - 1]) sorted = false;
        }
        perm.resize((size_t)e);
        std::iota(perm.begin(), perm.end(), (int64_t)0);
        if (!sorted) std::stable_sort(perm.begin(), perm.end(), [d](int64_t a, int64_t b) { return d[a] < d[b]; });
        for (int64_t i = 0; i < e; ++i) {
            out->src[eoff + i] = s[perm[(size_t)i]] + bases[t];
            out->dst[eoff + i] = d[perm[(size_t)i]] + bases[t];
            out->edge_perm[eoff + i] = perm[(size_t)i] + eoff;
        }
        eoff += e;
    }
    // CSR by destination, CSC (stable by source) as edge ids
    std::vector<int64_t> deg((size_t)n + 1, 0), sdeg((size_t)n + 1, 0);
    for (int64_t i = 0; i < E; ++i) {
        ++deg[(size_t)out->dst[i]];
        ++sdeg[(size_t)out->src[i]];
    }
    int maxdeg = 0;
    out->rowptr[0] = out->csc_ptr[0] = 0;
    for (int i = 0; i < n; ++i) {
        maxdeg = std::max<int>(maxdeg, (int)deg[(size_t)i]);
        out->rowptr[i + 1] = out->rowptr[i] + (int32_t)deg[(size_t)i];
        out->csc_ptr[i + 1] = out->csc_ptr[i] + (int32_t)sdeg[(size_t)i];
    }
    out->max_degree = maxdeg;
    if (E && maxdeg > in->tile_edges) return EQDH_ERR_DEGREE;
    {
        std::vector<int32_t> cur(out->csc_ptr, out->csc_ptr + n);
        for (int64_t i = 0; i < E; ++i) out->csc_eid[cur[(size_t)out->src[i]]++] = (int32_t)i;
    }
    // node-aligned edge tiles: greedy, a tile takes nodes while it stays within tile_edges edges and tile_nodes nodes
    int nt = 0;
    out->tile_node[0] = 0;
    if (n) {
        int i = 0;
        while (i < n) {
            int j = i;
            const int64_t e0 = out->rowptr[i];
            while (j < n && j - i < in->tile_nodes && (int64_t)out->rowptr[j + 1] - e0 <= in->tile_edges) ++j;
            if (j == i) j = i + 1;
            i = j;
            out->tile_node[++nt] = i;
        }
    } else {
        out->tile_node[1] = 0;
        nt = 1;
    }
    out->n_tiles = nt;
    // attention work list: blocks of att_block nodes x the partner's node range; biggest partner first (stable)
    struct Item { int32_t v[4]; };
    std::vector<Item> items;
    for (int b = 0; b < B; ++b) {
        const int32_t l0 = out->lig_off[b], l1 = out->lig_off[b + 1];
        const int32_t r0 = nl + out->rec_off[b], r1 = nl + out->rec_off[b + 1];
        const int32_t quad[2][4] = {{l0, l1, r0, r1}, {r0, r1, l0, l1}};
        for (int q = 0; q < 2; ++q)
            for (int32_t s0 = quad[q][0]; s0 < quad[q][1]; s0 += in->att_block)
                items.push_back(Item{{s0, std::min<int32_t>(s0 + in->att_block, quad[q][1]), quad[q][2], quad[q][3]}});
    }
    // XCD-aware order (graph.py: _xcd_interleave, same algorithm bit for bit): the groups (blocks of one (pair, direction),
    // consecutive in `items`) sorted by decreasing partner size and dealt in snake order into 8 lists, the lists
    // concatenated and cut into 8 queues of equal cost, each queue with its biggest partners first; slot 8 k + c = k-th
    // item of queue c, so the blocks that stream the same partner rows share one XCD's L2 (workgroup b runs on XCD b % 8)
    constexpr int XCD = 8;
    {
        struct Group { size_t first, count; int32_t partner; };
        std::vector<Group> groups;
        for (size_t i = 0; i < items.size(); ++i) {
            if (groups.empty() || items[i].v[2] != items[groups.back().first].v[2] || items[i].v[3] != items[groups.back().first].v[3])
                groups.push_back(Group{i, 0, items[i].v[3] - items[i].v[2]});
            ++groups.back().count;
        }
        std::stable_sort(groups.begin(), groups.end(), [](const Group& a, const Group& b) { return a.partner > b.partner; });
        std::vector<std::vector<size_t>> lists(XCD);
        for (size_t i = 0; i < groups.size(); ++i) {
            const size_t r = i % (2 * XCD);
            lists[r < (size_t)XCD ? r : 2 * XCD - 1 - r].push_back(i);
        }
        std::vector<Item> seq;
        seq.reserve(items.size());
        for (const auto& lst : lists)
            for (size_t gi : lst)
                for (size_t k = 0; k < groups[gi].count; ++k) seq.push_back(items[groups[gi].first + k]);
        items.swap(seq);
    }
    int64_t total = 0;
    for (const Item& it : items) total += it.v[3] - it.v[2];
    std::vector<int32_t> qlen(XCD, 0), slot(items.size());
    {
        std::vector<std::vector<Item>> queues(XCD);
        int c = 0;
        int64_t acc = 0;
        for (size_t i = 0; i < items.size(); ++i) {
            queues[c].push_back(items[i]);
            acc += items[i].v[3] - items[i].v[2];
            if (c < XCD - 1 && acc * XCD >= total * (c + 1)) ++c;
        }
        size_t i = 0;
        for (int q = 0; q < XCD; ++q) {
            std::stable_sort(queues[q].begin(), queues[q].end(),
                             [](const Item& a, const Item& b) { return (a.v[3] - a.v[2]) > (b.v[3] - b.v[2]); });
            for (const Item& it : queues[q]) {
                items[i] = it;
                slot[i++] = XCD * qlen[q]++ + q;
            }
        }
    }
    const int32_t depth = *std::max_element(qlen.begin(), qlen.end());
    if ((int64_t)depth * XCD > out->items_cap) return EQDH_ERR_SPACE;
    std::memset(out->att_items, 0, (size_t)depth * XCD * 16);
    for (size_t i = 0; i < items.size(); ++i) std::memcpy(out->att_items + 4 * (size_t)slot[i], items[i].v, 16);
    out->n_att_items = depth * XCD;
    // edge features in sorted order + bf16 copy (32 columns per edge, 27 used): the only pass over the big array
    // (41 MB at 64 x (300, 300)), split over worker threads by edge ranges
    const float* hes[2] = {in->he_ll, in->he_rr};
    const float* const* parts[2] = {in->he_ll_parts, in->he_rr_parts};
    const int64_t* cums[2] = {in->he_ll_cum, in->he_rr_cum};
    if (E == 0) std::memset(out->he_bf16, 0, 64);
    auto he_range = [&](int t, int64_t base, int64_t i0, int64_t i1) {
        int part = 0;
        for (int64_t i = i0; i < i1; ++i) {
            const int64_t src_row = out->edge_perm[base + i] - base;
            const float* row;
            if (parts[t]) {
                const int64_t* cum = cums[t];
                if (src_row < cum[part] || src_row >= cum[part + 1])
                    part = (int)(std::upper_bound(cum, cum + B + 1, src_row) - cum) - 1;
                row = parts[t][part] + (size_t)(src_row - cum[part]) * 27;
            } else {
                row = hes[t] + (size_t)src_row * 27;
            }
            float* o = out->he + (size_t)(base + i) * 27;
            uint16_t* ob = out->he_bf16 + (size_t)(base + i) * 32;
            for (int c = 0; c < 27; ++c) {
                o[c] = row[c];
                ob[c] = f2bf_rne(row[c]);
            }
            for (int c = 27; c < 32; ++c) ob[c] = 0;
        }
    };
    const int nth = std::max(1, std::min<int>(in->n_threads, 16));
    eoff = 0;
    for (int t = 0; t < 2; ++t) {
        const int64_t e = es[t];
        if (nth == 1 || e < 20000) {
            he_range(t, eoff, 0, e);
        } else {
            std::vector<std::thread> th;
            const int64_t chunk = (e + nth - 1) / nth;
            for (int k = 1; k < nth; ++k)
                th.emplace_back(he_range, t, eoff, std::min(e, k * chunk), std::min(e, (k + 1) * chunk));
            he_range(t, eoff, 0, std::min(e, chunk));
            for (auto& x : th) x.join();
        }
        eoff += e;
    }
    return EQDH_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Collate + pack in one call: what the reference does with one dgl.heterograph per pair + dgl.batch in its
// DataLoader's collate function (src/utils/train_utils.py:61-100), straight from the per-pair arrays into the batch's
// node arrays and the kernel layout - no per-pair tensor objects, no concatenated intermediate of the edge features.
// ---------------------------------------------------------------------------------------------------------------
struct EqdHostCollateIn {
    int32_t n_pairs;
    const int64_t *nl, *nr, *el, *er;                      // [n_pairs] node / edge counts
    const float* const *lig_x, *const *lig_new_x, *const *lig_res, *const *lig_mu;   // per pair [n][3], [n][3], [n], [n][5]
    const float* const *rec_x, *const *rec_res, *const *rec_mu;
    const float* const *he_ll, *const *he_rr;                // per pair [e][27]
    const int32_t* const *src_ll, *const *dst_ll, *const *src_rr, *const *dst_rr;    // per pair, pair-local node ids
    int32_t tile_edges, tile_nodes, att_block, n_threads;
};
struct EqdHostCollateOut {
    // caller-allocated batch-level node data: float [nl][3] lig_x, [nl][3] lig_new_x, [nl] lig_res, [nl][5] lig_mu,
    // [nr][3] rec_x, [nr] rec_res, [nr][5] rec_mu, [nl + nr][5] mu (both types), [nl + nr][3] x0 (new_x rows, then x rows);
    // int32 [nl + nr] res_id; int32 block-local endpoints with the batch's node offsets: [el_tot] x2, [er_tot] x2
    float *lig_x, *lig_new_x, *lig_res, *lig_mu, *rec_x, *rec_res, *rec_mu, *mu, *x0;
    int32_t *res_id, *src_ll, *dst_ll, *src_rr, *dst_rr;
    int32_t bad_res;   // result: 1 if a residue id is outside 0..20, 2 if a mu_r_norm entry is <= 0
};

int eqd_host_collate_pack(const EqdHostCollateIn* ci, EqdHostCollateOut* co, EqdHostPackOut* po) {
    const int B = ci->n_pairs;
    std::vector<int64_t> lc(B), rc(B), cum_ll(B + 1, 0), cum_rr(B + 1, 0);
    int64_t nl = 0, nr = 0;
    for (int b = 0; b < B; ++b) {
        lc[b] = ci->nl[b];
        rc[b] = ci->nr[b];
        nl += lc[b];
        nr += rc[b];
        cum_ll[b + 1] = cum_ll[b] + ci->el[b];
        cum_rr[b + 1] = cum_rr[b] + ci->er[b];
    }
    co->bad_res = 0;
    int64_t lo = 0, ro = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t a = lc[b], c = rc[b];
        std::memcpy(co->lig_x + lo * 3, ci->lig_x[b], (size_t)a * 12);
        std::memcpy(co->lig_new_x + lo * 3, ci->lig_new_x[b], (size_t)a * 12);
        std::memcpy(co->x0 + lo * 3, ci->lig_new_x[b], (size_t)a * 12);
        std::memcpy(co->lig_res + lo, ci->lig_res[b], (size_t)a * 4);
        std::memcpy(co->lig_mu + lo * 5, ci->lig_mu[b], (size_t)a * 20);
        std::memcpy(co->mu + lo * 5, ci->lig_mu[b], (size_t)a * 20);
        std::memcpy(co->rec_x + ro * 3, ci->rec_x[b], (size_t)c * 12);
        std::memcpy(co->x0 + (nl + ro) * 3, ci->rec_x[b], (size_t)c * 12);
        std::memcpy(co->rec_res + ro, ci->rec_res[b], (size_t)c * 4);
        std::memcpy(co->rec_mu + ro * 5, ci->rec_mu[b], (size_t)c * 20);
        std::memcpy(co->mu + (nl + ro) * 5, ci->rec_mu[b], (size_t)c * 20);
        for (int64_t i = 0; i < a; ++i) {
            const float r = ci->lig_res[b][i];
            co->res_id[lo + i] = (int32_t)r;
            if (!(r >= 0.f && r <= 20.f)) co->bad_res = 1;
        }
        for (int64_t i = 0; i < c; ++i) {
            const float r = ci->rec_res[b][i];
            co->res_id[nl + ro + i] = (int32_t)r;
            if (!(r >= 0.f && r <= 20.f)) co->bad_res = 1;
        }
        for (int64_t i = 0; i < ci->el[b]; ++i) {
            const int32_t s = ci->src_ll[b][i], d = ci->dst_ll[b][i];
            if (s < 0 || d < 0 || s >= a || d >= a) return EQDH_ERR_RANGE;
            co->src_ll[cum_ll[b] + i] = s + (int32_t)lo;
            co->dst_ll[cum_ll[b] + i] = d + (int32_t)lo;
        }
        for (int64_t i = 0; i < ci->er[b]; ++i) {
            const int32_t s = ci->src_rr[b][i], d = ci->dst_rr[b][i];
            if (s < 0 || d < 0 || s >= c || d >= c) return EQDH_ERR_RANGE;
            co->src_rr[cum_rr[b] + i] = s + (int32_t)ro;
            co->dst_rr[cum_rr[b] + i] = d + (int32_t)ro;
        }
        lo += a;
        ro += c;
    }
    if (!co->bad_res)
        for (int64_t i = 0; i < (nl + nr) * 5; ++i)
            if (!(co->mu[i] > 0.f)) {
                co->bad_res = 2;
                break;
            }
    EqdHostPackIn pi;
    std::memset(&pi, 0, sizeof(pi));
    pi.n_pairs = B; pi.n_lig = (int32_t)nl; pi.n_rec = (int32_t)nr;
    pi.lig_counts = lc.data(); pi.rec_counts = rc.data();
    pi.e_ll = cum_ll[B]; pi.e_rr = cum_rr[B];
    pi.src_ll = co->src_ll; pi.dst_ll = co->dst_ll; pi.src_rr = co->src_rr; pi.dst_rr = co->dst_rr;
    pi.tile_edges = ci->tile_edges; pi.tile_nodes = ci->tile_nodes; pi.att_block = ci->att_block;
    pi.he_ll_parts = ci->he_ll; pi.he_rr_parts = ci->he_rr;
    pi.he_ll_cum = cum_ll.data(); pi.he_rr_cum = cum_rr.data();
    pi.n_threads = ci->n_threads;
    return eqd_host_pack(&pi, po);
}

int eqd_host_pack_abi(void) { return 2; }

}  // extern "C"
