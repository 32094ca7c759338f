// tu_search.cpp (libhavoc_search.so) -- the residual-quadtree decisions of a picture's inter coding units as a BATCH client
// (tu_decision.hpp; VERDICT r2 next #2: "every TU of a CTU into one launch per primitive").
//
// An inter unit's transform-tree candidates depend on nothing but its own prediction and the (frozen) CABAC states RDOQ reads, so the
// whole picture's candidates -- both depths of every unit, a super-set of what the decision will look at -- go through ONE chain per
// transform size:   tu_forward -> rdoq -> tu_reconstruct (+ SSD) -> level_stats,   candidates reconstructing into private pieces
// (as the reference's ReconstructionCache pieces).  The decisions are then taken on the host by tu_decision.hpp from the downloaded
// flags / SSDs / level statistics (16 bytes per candidate; no coefficient or sample crosses the link), and the chosen candidates'
// levels are reconstructed once more, into the picture.
#include "batch_common.hpp"
#include "tu_decision.hpp"

using namespace havoc_search;

namespace {

#define RC(call) HAVOC_SEARCH_RC(call)

struct Candidate { int cu, depth, k; };      // which unit, which tree depth, which of the four blocks

struct LookupView
{
    const havoc_tu_outcome *zero, *one;      // of the unit being decided
    havoc_tu_outcome evaluate(int, int, int, int depth) { return depth ? one[k++ & 3] : *zero; }
    int k = 0;
};

} // namespace

extern "C" {

// d_src: source picture (src_origin = offset of sample (0, 0)); d_pred: the units' prediction, a plane of its own with sample (0, 0) at
// offset 0; d_rec: the reconstruction picture (rec_origin).  quant[log2 - 2]: parameters per transform size; d_states: CABAC snapshots.
// A unit of size 8 has 4x4 blocks at depth 1 (MinTbLog2SizeY = 2); sizes 8..64 with MaxTbLog2SizeY = 5 (a 64 unit's depth 0 is not a
// candidate: the tree is split by inference, Reconstruct.cpp:1296-1300 -- such units are rejected here).  out[i] per unit.
int havoc_search_rqt(havoc_mi355x_ctx *ctx, int S, int bitDepth, const void *d_src, int64_t src_origin, intptr_t src_stride, const void *d_pred,
                     intptr_t pred_stride, void *d_rec, int64_t rec_origin, intptr_t rec_stride, const uint8_t *d_states, const havoc_rqt_quant quant[4],
                     double lambda, double reciprocal_lambda, int sdh, const havoc_rqt_cu *cus, int n, havoc_rqt_result *out, havoc_rqt_stats *stats)
{
    if (!ctx || !d_src || !d_pred || !d_rec || !d_states || !quant || !cus || !out || n < 0 || (S != 1 && S != 2)) return HAVOC_MI355X_EINVAL;
    const double tStart = now();
    havoc_rqt_stats st;
    std::memset(&st, 0, sizeof(st));
    Arena arena(ctx);
    for (int i = 0; i < n; ++i)
        if (cus[i].log2_size < 3 || cus[i].log2_size > 5 || cus[i].x0 < 0 || cus[i].y0 < 0) return HAVOC_MI355X_EINVAL;

    // candidates by transform size
    std::vector<Candidate> bySize[4];
    for (int i = 0; i < n; ++i)
    {
        bySize[cus[i].log2_size - 2].push_back({i, 0, 0});
        for (int k = 0; k < 4; ++k) bySize[cus[i].log2_size - 3].push_back({i, 1, k});
    }
    struct Group
    {
        int log2, m;
        havoc_mi355x_tu_fused_job *hJobs;
        void *dJobs, *dCoef, *dLevel, *dPiece, *dWork;
        havoc_tu_outcome *res;              // host view of the results (pinned): filled from three device arrays below
        int32_t *hCbf, *hStats;
        uint32_t *hSsd;
        void *dCbf, *dSsd, *dStats, *dRj, *dSj;
    } groups[4];
    const double tGpu = now();
    for (int s = 0; s < 4; ++s)
    {
        Group &g = groups[s];
        g.log2 = s + 2;
        g.m = int(bySize[s].size());
        if (!g.m) continue;
        const int nn = 1 << g.log2, area = nn * nn;
        void *h, *hd;
        RC(arena.get(size_t(g.m) * sizeof(havoc_mi355x_tu_fused_job), &g.dJobs, &h, &hd));
        g.hJobs = static_cast<havoc_mi355x_tu_fused_job *>(h);
        void *dJobsDirect = hd;
        void *hRj, *hRjD, *hSj, *hSjD, *hx, *hxd;
        RC(arena.get(size_t(g.m) * sizeof(havoc_mi355x_rdoq_job), &g.dRj, &hRj, &hRjD));
        RC(arena.get(size_t(g.m) * 8, &g.dSj, &hSj, &hSjD));
        RC(arena.get(size_t(g.m) * area * 2, &g.dCoef, &hx));
        RC(arena.get(size_t(g.m) * area * 2, &g.dLevel, &hx));
        RC(arena.get(size_t(g.m) * area * S, &g.dPiece, &hx));
        RC(arena.get(havoc_mi355x_rdoq_workspace(g.m) + 64, &g.dWork, &hx));
        void *hc, *hs, *ht;
        RC(arena.get(size_t(g.m) * 4, &g.dCbf, &hc, &hxd));
        g.dCbf = hxd;
        RC(arena.get(size_t(g.m) * 4, &g.dSsd, &hs, &hxd));
        g.dSsd = hxd;
        RC(arena.get(size_t(g.m) * 8, &g.dStats, &ht, &hxd));
        g.dStats = hxd;
        g.hCbf = static_cast<int32_t *>(hc);
        g.hSsd = static_cast<uint32_t *>(hs);
        g.hStats = static_cast<int32_t *>(ht);
        havoc_mi355x_rdoq_job *rj = static_cast<havoc_mi355x_rdoq_job *>(hRj);
        int32_t *sj = static_cast<int32_t *>(hSj);
        int32_t lq, sf;
        havoc_mi355x_rdoq_lambda(lambda, quant[s].inv_scale, &lq, &sf);
        for (int j = 0; j < g.m; ++j)
        {
            const Candidate &c = bySize[s][j];
            const havoc_rqt_cu &cu = cus[c.cu];
            const int x = cu.x0 + (c.depth ? (c.k & 1) * nn : 0), y = cu.y0 + (c.depth ? (c.k >> 1) * nn : 0);
            g.hJobs[j] = {int32_t(size_t(j) * area), int32_t(src_origin + int64_t(y) * src_stride + x), int32_t(int64_t(y) * pred_stride + x), int32_t(size_t(j) * area)};
            std::memset(&rj[j], 0, sizeof(rj[j]));
            rj[j].dst_off = rj[j].src_off = int32_t(size_t(j) * area);
            rj[j].quant_scale = quant[s].quant_scale;
            rj[j].quant_shift = quant[s].quant_shift;
            rj[j].inv_scale = quant[s].inv_scale;
            rj[j].lambda_q16 = lq;
            rj[j].sdh_factor = sf;
            rj[j].ctx_index = cu.ctx_index;
            rj[j].sdh = uint8_t(sdh != 0);
            sj[2 * j] = int32_t(size_t(j) * area);
            sj[2 * j + 1] = area;
        }
        // the chain of this size: job tables are read from mapped host memory, the 16 bytes of results per candidate written to it
        const havoc_mi355x_tu_fused_job *dj = static_cast<const havoc_mi355x_tu_fused_job *>(dJobsDirect);
        RC(havoc_mi355x_tu_forward(ctx, S, bitDepth, 0, g.log2, static_cast<int16_t *>(g.dCoef), d_src, src_stride, d_pred, pred_stride, dj, g.m));
        RC(havoc_mi355x_rdoq(ctx, bitDepth, g.log2, static_cast<int16_t *>(g.dLevel), static_cast<const int16_t *>(g.dCoef), d_states,
                             static_cast<const havoc_mi355x_rdoq_job *>(hRjD), g.m, static_cast<int32_t *>(g.dCbf), g.dWork, havoc_mi355x_rdoq_workspace(g.m)));
        // candidates reconstruct into private pieces (n x n, stride n): jobs' rec_off = piece offset
        RC(havoc_mi355x_tu_reconstruct(ctx, S, bitDepth, 0, g.log2, quant[s].inv_scale, quant[s].inv_shift, g.dPiece, nn, d_pred, pred_stride, d_src, src_stride,
                                       static_cast<const int16_t *>(g.dLevel), dj, g.m, static_cast<uint32_t *>(g.dSsd)));
        RC(havoc_mi355x_level_stats(ctx, static_cast<const int16_t *>(g.dLevel), static_cast<const int32_t *>(hSjD), g.m, static_cast<int32_t *>(g.dStats)));
        st.launches += 4;
        st.candidates += g.m;
    }
    RC(havoc_mi355x_sync(ctx));
    st.seconds_gpu += now() - tGpu;

    // ---- decisions (tu_decision.hpp) from the downloaded outcomes
    const double tHost = now();
    std::vector<havoc_tu_outcome> outcome[4];
    std::vector<int> cursor(4, 0);
    for (int s = 0; s < 4; ++s)
    {
        outcome[s].resize(groups[s].m);
        for (int j = 0; j < groups[s].m; ++j)
            outcome[s][j] = {groups[s].hCbf[j], groups[s].hSsd[j], groups[s].hStats[2 * j], groups[s].hStats[2 * j + 1]};
    }
    Lambda rl;
    rl.set(reciprocal_lambda);
    std::vector<int> zeroAt(n), oneAt(n);
    for (int i = 0; i < n; ++i)      // candidates were appended unit by unit: depth 0 to its size's list, then four to the next smaller
    {
        const int s0 = cus[i].log2_size - 2, s1 = s0 - 1;
        zeroAt[i] = cursor[s0]++;
        oneAt[i] = cursor[s1];
        cursor[s1] += 4;
    }
    std::vector<havoc_mi355x_tu_fused_job> finalJobs[4];
    for (int i = 0; i < n; ++i)
    {
        const int s0 = cus[i].log2_size - 2, s1 = s0 - 1;
        LookupView view{&outcome[s0][zeroAt[i]], &outcome[s1][oneAt[i]]};
        out[i] = decideRqt(view, cus[i], rl);
        // the chosen candidates reconstruct into the picture
        if (out[i].depth == 0)
        {
            havoc_mi355x_tu_fused_job j = groups[s0].hJobs[zeroAt[i]];
            j.rec_off = int32_t(rec_origin + int64_t(cus[i].y0) * rec_stride + cus[i].x0);
            if (!out[i].tried_zero) j.coef_off = -1;      // no residual: marked, handled below
            finalJobs[s0].push_back(j);
        }
        else
            for (int k = 0; k < 4; ++k)
            {
                havoc_mi355x_tu_fused_job j = groups[s1].hJobs[oneAt[i] + k];
                const int nn = 1 << (s1 + 2);
                j.rec_off = int32_t(rec_origin + int64_t(cus[i].y0 + (k >> 1) * nn) * rec_stride + cus[i].x0 + (k & 1) * nn);
                finalJobs[s1].push_back(j);
            }
    }
    st.seconds_host += now() - tHost;
    const double tGpu2 = now();
    // units left without residual: reconstruction = prediction.  Their depth-1 candidates all quantised to zero, so reconstructing those
    // four (levels all zero) writes exactly the prediction: use them instead of a copy kernel
    for (int i = 0; i < n; ++i)
        if (out[i].depth == 0 && !out[i].tried_zero)
        {
            const int s1 = cus[i].log2_size - 3, nn = 1 << (s1 + 2);
            for (int k = 0; k < 4; ++k)
            {
                havoc_mi355x_tu_fused_job j = groups[s1].hJobs[oneAt[i] + k];
                j.rec_off = int32_t(rec_origin + int64_t(cus[i].y0 + (k >> 1) * nn) * rec_stride + cus[i].x0 + (k & 1) * nn);
                finalJobs[s1].push_back(j);
            }
        }
    for (int s = 0; s < 4; ++s)
    {
        std::vector<havoc_mi355x_tu_fused_job> &f = finalJobs[s];
        f.erase(std::remove_if(f.begin(), f.end(), [](const havoc_mi355x_tu_fused_job &j) { return j.coef_off < 0; }), f.end());
        if (f.empty()) continue;
        void *d, *h, *hd, *dSsd, *hx;
        RC(arena.get(f.size() * sizeof(f[0]), &d, &h, &hd));
        RC(arena.get(f.size() * 4, &dSsd, &hx));
        std::memcpy(h, f.data(), f.size() * sizeof(f[0]));
        RC(havoc_mi355x_tu_reconstruct(ctx, S, bitDepth, 0, s + 2, quant[s].inv_scale, quant[s].inv_shift, d_rec, rec_stride, d_pred, pred_stride, d_src, src_stride,
                                       static_cast<const int16_t *>(groups[s].dLevel), static_cast<const havoc_mi355x_tu_fused_job *>(hd), int(f.size()),
                                       static_cast<uint32_t *>(dSsd)));
        ++st.launches;
    }
    RC(havoc_mi355x_sync(ctx));
    st.seconds_gpu += now() - tGpu2;
    st.seconds_total = now() - tStart;
    if (stats) *stats = st;
    return 0;
}

// The picture's block structure after the decisions, as the 4x4 cells havoc_mi355x_derive_bs reads (host work, no launch): every unit one
// inter 2Nx2N prediction unit from list 0 (decoded picture `dpb_index0`) at the vector the motion field holds at its origin, its transform
// tree as decided by havoc_search_rqt (coded flag per block).  field: int16 [2 lists][height / 4][width / 4][x, y] (havoc_search_picture_uni);
// cells: [height / 4][width / 4], written completely.
int havoc_search_block_cells(int width, int height, int qp, int dpb_index0, const int16_t *field, const havoc_rqt_cu *cus, const havoc_rqt_result *dec, int n,
                             havoc_mi355x_cell *cells)
{
    if (!field || !cus || !dec || !cells || width <= 0 || height <= 0 || (width & 3) || (height & 3)) return HAVOC_MI355X_EINVAL;
    const int cw = width >> 2, ch = height >> 2;
    havoc_mi355x_cell blank;
    std::memset(&blank, 0, sizeof(blank));
    blank.dpb_index[0] = int8_t(dpb_index0);
    blank.dpb_index[1] = -1;
    blank.qp_y = int8_t(qp);
    blank.tu_log2 = 2;
    for (int i = 0; i < cw * ch; ++i) cells[i] = blank;
    for (int i = 0; i < n; ++i)
    {
        const havoc_rqt_cu &u = cus[i];
        const int x4 = u.x0 >> 2, y4 = u.y0 >> 2, n4 = (1 << u.log2_size) >> 2, half = n4 / 2;
        if (x4 < 0 || y4 < 0 || x4 + n4 > cw || y4 + n4 > ch) return HAVOC_MI355X_EINVAL;
        const int16_t *mv = field + (size_t(y4) * cw + x4) * 2;
        const bool split = dec[i].depth == 1, coded0 = dec[i].tried_zero == 1 && dec[i].zero.cbf != 0;
        for (int y = 0; y < n4; ++y)
            for (int x = 0; x < n4; ++x)
            {
                havoc_mi355x_cell &c = cells[size_t(y4 + y) * cw + x4 + x];
                c.mv[0][0] = mv[0];
                c.mv[0][1] = mv[1];
                const bool coded = split ? dec[i].one[(y >= half) * 2 + (x >= half)].cbf != 0 : coded0;
                c.flags = uint8_t((coded ? HAVOC_CELL_CODED : 0) | (x == 0 ? HAVOC_CELL_PU_LEFT : 0) | (y == 0 ? HAVOC_CELL_PU_TOP : 0));
                c.tu_log2 = uint8_t(u.log2_size - (split ? 1 : 0));
            }
    }
    return 0;
}

// The RD refinement of n intra partitions of ONE size (tu_decision.hpp: decideIntraRd): every candidate mode of every partition -- the
// refinement orders havoc_search_intra_modes returned -- through   intra prediction -> tu_forward -> rdoq -> tu_reconstruct (+ SSD) ->
// level_stats   in one chain, the champions picked on the host from 16 bytes per candidate, their reconstructions written to d_rec
// (block i = n x n samples at i * n * n, stride n).  jobs (HOST memory) / d_neighbours as for havoc_mi355x_intra_satd35: the partitions' source
// blocks and neighbour arrays; independent partitions -- the caller provides the neighbours (in the encoder they are the previous
// partition's reconstruction, Reconstruct.cpp:609-615: partitions that depend on each other go in successive calls).
int havoc_search_intra_rd(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2TrafoSize, const void *d_src, intptr_t src_stride, const void *d_neighbours,
                          const havoc_mi355x_intra_search_job *jobs, int n, const havoc_search_intra_result *order, const havoc_search_intra_ctx *ictx,
                          const int32_t *ctx_index, const uint8_t *d_states, const havoc_rqt_quant *quant, double lambda, double reciprocal_lambda, int sdh, void *d_rec,
                          havoc_intra_rd_result *out, havoc_rqt_stats *stats)
{
    if (!ctx || !d_src || !d_neighbours || !jobs || !order || !ictx || !ctx_index || !d_states || !quant || !d_rec || !out || n < 0 || (S != 1 && S != 2) ||
        log2TrafoSize < 2 || log2TrafoSize > 5)
        return HAVOC_MI355X_EINVAL;
    const double tStart = now();
    havoc_rqt_stats st;
    std::memset(&st, 0, sizeof(st));
    if (n == 0)
    {
        if (stats) *stats = st;
        return 0;
    }
    Arena arena(ctx);
    const int nn = 1 << log2TrafoSize, area = nn * nn, tr = log2TrafoSize == 2 ? 1 : 0;
    std::vector<int> first(n + 1, 0);
    for (int i = 0; i < n; ++i)
    {
        if (order[i].count < 0 || order[i].count > 35) return HAVOC_MI355X_EINVAL;
        first[i + 1] = first[i] + order[i].count;
    }
    const int m = first[n];
    void *dIj, *hIj, *vIj, *dTj, *hTj, *vTj, *dRj, *hRj, *vRj, *dSj, *hSj, *vSj, *dPred, *dPiece, *dCoef, *dLevel, *dWork, *hx, *dCbf, *hCbf, *vCbf, *dSsd, *hSsd, *vSsd,
        *dStats, *hStats, *vStats;
    RC(arena.get(size_t(m) * sizeof(havoc_mi355x_intra_job), &dIj, &hIj, &vIj));
    RC(arena.get(size_t(m) * sizeof(havoc_mi355x_tu_fused_job), &dTj, &hTj, &vTj));
    RC(arena.get(size_t(m) * sizeof(havoc_mi355x_rdoq_job), &dRj, &hRj, &vRj));
    RC(arena.get(size_t(m) * 8, &dSj, &hSj, &vSj));
    RC(arena.get(size_t(m) * area * S, &dPred, &hx));
    RC(arena.get(size_t(m) * area * S, &dPiece, &hx));
    RC(arena.get(size_t(m) * area * 2, &dCoef, &hx));
    RC(arena.get(size_t(m) * area * 2, &dLevel, &hx));
    RC(arena.get(havoc_mi355x_rdoq_workspace(m) + 64, &dWork, &hx));
    RC(arena.get(size_t(m) * 4, &dCbf, &hCbf, &vCbf));
    RC(arena.get(size_t(m) * 4, &dSsd, &hSsd, &vSsd));
    RC(arena.get(size_t(m) * 8, &dStats, &hStats, &vStats));
    havoc_mi355x_intra_job *ij = static_cast<havoc_mi355x_intra_job *>(hIj);
    havoc_mi355x_tu_fused_job *tj = static_cast<havoc_mi355x_tu_fused_job *>(hTj);
    havoc_mi355x_rdoq_job *rj = static_cast<havoc_mi355x_rdoq_job *>(hRj);
    int32_t *sj = static_cast<int32_t *>(hSj);
    int32_t lq, sf;
    havoc_mi355x_rdoq_lambda(lambda, quant->inv_scale, &lq, &sf);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < order[i].count; ++k)
        {
            const int c = first[i] + k, mode = order[i].order[k];
            const uint64_t mask = uint64_t(jobs[i].filt_lo) | (uint64_t(jobs[i].filt_hi) << 32);
            std::memset(&ij[c], 0, sizeof(ij[c]));
            ij[c].dst_off = c * area;
            ij[c].nb_off = ((mask >> mode) & 1) ? jobs[i].nbf_off : jobs[i].nb_off;
            ij[c].log2 = log2TrafoSize;
            ij[c].mode = mode;
            ij[c].edge = jobs[i].edge;
            tj[c] = {c * area, jobs[i].src_off, c * area, c * area};
            std::memset(&rj[c], 0, sizeof(rj[c]));
            rj[c].dst_off = rj[c].src_off = c * area;
            rj[c].quant_scale = quant->quant_scale;
            rj[c].quant_shift = quant->quant_shift;
            rj[c].inv_scale = quant->inv_scale;
            rj[c].lambda_q16 = lq;
            rj[c].sdh_factor = sf;
            rj[c].ctx_index = ctx_index[i];
            rj[c].scan_idx = uint8_t(intraScanIdx(log2TrafoSize, mode));
            rj[c].is_intra = 1;
            rj[c].sdh = uint8_t(sdh != 0);
            sj[2 * c] = c * area;
            sj[2 * c + 1] = area;
        }
    const double tGpu = now();
    const havoc_mi355x_tu_fused_job *dj = static_cast<const havoc_mi355x_tu_fused_job *>(vTj);
    RC(havoc_mi355x_intra(ctx, S, bitDepth, log2TrafoSize, dPred, nn, d_neighbours, static_cast<const havoc_mi355x_intra_job *>(vIj), m));
    RC(havoc_mi355x_tu_forward(ctx, S, bitDepth, tr, log2TrafoSize, static_cast<int16_t *>(dCoef), d_src, src_stride, dPred, nn, dj, m));
    RC(havoc_mi355x_rdoq(ctx, bitDepth, log2TrafoSize, static_cast<int16_t *>(dLevel), static_cast<const int16_t *>(dCoef), d_states,
                         static_cast<const havoc_mi355x_rdoq_job *>(vRj), m, static_cast<int32_t *>(vCbf), dWork, havoc_mi355x_rdoq_workspace(m)));
    RC(havoc_mi355x_tu_reconstruct(ctx, S, bitDepth, tr, log2TrafoSize, quant->inv_scale, quant->inv_shift, dPiece, nn, dPred, nn, d_src, src_stride,
                                   static_cast<const int16_t *>(dLevel), dj, m, static_cast<uint32_t *>(vSsd)));
    RC(havoc_mi355x_level_stats(ctx, static_cast<const int16_t *>(dLevel), static_cast<const int32_t *>(vSj), m, static_cast<int32_t *>(vStats)));
    RC(havoc_mi355x_sync(ctx));
    st.launches += 5;
    st.candidates = m;
    st.seconds_gpu += now() - tGpu;

    const double tHost = now();
    const int32_t *cbf = static_cast<const int32_t *>(hCbf), *stats32 = static_cast<const int32_t *>(hStats);
    const uint32_t *ssd = static_cast<const uint32_t *>(hSsd);
    Lambda rl;
    rl.set(reciprocal_lambda);
    struct Lookup
    {
        const int32_t *cbf, *stats;
        const uint32_t *ssd;
        int base;
        havoc_tu_outcome evaluate(int, int index) { const int c = base + index; return {cbf[c], ssd[c], stats[2 * c], stats[2 * c + 1]}; }
    };
    std::vector<havoc_mi355x_tu_fused_job> fin(n);
    for (int i = 0; i < n; ++i)
    {
        Lookup view{cbf, stats32, ssd, first[i]};
        out[i] = decideIntraRd(view, order[i], ictx[i], rl);
        const int c = first[i] + std::max(0, out[i].index);
        fin[i] = tj[c];
        fin[i].rec_off = i * area;
    }
    st.seconds_host += now() - tHost;
    // the champions' reconstructions (their levels are still on the device) into d_rec
    const double tGpu2 = now();
    void *dF, *hF, *vF, *dS2, *hS2;
    RC(arena.get(size_t(n) * sizeof(fin[0]), &dF, &hF, &vF));
    RC(arena.get(size_t(n) * 4, &dS2, &hS2));
    std::memcpy(hF, fin.data(), size_t(n) * sizeof(fin[0]));
    RC(havoc_mi355x_tu_reconstruct(ctx, S, bitDepth, tr, log2TrafoSize, quant->inv_scale, quant->inv_shift, d_rec, nn, dPred, nn, d_src, src_stride,
                                   static_cast<const int16_t *>(dLevel), static_cast<const havoc_mi355x_tu_fused_job *>(vF), n, static_cast<uint32_t *>(dS2)));
    RC(havoc_mi355x_sync(ctx));
    ++st.launches;
    st.seconds_gpu += now() - tGpu2;
    st.seconds_total = now() - tStart;
    if (stats) *stats = st;
    return 0;
}

// Both stages of a picture's intra partitions (all sizes) with the decisions between the launches taken ON THE DEVICE
// (havoc_mi355x_intra_order / _expand / _decide: csrc/kernels_decide.hip restates intraModeOrder and decideIntraRd lane per partition):
//   per size:  intra_satd35 -> intra_order                                      | one wait: how many candidates each size has (8 bytes per size)
//   per size:  intra_expand -> intra -> tu_forward -> rdoq -> tu_reconstruct -> level_stats -> intra_decide -> tu_reconstruct (champions)
// What crosses the link: the group descriptions down, 40 bytes per partition (the decision) up.  The same results as
// havoc_search_intra_modes + havoc_search_intra_rd, which take the decisions on the host and are what the tests compare this with.
int havoc_search_intra_device(havoc_mi355x_ctx *ctx, int S, int bitDepth, const void *d_src, intptr_t src_stride, const havoc_intra_group *groups, int ngroups,
                              const uint8_t *d_states, const havoc_rqt_quant quant[4], double reciprocal_sqrt_lambda, double lambda, double reciprocal_lambda, int sdh,
                              havoc_rqt_stats *stats)
{
    if (!ctx || !d_src || !groups || ngroups < 0 || ngroups > 16 || !d_states || !quant || (S != 1 && S != 2)) return HAVOC_MI355X_EINVAL;
    const double tStart = now();
    havoc_rqt_stats st;
    std::memset(&st, 0, sizeof(st));
    Arena arena(ctx);
    Lambda lsq, rl;
    lsq.set(reciprocal_sqrt_lambda);
    rl.set(reciprocal_lambda);
    struct Work { void *dCost, *dOrder, *dCount, *dSlot, *dTotal, *hTotal; } work[16];
    void *hx;
    for (int g = 0; g < ngroups; ++g)
    {
        const havoc_intra_group &G = groups[g];
        if (G.log2 < 2 || G.log2 > 5 || G.n < 0 || (G.n && (!G.d_neighbours || !G.d_jobs || !G.d_ictx || !G.d_ctx_index || !G.d_rec || !G.out))) return HAVOC_MI355X_EINVAL;
        if (!G.n) continue;
        Work &w = work[g];
        RC(arena.get(size_t(G.n) * 35 * 4, &w.dCost, &hx));
        RC(arena.get(size_t(G.n) * HAVOC_MI355X_INTRA_MAX_ORDER * 4, &w.dOrder, &hx));
        RC(arena.get(size_t(G.n) * 4, &w.dCount, &hx));
        RC(arena.get(size_t(G.n) * 4, &w.dSlot, &hx));
        RC(arena.get(8, &w.dTotal, &w.hTotal));
        RC(havoc_mi355x_intra_satd35(ctx, S, bitDepth, G.log2, d_src, src_stride, G.d_neighbours, static_cast<const havoc_mi355x_intra_search_job *>(G.d_jobs), G.n, static_cast<int32_t *>(w.dCost)));
        RC(havoc_mi355x_intra_order(ctx, static_cast<const int32_t *>(w.dCost), reinterpret_cast<const havoc_mi355x_intra_mpm *>(G.d_ictx), G.n, lsq.value,
                                    static_cast<int32_t *>(w.dOrder), static_cast<int32_t *>(w.dCount), static_cast<int32_t *>(w.dSlot), static_cast<int32_t *>(w.dTotal)));
        RC(havoc_mi355x_d2h_async(ctx, w.hTotal, w.dTotal, 8));
        st.launches += 2;
    }
    RC(havoc_mi355x_sync(ctx));
    void *hOut[16] = {nullptr};
    for (int g = 0; g < ngroups; ++g)
    {
        const havoc_intra_group &G = groups[g];
        if (!G.n) continue;
        const Work &w = work[g];
        const int32_t *total = static_cast<const int32_t *>(w.hTotal);
        if (total[1] || total[0] < 0 || total[0] > G.n * HAVOC_MI355X_INTRA_MAX_ORDER) return HAVOC_MI355X_EINVAL;      // max_refine + neighbour_modes beyond the slots
        const int m = total[0], nn = 1 << G.log2, area = nn * nn, tr = G.log2 == 2 ? 1 : 0;
        const havoc_rqt_quant &q = quant[G.log2 - 2];
        int32_t lq, sf;
        havoc_mi355x_rdoq_lambda(lambda, q.inv_scale, &lq, &sf);
        void *dIj, *dTj, *dRj, *dSj, *dOwner, *dPred, *dPiece, *dCoef, *dLevel, *dWork, *dCbf, *dSsd, *dStats, *dFin, *dSsd2, *dOut, *vOut;
        RC(arena.get(size_t(m) * sizeof(havoc_mi355x_intra_job), &dIj, &hx));
        RC(arena.get(size_t(m) * sizeof(havoc_mi355x_tu_fused_job), &dTj, &hx));
        RC(arena.get(size_t(m) * sizeof(havoc_mi355x_rdoq_job), &dRj, &hx));
        RC(arena.get(size_t(m) * 8, &dSj, &hx));
        RC(arena.get(size_t(m) * 4, &dOwner, &hx));
        RC(arena.get(size_t(m) * area * S, &dPred, &hx));
        RC(arena.get(size_t(m) * area * S, &dPiece, &hx));
        RC(arena.get(size_t(m) * area * 2, &dCoef, &hx));
        RC(arena.get(size_t(m) * area * 2, &dLevel, &hx));
        RC(arena.get(havoc_mi355x_rdoq_workspace(m) + 64, &dWork, &hx));
        RC(arena.get(size_t(m) * 4, &dCbf, &hx));
        RC(arena.get(size_t(m) * 4, &dSsd, &hx));
        RC(arena.get(size_t(m) * 8, &dStats, &hx));
        RC(arena.get(size_t(G.n) * sizeof(havoc_mi355x_tu_fused_job), &dFin, &hx));
        RC(arena.get(size_t(G.n) * 4, &dSsd2, &hx));
        RC(arena.get(size_t(G.n) * sizeof(havoc_intra_rd_result), &dOut, &hOut[g], &vOut));
        const havoc_mi355x_tu_fused_job *tj = static_cast<const havoc_mi355x_tu_fused_job *>(dTj);
        RC(havoc_mi355x_intra_expand(ctx, static_cast<const havoc_mi355x_intra_search_job *>(G.d_jobs), static_cast<const int32_t *>(w.dOrder), static_cast<const int32_t *>(w.dCount), static_cast<const int32_t *>(w.dSlot),
                                     G.d_ctx_index, G.n, G.log2, q.quant_scale, q.quant_shift, q.inv_scale, lq, sf, sdh, static_cast<havoc_mi355x_intra_job *>(dIj),
                                     static_cast<havoc_mi355x_tu_fused_job *>(dTj), static_cast<havoc_mi355x_rdoq_job *>(dRj), static_cast<int32_t *>(dSj),
                                     static_cast<int32_t *>(dOwner)));
        RC(havoc_mi355x_intra(ctx, S, bitDepth, G.log2, dPred, nn, G.d_neighbours, static_cast<const havoc_mi355x_intra_job *>(dIj), m));
        RC(havoc_mi355x_tu_forward(ctx, S, bitDepth, tr, G.log2, static_cast<int16_t *>(dCoef), d_src, src_stride, dPred, nn, tj, m));
        RC(havoc_mi355x_rdoq(ctx, bitDepth, G.log2, static_cast<int16_t *>(dLevel), static_cast<const int16_t *>(dCoef), d_states, static_cast<const havoc_mi355x_rdoq_job *>(dRj),
                             m, static_cast<int32_t *>(dCbf), dWork, havoc_mi355x_rdoq_workspace(m)));
        RC(havoc_mi355x_tu_reconstruct(ctx, S, bitDepth, tr, G.log2, q.inv_scale, q.inv_shift, dPiece, nn, dPred, nn, d_src, src_stride, static_cast<const int16_t *>(dLevel), tj,
                                       m, static_cast<uint32_t *>(dSsd)));
        RC(havoc_mi355x_level_stats(ctx, static_cast<const int16_t *>(dLevel), static_cast<const int32_t *>(dSj), m, static_cast<int32_t *>(dStats)));
        RC(havoc_mi355x_intra_decide(ctx, reinterpret_cast<const havoc_mi355x_intra_mpm *>(G.d_ictx), static_cast<const int32_t *>(w.dOrder),
                                     static_cast<const int32_t *>(w.dCount), static_cast<const int32_t *>(w.dSlot), static_cast<const int32_t *>(dCbf),
                                     static_cast<const uint32_t *>(dSsd), static_cast<const int32_t *>(dStats), tj, G.n, G.log2, rl.value,
                                     static_cast<havoc_mi355x_intra_choice *>(vOut), static_cast<havoc_mi355x_tu_fused_job *>(dFin)));
        RC(havoc_mi355x_tu_reconstruct(ctx, S, bitDepth, tr, G.log2, q.inv_scale, q.inv_shift, G.d_rec, nn, dPred, nn, d_src, src_stride, static_cast<const int16_t *>(dLevel),
                                       static_cast<const havoc_mi355x_tu_fused_job *>(dFin), G.n, static_cast<uint32_t *>(dSsd2)));
        st.launches += 8;
        st.candidates += m;
    }
    RC(havoc_mi355x_sync(ctx));
    st.seconds_gpu = now() - tStart;
    const double tHost = now();
    for (int g = 0; g < ngroups; ++g)
        if (groups[g].n) std::memcpy(groups[g].out, hOut[g], size_t(groups[g].n) * sizeof(havoc_intra_rd_result));
    st.seconds_host = now() - tHost;
    st.seconds_total = now() - tStart;
    if (stats) *stats = st;
    return 0;
}

// An INTRA picture with the real dependencies between its partitions (turing/Reconstruct.cpp:609-615: a partition predicts from the reconstruction of the ones before
// it; CandModeList.h:33-95: its most probable modes are its neighbours' champions), level by level WITHOUT a wait between the levels: per level and size
//   intra_gather -> intra_satd35 -> intra_order -> intra_expand -> intra_fill_spare -> intra -> tu_forward -> rdoq -> tu_reconstruct -> level_stats -> intra_decide
//   -> tu_reconstruct (champions) -> intra_commit
// -- the chain runs over n * HAVOC_MI355X_INTRA_MAX_ORDER candidate slots (the spare ones recompute slot 0 into their own space), so the host never has to learn a count,
// and the champions' modes reach intra_commit through the choice records in device memory.  One wait at the end; 40 bytes per partition come back.
int havoc_search_intra_chain(havoc_mi355x_ctx *ctx, int S, int bitDepth, const havoc_mi355x_intra_chain_layout *layout, const void *d_src, intptr_t src_stride, void *d_rec,
                             const int32_t *d_owner, uint8_t *d_modes, const havoc_intra_chain_size *sizes, int nsizes, int nlevels, const uint8_t *d_states,
                             const havoc_rqt_quant quant[4], double reciprocal_sqrt_lambda, double lambda, double reciprocal_lambda, int sdh, havoc_rqt_stats *stats)
{
    if (!ctx || !layout || !d_src || !d_rec || !d_owner || !d_modes || !sizes || nsizes < 0 || nsizes > 4 || nlevels < 0 || !d_states || !quant || (S != 1 && S != 2))
        return HAVOC_MI355X_EINVAL;
    const double tStart = now();
    havoc_rqt_stats st;
    std::memset(&st, 0, sizeof(st));
    Arena arena(ctx);
    Lambda lsq, rl;
    lsq.set(reciprocal_sqrt_lambda);
    rl.set(reciprocal_lambda);
    constexpr int K = HAVOC_MI355X_INTRA_MAX_ORDER;
    struct Work
    {
        void *dCost, *dOrder, *dCount, *dSlot, *dTotal, *hTotal, *dIj, *dTj, *dRj, *dSj, *dOwner, *dPred, *dPiece, *dCoef, *dLevel, *dWork, *dCbf, *dSsd, *dStats, *dFin, *dSsd2, *dChoice, *hChoice;
        int32_t lq, sf;
        size_t workBytes;
    } work[4];
    void *hx;
    for (int g = 0; g < nsizes; ++g)
    {
        const havoc_intra_chain_size &G = sizes[g];
        if (G.log2 < 2 || G.log2 > 5 || G.n < 0 || (G.n && (!G.d_neighbours || !G.d_jobs || !G.d_ictx || !G.d_ctx_index || !G.d_parts || !G.d_blocks || !G.first || !G.out)))
            return HAVOC_MI355X_EINVAL;
        if (!G.n) continue;
        int most = 0;
        for (int l = 0; l < nlevels; ++l)
        {
            if (G.first[l] < 0 || G.first[l + 1] < G.first[l] || G.first[l + 1] > G.n) return HAVOC_MI355X_EINVAL;
            most = std::max(most, G.first[l + 1] - G.first[l]);
        }
        if (G.first[0] != 0 || G.first[nlevels] != G.n) return HAVOC_MI355X_EINVAL;
        Work &w = work[g];
        const size_t cap = size_t(most) * K, area = size_t(1) << (2 * G.log2);
        RC(arena.get(size_t(most) * 35 * 4, &w.dCost, &hx));
        RC(arena.get(size_t(most) * K * 4, &w.dOrder, &hx));
        RC(arena.get(size_t(most) * 4, &w.dCount, &hx));
        RC(arena.get(size_t(most) * 4, &w.dSlot, &hx));
        RC(arena.get(size_t(nlevels) * 8, &w.dTotal, &w.hTotal));      // a (slots handed out, flags) pair PER LEVEL: the flags of every level are looked at after the wait
        RC(arena.get(cap * sizeof(havoc_mi355x_intra_job), &w.dIj, &hx));
        RC(arena.get(cap * sizeof(havoc_mi355x_tu_fused_job), &w.dTj, &hx));
        RC(arena.get(cap * sizeof(havoc_mi355x_rdoq_job), &w.dRj, &hx));
        RC(arena.get(cap * 8, &w.dSj, &hx));
        RC(arena.get(cap * 4, &w.dOwner, &hx));
        RC(arena.get(cap * area * S, &w.dPred, &hx));
        RC(arena.get(cap * area * S, &w.dPiece, &hx));
        RC(arena.get(cap * area * 2, &w.dCoef, &hx));
        RC(arena.get(cap * area * 2, &w.dLevel, &hx));
        w.workBytes = havoc_mi355x_rdoq_workspace(int(cap));
        RC(arena.get(w.workBytes + 64, &w.dWork, &hx));
        RC(arena.get(cap * 4, &w.dCbf, &hx));
        RC(arena.get(cap * 4, &w.dSsd, &hx));
        RC(arena.get(cap * 8, &w.dStats, &hx));
        RC(arena.get(size_t(most) * sizeof(havoc_mi355x_tu_fused_job), &w.dFin, &hx));
        RC(arena.get(size_t(most) * 4, &w.dSsd2, &hx));
        RC(arena.get(size_t(G.n) * sizeof(havoc_intra_rd_result), &w.dChoice, &w.hChoice));
        havoc_mi355x_rdoq_lambda(lambda, quant[G.log2 - 2].inv_scale, &w.lq, &w.sf);
    }
    for (int l = 0; l < nlevels; ++l)
    {
        // Round 6: the partitions of a level are independent of each other WHATEVER their size (that is what a level is), so the sizes' launch chains -- 13 launches each, one
        // behind the other -- go side by side on the context's fork / join lanes instead of in a row on one stream; the join is the level's end (the next level's gathers
        // read what every size committed).  A level then lasts as long as its longest chain, not as long as their sum.
        int active = 0;
        for (int g = 0; g < nsizes; ++g)
            if (sizes[g].n && sizes[g].first[l + 1] > sizes[g].first[l]) ++active;
        const bool forked = active > 1;
        if (forked) RC(havoc_mi355x_fork(ctx, active));
        int laneAt = 0;
        for (int g = 0; g < nsizes; ++g)
        {
            const havoc_intra_chain_size &G = sizes[g];
            if (!G.n) continue;
            const int a = G.first[l], cnt = G.first[l + 1] - a;
            if (!cnt) continue;
            if (forked) RC(havoc_mi355x_lane(ctx, laneAt++));
            const Work &w = work[g];
            const int nn = 1 << G.log2, area = nn * nn, tr = G.log2 == 2 ? 1 : 0, cap = cnt * K;
            const havoc_rqt_quant &q = quant[G.log2 - 2];
            const auto *jobs = static_cast<const havoc_mi355x_intra_search_job *>(G.d_jobs) + a;
            const auto *parts = static_cast<const havoc_mi355x_intra_chain_part *>(G.d_parts) + a;
            auto *mpm = reinterpret_cast<havoc_mi355x_intra_mpm *>(G.d_ictx + a);
            auto *choice = static_cast<havoc_mi355x_intra_choice *>(w.dChoice) + a;
            void *blocks = static_cast<char *>(G.d_blocks) + size_t(a) * area * S;
            const havoc_mi355x_tu_fused_job *tj = static_cast<const havoc_mi355x_tu_fused_job *>(w.dTj);
            RC(havoc_mi355x_intra_gather(ctx, S, layout, d_rec, d_owner, d_modes, parts, cnt, jobs, G.d_neighbours, mpm));
            RC(havoc_mi355x_intra_satd35(ctx, S, bitDepth, G.log2, d_src, src_stride, G.d_neighbours, jobs, cnt, static_cast<int32_t *>(w.dCost)));
            int32_t *total = static_cast<int32_t *>(w.dTotal) + 2 * l;
            RC(havoc_mi355x_intra_order(ctx, static_cast<const int32_t *>(w.dCost), mpm, cnt, lsq.value, static_cast<int32_t *>(w.dOrder), static_cast<int32_t *>(w.dCount),
                                        static_cast<int32_t *>(w.dSlot), total));
            RC(havoc_mi355x_intra_expand(ctx, jobs, static_cast<const int32_t *>(w.dOrder), static_cast<const int32_t *>(w.dCount), static_cast<const int32_t *>(w.dSlot), G.d_ctx_index + a,
                                         cnt, G.log2, q.quant_scale, q.quant_shift, q.inv_scale, w.lq, w.sf, sdh, static_cast<havoc_mi355x_intra_job *>(w.dIj),
                                         static_cast<havoc_mi355x_tu_fused_job *>(w.dTj), static_cast<havoc_mi355x_rdoq_job *>(w.dRj), static_cast<int32_t *>(w.dSj),
                                         static_cast<int32_t *>(w.dOwner)));
            RC(havoc_mi355x_intra_fill_spare(ctx, total, cap, G.log2, static_cast<havoc_mi355x_intra_job *>(w.dIj),
                                             static_cast<havoc_mi355x_tu_fused_job *>(w.dTj), static_cast<havoc_mi355x_rdoq_job *>(w.dRj), static_cast<int32_t *>(w.dSj),
                                             static_cast<int32_t *>(w.dOwner)));
            RC(havoc_mi355x_intra(ctx, S, bitDepth, G.log2, w.dPred, nn, G.d_neighbours, static_cast<const havoc_mi355x_intra_job *>(w.dIj), cap));
            RC(havoc_mi355x_tu_forward(ctx, S, bitDepth, tr, G.log2, static_cast<int16_t *>(w.dCoef), d_src, src_stride, w.dPred, nn, tj, cap));
            RC(havoc_mi355x_rdoq(ctx, bitDepth, G.log2, static_cast<int16_t *>(w.dLevel), static_cast<const int16_t *>(w.dCoef), d_states, static_cast<const havoc_mi355x_rdoq_job *>(w.dRj),
                                 cap, static_cast<int32_t *>(w.dCbf), w.dWork, havoc_mi355x_rdoq_workspace(cap)));
            RC(havoc_mi355x_tu_reconstruct(ctx, S, bitDepth, tr, G.log2, q.inv_scale, q.inv_shift, w.dPiece, nn, w.dPred, nn, d_src, src_stride, static_cast<const int16_t *>(w.dLevel), tj,
                                           cap, static_cast<uint32_t *>(w.dSsd)));
            RC(havoc_mi355x_level_stats(ctx, static_cast<const int16_t *>(w.dLevel), static_cast<const int32_t *>(w.dSj), cap, static_cast<int32_t *>(w.dStats)));
            RC(havoc_mi355x_intra_decide(ctx, mpm, static_cast<const int32_t *>(w.dOrder), static_cast<const int32_t *>(w.dCount), static_cast<const int32_t *>(w.dSlot),
                                         static_cast<const int32_t *>(w.dCbf), static_cast<const uint32_t *>(w.dSsd), static_cast<const int32_t *>(w.dStats), tj, cnt, G.log2, rl.value,
                                         choice, static_cast<havoc_mi355x_tu_fused_job *>(w.dFin)));
            RC(havoc_mi355x_tu_reconstruct(ctx, S, bitDepth, tr, G.log2, q.inv_scale, q.inv_shift, blocks, nn, w.dPred, nn, d_src, src_stride, static_cast<const int16_t *>(w.dLevel),
                                           static_cast<const havoc_mi355x_tu_fused_job *>(w.dFin), cnt, static_cast<uint32_t *>(w.dSsd2)));
            RC(havoc_mi355x_intra_commit(ctx, S, layout, d_rec, d_modes, parts, cnt, blocks, reinterpret_cast<const int32_t *>(choice), int(sizeof(havoc_mi355x_intra_choice) / 4)));
            st.launches += 13;
            st.candidates += cap;
        }
        if (forked) RC(havoc_mi355x_join(ctx));
    }
    for (int g = 0; g < nsizes; ++g)
        if (sizes[g].n)
        {
            RC(havoc_mi355x_d2h_async(ctx, work[g].hChoice, work[g].dChoice, size_t(sizes[g].n) * sizeof(havoc_intra_rd_result)));
            RC(havoc_mi355x_d2h_async(ctx, work[g].hTotal, work[g].dTotal, size_t(nlevels) * 8));
        }
    RC(havoc_mi355x_sync(ctx));
    // havoc_mi355x_intra_order's flags of every level: bit 0 = a partition's refinement order was cut at HAVOC_MI355X_INTRA_MAX_ORDER, bit 1 = a record was
    // out of range -- the champions are then not the reference's: no result is returned
    int flags = 0;
    for (int g = 0; g < nsizes; ++g)
        if (sizes[g].n)
            for (int l = 0; l < nlevels; ++l)
                if (sizes[g].first[l + 1] > sizes[g].first[l]) flags |= static_cast<const int32_t *>(work[g].hTotal)[2 * l + 1];
    if (flags & 3) return HAVOC_SEARCH_EORDER - (flags & 3);
    st.seconds_gpu = now() - tStart;
    for (int g = 0; g < nsizes; ++g)
        if (sizes[g].n) std::memcpy(sizes[g].out, work[g].hChoice, size_t(sizes[g].n) * sizeof(havoc_intra_rd_result));
    st.seconds_total = now() - tStart;
    if (stats) *stats = st;
    return 0;
}

} // extern "C"
